// Hand mathematics for gfx950, designed for one-wavefront-per-hand execution:
//   * the 34-tile histogram lives in 4 VGPRs as packed 3-bit counters (no per-lane arrays,
//     no scratch): s[0..2] = man/pin/sou (9 x 3 bit), s[3] = honors (7 x 3 bit);
//   * win-shape test is closed-form per suit (greedy mentsu peel, exact for the boolean
//     question the reference answers by backtracking, agari.rs:183-245);
//   * waits = 34 lanes x one candidate tile each, gathered with a wavefront ballot
//     (hand_evaluator.rs:196-213);
//   * yaku/fu: rmj_eval4.hip.h (row form: lane = decision string of the division search, then lane = (division, winning
//     group) candidate; the row maximum of (han, fu, reference order) is the reference's "strictly better replaces" scan).
//
// Reference semantics followed: agari.rs, hand_evaluator.rs, hand_evaluator_3p.rs, yaku.rs,
// yaku_3p.rs, score.rs (riichienv-core/src).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rmj {

// ---------------------------------------------------------------- packed histogram
struct PH {
    uint32_t a, b, c, d;  // man, pin, sou, honors
};
#define O9_1 0111111111u /* bit0 of each of 9 fields */
#define O9_2 0222222222u
#define O9_4 0444444444u
#define O7_1 01111111u
#define O7_2 02222222u
#define O7_4 04444444u

// Sum over each 16-lane row (DPP row_shr 1,2,4,8: no LDS traffic); the total lands in lane 15 of the row.
__device__ __forceinline__ uint32_t row_sum16(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    return v;
}
// The 16-bit slice of a 64-lane ballot that belongs to the lane's own row (rb = lane & 48): ONE v_perm_b32 - a byte permute of
// {hi, lo} under a per-lane selector (row k takes bytes 2k, 2k + 1; 0x0C = constant zero) - behind one v_mov of the upper half.
// Rounds 2-5 wrote it as "select the half, then bit-field extract", which LLVM canonicalises to a 64-bit shift of the ballot by a
// vector register (v_lshrrev_b64 v[a:b], v_amount, vcc).  On gfx950 that instruction returns a WRONG value now and then when the
// shift amount sits in the LAST vector register the wave is allocated and the wave is not the first of its SIMD
// (scripts/micro/ballot_shift_hazard.hip: ~1e-3 of the executions in wave slots >= 3, none in slot 0, none with the amount one
// register lower) - the build with -mllvm -disable-machine-licm put the amount into v87 of the 88-register k_step4_act_enc and
// published legal lists with entries missing (docs/journal_r06.md section 1; scripts/lint_isa_last_vgpr.py guards every build).
// The permute has no 64-bit operand, needs one lane-constant register (the selector) instead of two (rb & 32, rb & 16), and is two
// full-rate instructions.
#ifndef RMJ_ROW_BALLOT_SHIFT64
#define RMJ_ROW_BALLOT_SHIFT64 0   /* 1: the form of rounds 2-5 (A/B and the reproduction of the hazard only; scripts/lint_isa_last_vgpr.py then decides whether a build is safe) */
#endif
__device__ __forceinline__ uint32_t row_ballot16(bool p, int rb) {
    const uint64_t b = __ballot(p);
#if RMJ_ROW_BALLOT_SHIFT64
    const uint32_t w = (rb & 32) ? (uint32_t)(b >> 32) : (uint32_t)b;
    return __builtin_amdgcn_ubfe(w, (uint32_t)(rb & 16), 16u);
#else
    const uint32_t sel = 0x0C0C0100u + ((uint32_t)rb >> 4) * 0x0202u;
    return __builtin_amdgcn_perm((uint32_t)(b >> 32), (uint32_t)b, sel);
#endif
}
__device__ __forceinline__ int t_suit(int t) { return t >= 27 ? 3 : (t >= 18 ? 2 : (t >= 9 ? 1 : 0)); }
__device__ __forceinline__ uint32_t ph_get(const PH& h, int s) { return s == 0 ? h.a : (s == 1 ? h.b : (s == 2 ? h.c : h.d)); }
__device__ __forceinline__ void ph_addv(PH& h, int s, uint32_t v) {
    h.a += (s == 0) ? v : 0u;
    h.b += (s == 1) ? v : 0u;
    h.c += (s == 2) ? v : 0u;
    h.d += (s == 3) ? v : 0u;
}
__device__ __forceinline__ void ph_add(PH& h, int t) {
    int s = t_suit(t);
    ph_addv(h, s, 1u << (3 * (t - 9 * s)));
}
__device__ __forceinline__ void ph_sub(PH& h, int t) {
    int s = t_suit(t);
    ph_addv(h, s, 0u - (1u << (3 * (t - 9 * s))));
}
__device__ __forceinline__ int ph_cnt(const PH& h, int t) {
    int s = t_suit(t);
    return (ph_get(h, s) >> (3 * (t - 9 * s))) & 7;
}
__device__ __forceinline__ int field_sum(uint32_t x) {  // sum of the nine 3-bit fields (each <= 4)
    uint32_t y = (x & 0707070707u) + ((x >> 3) & 0707070707u);  // 6-bit fields at 0,6,..,24
    return (int)(((y * 0x01041041u) >> 24) & 0x3Fu);
}
__device__ __forceinline__ int ph_total(const PH& h) { return field_sum(h.a) + field_sum(h.b) + field_sum(h.c) + field_sum(h.d); }
// first tile type with a non-zero count (wave lanes may differ); h must be non-empty
__device__ __forceinline__ bool ph_empty(const PH& h) { return (h.a | h.b | h.c | h.d) == 0u; }
__device__ __forceinline__ int ph_first(const PH& h) {
    if (h.a) return (__ffs((int)h.a) - 1) / 3;
    if (h.b) return 9 + (__ffs((int)h.b) - 1) / 3;
    if (h.c) return 18 + (__ffs((int)h.c) - 1) / 3;
    return 27 + (__ffs((int)h.d) - 1) / 3;
}
// 34-bit presence mask (bit t set iff count[t] > 0)
__device__ __forceinline__ uint64_t ph_presence(const PH& h) {
    uint64_t r = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        r |= (uint64_t)(((h.a >> (3 * i)) & 7u) != 0) << i;
        r |= (uint64_t)(((h.b >> (3 * i)) & 7u) != 0) << (9 + i);
        r |= (uint64_t)(((h.c >> (3 * i)) & 7u) != 0) << (18 + i);
    }
#pragma unroll
    for (int i = 0; i < 7; i++) r |= (uint64_t)(((h.d >> (3 * i)) & 7u) != 0) << (27 + i);
    return r;
}

// tile-type class masks (bit per 34-type)
__device__ __forceinline__ constexpr uint64_t mk_mask_number_terminals() {
    return (1ull << 0) | (1ull << 8) | (1ull << 9) | (1ull << 17) | (1ull << 18) | (1ull << 26);
}
#define MASK_NUMTERM (mk_mask_number_terminals())
#define MASK_HONORS (0x7Full << 27)
#define MASK_TERM (MASK_NUMTERM | MASK_HONORS)
#define MASK_MAN (0x1FFull)
#define MASK_PIN (0x1FFull << 9)
#define MASK_SOU (0x1FFull << 18)
#define MASK_GREEN ((1ull << 19) | (1ull << 20) | (1ull << 21) | (1ull << 23) | (1ull << 25) | (1ull << 32))

// ---------------------------------------------------------------- win shape (agari.rs)
// Exact test "x decomposes into koutsu/shuntsu only" for one numbered suit.
// Lowest tile with count c must start (c mod 3) shuntsu (3 shuntsu == 3 koutsu), so the peel is forced.
__device__ __forceinline__ bool mentsu_ok(uint32_t x) {
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        uint32_t c = x & 7u;
        uint32_t r = c >= 3u ? c - 3u : c;
        if (i <= 6) {
            uint32_t n1 = (x >> 3) & 7u, n2 = (x >> 6) & 7u;
            ok = ok && (n1 >= r) && (n2 >= r);
            x -= r * ((1u << 3) | (1u << 6));
        } else {
            ok = ok && (r == 0u);
        }
        x >>= 3;
        x &= 0x07FFFFFFu;  // keep garbage from failed borrows out of the way
    }
    return ok;
}
__device__ __forceinline__ bool suit_pair_ok(uint32_t x) {  // pair + mentsu within one numbered suit
    bool any = false;
#pragma unroll
    for (int j = 0; j < 9; j++) {
        uint32_t c = (x >> (3 * j)) & 7u;
        if (c >= 2u) any = any || mentsu_ok(x - (2u << (3 * j)));
    }
    return any;
}
__device__ __forceinline__ bool honors_ok0(uint32_t x) { return (x & O7_4) == 0u && (((x ^ (x >> 1)) & O7_1) == 0u); }
__device__ __forceinline__ bool honors_ok2(uint32_t x) {
    uint32_t m2 = (x >> 1) & ~x & O7_1;  // fields equal to 2 (bit2 known clear below)
    return (x & O7_4) == 0u && __popc(m2) == 1 && (((x ^ (x >> 1)) & O7_1) == m2);
}
// agari.rs:143-168 (checks only the 13 terminal kinds, like the reference)
__device__ __forceinline__ bool is_kokushi(const PH& h) {
    uint32_t hz = (h.d | (h.d >> 1) | (h.d >> 2)) & O7_1;
    if (hz != O7_1) return false;
    int pairs = 0;
    bool ok = true;
#pragma unroll
    for (int s = 0; s < 3; s++) {
        uint32_t x = ph_get(h, s);
        uint32_t c0 = x & 7u, c8 = (x >> 24) & 7u;
        ok = ok && c0 >= 1u && c0 <= 2u && c8 >= 1u && c8 <= 2u;
        pairs += (c0 == 2u) + (c8 == 2u);
    }
#pragma unroll
    for (int i = 0; i < 7; i++) {
        uint32_t c = (h.d >> (3 * i)) & 7u;
        ok = ok && c <= 2u;
        pairs += (c == 2u);
    }
    return ok && pairs == 1;
}
// agari.rs:170-181
__device__ __forceinline__ bool is_chiitoi(const PH& h) {
    uint32_t bad = (h.a & (O9_1 | O9_4)) | (h.b & (O9_1 | O9_4)) | (h.c & (O9_1 | O9_4)) | (h.d & (O7_1 | O7_4));
    int pairs = __popc(h.a & O9_2) + __popc(h.b & O9_2) + __popc(h.c & O9_2) + __popc(h.d & O7_2);
    return bad == 0u && pairs == 7;
}
// agari.rs:183-199 (boolean result only)
__device__ __forceinline__ bool is_standard_agari(const PH& h) {
    int ta = field_sum(h.a) % 3, tb = field_sum(h.b) % 3, tc = field_sum(h.c) % 3, td = field_sum(h.d) % 3;
    int n2 = (ta == 2) + (tb == 2) + (tc == 2) + (td == 2);
    int n0 = (ta == 0) + (tb == 0) + (tc == 0) + (td == 0);
    if (n2 != 1 || n0 != 3) return false;
    bool ok = true;
    ok = ok && (ta == 2 ? suit_pair_ok(h.a) : mentsu_ok(h.a));
    ok = ok && (tb == 2 ? suit_pair_ok(h.b) : mentsu_ok(h.b));
    ok = ok && (tc == 2 ? suit_pair_ok(h.c) : mentsu_ok(h.c));
    ok = ok && (td == 2 ? honors_ok2(h.d) : honors_ok0(h.d));
    return ok;
}
// agari.rs:65-73
__device__ __forceinline__ bool is_agari(const PH& h) { return is_kokushi(h) || is_chiitoi(h) || is_standard_agari(h); }

// pair + mentsu within one numbered suit, three candidates only: with the pair at rank j and everything else in
// mentsu, sum(rank * count) = 2j (mod 3) (a shuntsu adds 3i+3, a koutsu 3i), hence j = 2 * sum (mod 3).
__device__ __forceinline__ int pair_residue(uint32_t x) {
    int s1 = field_sum(x & 0070070070u), s2 = field_sum(x & 0700700700u);  // ranks 1,4,7 / 2,5,8
    return (2 * (s1 + 2 * s2)) % 3;
}

// hand_evaluator.rs:196-213 : lane t < 34 tests tile t; returns the wave-uniform 34-bit wait mask.
// `h` must be wave-uniform; caller guarantees current_total == 13.
//
// Adding tile t changes ONE suit word, so the standard-form test factorises: lanes 0..33 judge their modified suit
// word, lanes 34..37 judge the four unmodified words in the same instruction stream, and a candidate wins iff its own
// word and the three unmodified others are all consistent ("mentsu only" when the word holds 0 mod 3 tiles,
// "pair + mentsu" when it holds 2 mod 3) with exactly one pair suit (agari.rs:183-245, boolean result).
// Out of line on purpose: one shared copy for the ~10 call sites of the step kernel (instruction-cache footprint).
__device__ __noinline__ uint64_t wave_waits(PH h, int lane) {
    h.a = __builtin_amdgcn_readfirstlane(h.a); h.b = __builtin_amdgcn_readfirstlane(h.b);
    h.c = __builtin_amdgcn_readfirstlane(h.c); h.d = __builtin_amdgcn_readfirstlane(h.d);
    const bool cand = lane < 34;
    const int s = cand ? t_suit(lane) : ((lane - 34) & 3);
    uint32_t w = ph_get(h, s);
    bool live = lane < 38;
    if (cand) {
        int sh = 3 * (lane - 9 * s);
        live = ((w >> sh) & 7u) < 4u;
        w += 1u << sh;
    }
    const int tot = field_sum(w) % 3;
    bool r = false;
    if (live && tot != 1) {
        if (s == 3) {
            r = tot == 0 ? honors_ok0(w) : honors_ok2(w);
        } else {
            // first pass serves both modes: the word itself (mentsu only) or the word minus the first pair candidate
            int j = pair_residue(w);
            uint32_t y = w;
            bool go = true;
            if (tot == 2) {
                go = ((w >> (3 * j)) & 7u) >= 2u;
                y = w - (2u << (3 * j));
            }
            r = go && mentsu_ok(y);
            if (tot == 2 && !r) {
#pragma unroll 1
                for (int q = 0; q < 2 && !r; q++) {
                    j += 3;
                    if (((w >> (3 * j)) & 7u) >= 2u) r = mentsu_ok(w - (2u << (3 * j)));
                }
            }
        }
    }
    const uint64_t R = __ballot(r), T2 = __ballot(tot == 2);
    const uint32_t bR = (uint32_t)(R >> 34) & 15u, bT2 = (uint32_t)(T2 >> 34) & 15u;
    const uint32_t others = 15u & ~(1u << s);
    bool win = cand && live && r && (bR & others) == others && (__popc(bT2 & others) + (tot == 2 ? 1 : 0)) == 1;
    // chiitoi needs six pairs among the 13 tiles, kokushi twelve of the thirteen terminal kinds (wave-uniform gates)
    const int pairs = __popc(h.a & O9_2 & ~(h.a << 1)) + __popc(h.b & O9_2 & ~(h.b << 1)) + __popc(h.c & O9_2 & ~(h.c << 1)) +
                      __popc(h.d & O7_2 & ~(h.d << 1));  // fields equal to 2 or 6
    uint32_t pres_d = (h.d | (h.d >> 1) | (h.d >> 2)) & O7_1;
    const int kinds = __popc(pres_d) + ((h.a & 7u) != 0u) + ((h.a >> 24) != 0u) + ((h.b & 7u) != 0u) + ((h.b >> 24) != 0u) +
                      ((h.c & 7u) != 0u) + ((h.c >> 24) != 0u);
    if (pairs >= 6 || kinds >= 12) {
        if (cand && live && !win) {
            PH x = h;
            ph_add(x, lane);
            win = is_kokushi(x) || is_chiitoi(x);
        }
    }
    return __ballot(win) & 0x3FFFFFFFFull;
}

// ---------------------------------------------------------------- melds (wave-uniform aggregate)
struct MeldAgg {
    int n, n_kan, n_ankan, n_nonchi;
    bool menzen;              // all melds !opened (hand_evaluator.rs:136)
    uint64_t types;           // presence mask over all meld tiles (34-types)
    uint64_t mtypes[4];       // per-meld presence mask
    uint8_t mtype[4];         // RMJ_MELD_*
    uint8_t t0[4];            // tiles[0] as 34-type (chi: lowest, hand_evaluator.rs:63-65)
    int fu;                   // summed meld fu (yaku.rs:614-632)
    int aka;                  // red fives among meld tiles
};

// ---------------------------------------------------------------- yaku context (types.rs:193-210 -> yaku.rs:192-209)
#define CF_TSUMO 1u
#define CF_RIICHI 2u
#define CF_DOUBLE_RIICHI 4u
#define CF_IPPATSU 8u
#define CF_HAITEI 16u
#define CF_HOUTEI 32u
#define CF_RINSHAN 64u
#define CF_CHANKAN 128u
#define CF_FIRST_TURN 256u

struct CalcIn {
    PH hand14;      // concealed histogram incl. the win tile
    MeldAgg ma;
    int win34;
    uint32_t cf;    // CF_*
    int dora, aka, ura, nuki;
    int round_wind34, seat_wind34;
    bool sanma;
    uint32_t honba;
};
struct CalcOut {
    bool shape;     // has_win_shape (is_agari)
    bool is_win, yakuman;
    int han, fu, yakuman_count;
    int kind;       // 0 normal, 1 yakuman-only (division path), 2 chiitoi path, 3 kokushi
    uint64_t ym;    // bit id = yaku id present
    uint32_t ron, tsumo_oya, tsumo_ko;
};

#define YB(id) (1ull << (id))
#define YMASK_DORA (YB(31) | YB(32) | YB(33) | YB(34))

// score.rs:13-52
struct ScoreOut {
    uint32_t total, ron, tsumo_oya, tsumo_ko;
};
__device__ __forceinline__ uint32_t ceil100(uint32_t v) { return (v + 99u) / 100u * 100u; }
__device__ inline ScoreOut calc_score(uint32_t han, uint32_t fu, bool is_oya, bool is_tsumo, uint32_t honba, uint32_t np) {
    uint32_t base;
    if (han >= 5u) {
        base = han == 5u ? 2000u : (han <= 7u ? 3000u : (han <= 10u ? 4000u : (han <= 12u ? 6000u : 8000u * (han / 13u))));
    } else {
        uint32_t f = (fu == 25u) ? 25u : (fu + 9u) / 10u * 10u;
        f &= 0xFFu;
        uint32_t bp = f << (2u + han);
        base = bp > 2000u ? 2000u : bp;
    }
    uint32_t total_ron = is_oya ? ceil100(base * 6u) : ceil100(base * 4u);
    uint32_t pay_oya = is_oya ? 0u : ceil100(base * 2u);
    uint32_t pay_ko = is_oya ? ceil100(base * 2u) : ceil100(base);
    uint32_t total_tsumo = is_oya ? pay_ko * (np - 1u) : pay_oya + pay_ko * (np - 2u);
    ScoreOut s;
    if (is_tsumo) {
        s.total = total_tsumo + honba * 100u * (np - 1u);
        s.ron = 0u;
        s.tsumo_oya = pay_oya + honba * 100u;
        s.tsumo_ko = pay_ko + honba * 100u;
    } else {
        uint32_t hr = honba * 100u * (np - 1u);
        s.total = total_ron + hr;
        s.ron = total_ron + hr;
        s.tsumo_oya = 0u;
        s.tsumo_ko = 0u;
    }
    return s;
}

__device__ __forceinline__ bool is_aka(int t) { return t == 16 || t == 52 || t == 88; }
// body code of a division: 4 x 8 bit, each (koutsu << 6) | lowest type
__device__ __forceinline__ int b_tile(uint32_t body, int i) { return (body >> (8 * i)) & 0x3F; }
__device__ __forceinline__ bool b_kou(uint32_t body, int i) { return (body >> (8 * i + 6)) & 1u; }
__device__ __forceinline__ bool t_is_terminal(int t) { return (MASK_TERM >> t) & 1ull; }
__device__ __forceinline__ bool t_is_numterm(int t) { return (MASK_NUMTERM >> t) & 1ull; }

// (the yaku / fu evaluation itself: rmj_eval4.hip.h - four hands per wave, one 16-lane row per hand; round 4 retired the one-hand-per-wave
//  evaluator that lived here: lane = candidate head, 236 registers, 71 M hands/s)

// yaku id emission orders: list = order table filtered by the mask
__device__ __constant__ const uint8_t ORDER_STATIC[12] = {2, 18, 30, 1, 5, 6, 4, 3, 31, 32, 33, 34};
__device__ __constant__ const uint8_t ORDER_YAKUMAN[13] = {39, 41, 40, 44, 47, 45, 35, 36, 48, 38, 37, 50, 43};
__device__ __constant__ const uint8_t ORDER_NORMAL_TAIL[21] = {12, 14, 7, 8, 9, 11, 10, 23, 21, 22, 20, 28, 13, 16, 17, 19, 29, 27, 24, 26, 15};
__device__ __constant__ const uint8_t ORDER_CHIITOI_HEAD[5] = {25, 12, 29, 27, 24};

__device__ inline int yaku_list(int kind, uint64_t ym, uint8_t* out, int cap) {
    int n = 0;
    auto emit = [&](const uint8_t* tab, int len) {
        for (int i = 0; i < len; i++)
            if ((ym >> tab[i]) & 1ull) {
                if (n < cap) out[n] = tab[i];
                n++;
            }
    };
    if (kind == 0) { emit(ORDER_STATIC, 12); emit(ORDER_NORMAL_TAIL, 21); }
    else if (kind == 1) { emit(ORDER_YAKUMAN, 13); }
    else if (kind == 2) { emit(ORDER_CHIITOI_HEAD, 5); emit(ORDER_YAKUMAN, 13); emit(ORDER_STATIC, 12); }
    else { if ((ym >> 49) & 1ull) out[n++] = 49; else if ((ym >> 42) & 1ull) out[n++] = 42; }
    return n < cap ? n : cap;
}

}  // namespace rmj
