// HandEvaluator::calc in ROW FORM: four hands per wavefront, one 16-lane DPP row per hand (round 4).
//
// The one-hand-per-wave evaluator (wave_calc, rmj_hand.hip.h) gives every candidate HEAD a lane and lets that lane walk all 16
// decision strings of the reference's division search and every winning group one after the other: at most five of its 64 lanes
// do anything, each of them runs a serial loop of up to 16 x 5 yaku evaluations, and the compiler needs 236 registers for it
// (576 B of scratch per lane under the occupancy the kernel wants).  Here the two loops are turned into lanes:
//   * a standard hand has its pair in the ONE suit whose tile count is 2 mod 3, at a rank j = 2 * sum(rank * count) mod 3
//     (pair_residue) - at most three head candidates, tried in ascending order like agari.rs:75-94;
//   * per head candidate, lane = decision string (16 of them: koutsu-before-shuntsu at each of the four sets, agari.rs:96-141):
//     one forced peel per lane, a row ballot collects the valid divisions in the reference's order;
//   * lane = (division, winning group) for the yaku / fu evaluation (yaku.rs:298-556): three divisions x (pair + four sets) per
//     pass, one pass for all but a handful of hands; the row maximum of (han, fu, reference order) is the reference's
//     "strictly better replaces" scan.
// Everything per hand is a per-lane value that is uniform inside its row; the melds arrive as five packed words (E4Meld) reduced
// over lane = meld.  No arrays, no scratch.  The same function serves the batched kernel gate (k_eval_hands) and the settlement
// code of the step kernel.
//
// Reference: hand_evaluator.rs:77-176, agari.rs:65-141, yaku.rs:232-1280 (riichienv-core/src); results identical to wave_calc.
#pragma once
#include "rmj_hand.hip.h"

namespace rmj {

// ---------------------------------------------------------------- row helpers (r = lane & 15, rb = lane & 48)
__device__ __forceinline__ uint32_t e4_ballot(bool p, int rb) { return row_ballot16(p, rb); }   // (one byte permute: see row_ballot16)
__device__ __forceinline__ int e4_bc(int v, int src_lane) { return __builtin_amdgcn_ds_bpermute(src_lane << 2, v); }
__device__ __forceinline__ uint32_t e4_rsum(uint32_t v, int rb) { return (uint32_t)e4_bc((int)row_sum16(v), rb + 15); }
__device__ __forceinline__ uint32_t e4_ror(uint32_t v, int rb) {
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    return (uint32_t)e4_bc((int)v, rb + 15);
}
__device__ __forceinline__ uint32_t e4_rmax(uint32_t v, int rb) {
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false));
    return (uint32_t)e4_bc((int)v, rb + 15);
}

// ---------------------------------------------------------------- melds as packed words
// Row-uniform aggregate of a hand's melds = everything MeldAgg carried, in five words.  A number tile's bit index 9 * suit + rank
// is its 34-type, so "chi starting at" / "pon or kan of" are plain type masks.
struct E4Meld {
    uint32_t chi;       // bit t (< 27): a chi whose lowest tile is type t; E4M_* flags above
    uint32_t kou;       // bit t (< 27): a pon / kan of type t
    uint32_t cnt;       // sums: melds @0 (3 bits), kans @3, ankans @6, non-chi @9, fu @12 (8 bits), red fives @20 (4 bits)
    uint32_t tlo, thi;  // presence mask of all meld tiles (types 0..31 / 32..33); thi bits 8..14: pon / kan of honor type 27 + k
};
#define E4M_NO_NUMTERM (1u << 27)   /* some meld holds no 1 / 9 (junchan) */
#define E4M_NO_TERM (1u << 28)      /* some meld holds neither a terminal nor an honor (chanta) */
#define E4M_OPENED (1u << 29)       /* some meld is open (menzen = none) */
__device__ __forceinline__ int e4m_n(const E4Meld& m) { return (int)(m.cnt & 7u); }
__device__ __forceinline__ int e4m_kan(const E4Meld& m) { return (int)((m.cnt >> 3) & 7u); }
__device__ __forceinline__ int e4m_ankan(const E4Meld& m) { return (int)((m.cnt >> 6) & 7u); }
__device__ __forceinline__ int e4m_nonchi(const E4Meld& m) { return (int)((m.cnt >> 9) & 7u); }
__device__ __forceinline__ int e4m_fu(const E4Meld& m) { return (int)((m.cnt >> 12) & 0xFFu); }
__device__ __forceinline__ int e4m_aka(const E4Meld& m) { return (int)((m.cnt >> 20) & 0xFu); }
__device__ __forceinline__ bool e4m_menzen(const E4Meld& m) { return (m.chi & E4M_OPENED) == 0u; }
__device__ __forceinline__ uint64_t e4m_types(const E4Meld& m) { return (uint64_t)m.tlo | ((uint64_t)(m.thi & 3u) << 32); }
__device__ __forceinline__ uint32_t e4m_honors(const E4Meld& m) { return (m.thi >> 8) & 0x7Fu; }

// The words of ONE meld, computed by its lane (`valid`: the lane holds a meld).  nt tiles (136-ids) in t0id..t3id; t0 = the type
// the meld counts as (chi: its lowest type, hand_evaluator.rs:63-65); trip = a pon / kan (its fu: yaku.rs:614-632); opened.
__device__ __forceinline__ E4Meld e4_meld_lane(bool valid, uint32_t type, int nt, uint32_t t0id, uint32_t t1id, uint32_t t2id, uint32_t t3id, int t0,
                                               bool trip, bool opened) {
    E4Meld m = {0u, 0u, 0u, 0u, 0u};
    if (valid) {
        uint64_t mm = 0ull;
        uint32_t aka = 0u;
        if (nt > 0) { mm |= 1ull << (t0id >> 2); aka += is_aka((int)t0id); }
        if (nt > 1) { mm |= 1ull << (t1id >> 2); aka += is_aka((int)t1id); }
        if (nt > 2) { mm |= 1ull << (t2id >> 2); aka += is_aka((int)t2id); }
        if (nt > 3) { mm |= 1ull << (t3id >> 2); aka += is_aka((int)t3id); }
        const bool chi = type == RMJ_MELD_CHI;
        const uint32_t bit = t0 < 27 ? 1u << t0 : 0u;
        m.chi = chi ? bit : 0u;
        m.kou = chi ? 0u : bit;
        if (!(mm & MASK_NUMTERM)) m.chi |= E4M_NO_NUMTERM;
        if (!(mm & MASK_TERM)) m.chi |= E4M_NO_TERM;
        if (opened) m.chi |= E4M_OPENED;
        m.tlo = (uint32_t)mm;
        m.thi = (uint32_t)(mm >> 32) & 3u;
        if (!chi && t0 >= 27 && t0 < 34) m.thi |= 1u << (8 + t0 - 27);
        const bool kan = type >= RMJ_MELD_DAIMINKAN;
        uint32_t f = 0u;
        if (trip) {
            f = opened ? 2u : 4u;
            if (t_is_terminal(t0 < 34 ? t0 : 1)) f *= 2u;
            if (kan) f *= 4u;
        }
        m.cnt = 1u | ((uint32_t)kan << 3) | ((uint32_t)(type == RMJ_MELD_ANKAN) << 6) | ((uint32_t)!chi << 9) | (f << 12) | (aka << 20);
    }
    return m;
}
__device__ __forceinline__ E4Meld e4_meld_reduce(const E4Meld& p, int rb) {
    E4Meld m;
    m.chi = e4_ror(p.chi, rb); m.kou = e4_ror(p.kou, rb); m.tlo = e4_ror(p.tlo, rb); m.thi = e4_ror(p.thi, rb);
    m.cnt = e4_rsum(p.cnt, rb);
    return m;
}

// ---------------------------------------------------------------- inputs / outputs (row-uniform)
struct E4In {
    bool on;            // the row holds a hand to evaluate
    PH hand14;          // concealed histogram incl. the winning tile
    E4Meld ma;
    int win34;
    uint32_t cf;        // CF_*
    int dora, aka, ura, nuki;
    int round_wind34, seat_wind34;
    bool sanma;
    uint32_t honba;
};
struct E4Out {
    bool shape;         // has_win_shape (is_agari)
    bool is_win, yakuman;
    int han, fu, yakuman_count;
    int kind;           // 0 normal, 1 yakuman-only (division path), 2 chiitoi path, 3 kokushi
    uint64_t ym;
    uint32_t ron, tsumo_oya, tsumo_ko;
};

// hand-level predicates (yaku.rs:692-705, 777-841, 1057-1146, 1212-1228), straight from the packed words
#define E4F_TANYAO 1u
#define E4F_CHINITSU 2u
#define E4F_HONITSU 4u
#define E4F_HONROUTOU 8u
#define E4F_TSUUIISOU 16u
#define E4F_CHINROUTOU 32u
#define E4F_RYUUIISOU 64u
#define E4F_CHUUREN 128u
#define E4F_CHUUREN9 256u
__device__ __forceinline__ uint32_t e4_hand_flags(const E4In& in) {
    const PH& h = in.hand14;
    const uint64_t mt = e4m_types(in.ma);
    const uint32_t T9 = 1u | (1u << 24);
    const uint32_t na = (h.a | (h.a >> 1) | (h.a >> 2)) & O9_1, nb = (h.b | (h.b >> 1) | (h.b >> 2)) & O9_1, nc = (h.c | (h.c >> 1) | (h.c >> 2)) & O9_1;
    const uint32_t nd = (h.d | (h.d >> 1) | (h.d >> 2)) & O7_1;
    const bool man = na != 0u || (mt & MASK_MAN) != 0ull, pin = nb != 0u || (mt & MASK_PIN) != 0ull, sou = nc != 0u || (mt & MASK_SOU) != 0ull;
    const bool honor = nd != 0u || (mt & MASK_HONORS) != 0ull;
    const bool numterm = ((na | nb | nc) & T9) != 0u || (mt & MASK_NUMTERM) != 0ull;
    const bool simples = ((na | nb | nc) & ~T9) != 0u || (mt & ~MASK_TERM) != 0ull;
    const int suits = (int)man + (int)pin + (int)sou;
    uint32_t f = 0u;
    if (!numterm && !honor) f |= E4F_TANYAO;
    if (suits == 1 && !honor) f |= E4F_CHINITSU;
    if (suits == 1 && honor) f |= E4F_HONITSU;
    if (!simples) f |= E4F_HONROUTOU;
    if (suits == 0) f |= E4F_TSUUIISOU;
    if (!simples && !honor) f |= E4F_CHINROUTOU;
    // all green: 2s 3s 4s 6s 8s + hatsu
    const uint32_t green_c = (1u << 3) | (1u << 6) | (1u << 9) | (1u << 15) | (1u << 21);
    if (na == 0u && nb == 0u && (nc & ~green_c) == 0u && (nd & ~(1u << 15)) == 0u && (mt & ~MASK_GREEN) == 0ull) f |= E4F_RYUUIISOU;
    // chuuren: the concealed tiles only (yaku.rs:1101-1131)
    const int hs = (int)(na != 0u) + (int)(nb != 0u) + (int)(nc != 0u);
    const uint32_t x = na ? h.a : (nb ? h.b : h.c), nx = na ? na : (nb ? nb : nc);
    if (hs == 1 && nd == 0u && (x & 7u) >= 3u && ((x >> 24) & 7u) >= 3u && nx == O9_1) f |= E4F_CHUUREN;
    if (in.win34 < 27) {  // yaku.rs:1133-1146 - counts of the WIN TILE's suit
        const int ws = in.win34 / 9, val = in.win34 - 9 * ws;
        const uint32_t c = (ph_get(h, ws) >> (3 * val)) & 7u;
        if ((val == 0 || val == 8) ? (c == 4u) : (c == 2u)) f |= E4F_CHUUREN9;
    }
    return f;
}

// yaku.rs:892-1055.  wg: -1 = pair wait, else body index.
__device__ __forceinline__ void e4_yakuman(const E4In& in, uint32_t hf, int head, uint32_t body, int nb, int wg, bool div_valid, int& han, int& ycount,
                                           uint64_t& ym) {
    int yc = 0;
    const bool tsumo = in.cf & CF_TSUMO;
    const bool menzen = e4m_menzen(in.ma);
    if (hf & E4F_TSUUIISOU) { yc += 1; ym |= YB(39); }
    if (hf & E4F_CHINROUTOU) { yc += 1; ym |= YB(41); }
    if (hf & E4F_RYUUIISOU) { yc += 1; ym |= YB(40); }
    if (e4m_kan(in.ma) == 4) { yc += 1; ym |= YB(44); }
    if (menzen && (nb + e4m_n(in.ma)) == 4 && (hf & E4F_CHUUREN)) {
        if (hf & E4F_CHUUREN9) { yc += 2; ym |= YB(47); } else { yc += 1; ym |= YB(45); }
    }
    if ((in.cf & CF_FIRST_TURN) && menzen && tsumo) {
        yc += 1;
        ym |= (in.seat_wind34 == 27) ? YB(35) : YB(36);
    }
    int closed = e4m_ankan(in.ma);
    bool hk = (in.ma.tlo >> 31) & 1u, ht = in.ma.thi & 1u, ck = (in.ma.thi >> 1) & 1u;
    uint32_t wk = e4m_honors(in.ma) & 0xFu;  // winds with a pon / kan
#pragma unroll
    for (int i = 0; i < 4; i++) {
        if (i < nb && b_kou(body, i)) {
            const int t = b_tile(body, i);
            if (!(!tsumo && i == wg)) closed++;
            hk = hk || t == 31;
            ht = ht || t == 32;
            ck = ck || t == 33;
            if (t >= 27 && t <= 30) wk |= 1u << (t - 27);
        }
    }
    if (closed == 4) {
        if (wg < 0) { yc += 2; ym |= YB(48); } else { yc += 1; ym |= YB(38); }
    }
    if (hk && ht && ck) { yc += 1; ym |= YB(37); }
    const int wind_k = __popc(wk);
    const int wind_p = (div_valid && head >= 27 && head <= 30 && !((wk >> (head - 27)) & 1u)) ? 1 : 0;
    if (wind_k == 4) { yc += 2; ym |= YB(50); } else if (wind_k == 3 && wind_p == 1) { yc += 1; ym |= YB(43); }
    if (yc > 0) {
        han = 13 * yc;
        ycount = yc;
    }
}
// yaku.rs:843-890 (+ yaku_3p.rs nukidora)
__device__ __forceinline__ void e4_static_yaku(const E4In& in, int& han, uint64_t& ym) {
    const uint32_t cf = in.cf;
    const bool tsumo = cf & CF_TSUMO;
    if ((cf & CF_RIICHI) && !(cf & CF_DOUBLE_RIICHI)) { han += 1; ym |= YB(2); }
    if (cf & CF_DOUBLE_RIICHI) { han += 2; ym |= YB(18); }
    if (cf & CF_IPPATSU) { han += 1; ym |= YB(30); }
    if (e4m_menzen(in.ma) && tsumo) { han += 1; ym |= YB(1); }
    if ((cf & CF_HAITEI) && tsumo) { han += 1; ym |= YB(5); }
    if ((cf & CF_HOUTEI) && !tsumo) { han += 1; ym |= YB(6); }
    if ((cf & CF_RINSHAN) && tsumo) { han += 1; ym |= YB(4); }
    if ((cf & CF_CHANKAN) && !tsumo) { han += 1; ym |= YB(3); }
    if (in.dora > 0) { han += in.dora; ym |= YB(31); }
    if (in.aka > 0) { han += in.aka; ym |= YB(32); }
    if (in.ura > 0) { han += in.ura; ym |= YB(33); }
    if (in.sanma && in.nuki > 0) { han += in.nuki; ym |= YB(34); }
}

// One (division, winning group) candidate: yaku.rs:298-556 (eval_candidate of rmj_hand.hip.h over the packed melds)
__device__ __forceinline__ void e4_candidate(const E4In& in, uint32_t hf, int head, uint32_t body, int nb, int wg, int& o_han, int& o_fu, int& o_yc,
                                             int& o_kind, uint64_t& o_ym) {
    o_han = 0; o_fu = 0; o_yc = 0; o_kind = 0; o_ym = 0ull;
    e4_yakuman(in, hf, head, body, nb, wg, true, o_han, o_yc, o_ym);
    if (o_han >= 13) {
        o_kind = 1;
        return;
    }
    const bool tsumo = in.cf & CF_TSUMO;
    const bool menzen = e4m_menzen(in.ma);
    const int win = in.win34;
    int han = 0;
    uint64_t ym = 0ull;
    e4_static_yaku(in, han, ym);
    if (hf & E4F_TANYAO) { han += 1; ym |= YB(12); }
    int n_kou = 0, closed = e4m_ankan(in.ma);
    bool any_kou = false;
    uint32_t shun = 0u, kou = 0u;   // type masks (bit t < 27): a shuntsu starting at t / a koutsu of t among the concealed sets
    int fu_body = 0;
    bool junchan = t_is_numterm(head), chanta = t_is_terminal(head);
    bool chanta_honor = head >= 27;
    const uint32_t mh = e4m_honors(in.ma);
    int yk_p = (int)((mh >> 4) & 1u), yk_f = (int)((mh >> 5) & 1u), yk_c = (int)((mh >> 6) & 1u);
    int yk_round = (int)((mh >> (in.round_wind34 - 27)) & 1u), yk_seat = (int)((mh >> (in.seat_wind34 - 27)) & 1u);
    int n_eq = 0;  // number of equal (i < j) shuntsu pairs, for iipeikou / ryanpeikou
#pragma unroll
    for (int i = 0; i < 4; i++) {
        if (i < nb) {
            const int t = b_tile(body, i);
            if (b_kou(body, i)) {
                any_kou = true;
                n_kou++;
                const bool ron_open = (!tsumo && i == wg);
                if (!ron_open) closed++;
                int f = ron_open ? 2 : 4;
                if (t_is_terminal(t)) f *= 2;
                fu_body += f;
                if (t < 27) kou |= 1u << t;
                junchan = junchan && t_is_numterm(t);
                chanta = chanta && t_is_terminal(t);
                chanta_honor = chanta_honor || t >= 27;
                yk_p += (t == 31); yk_f += (t == 32); yk_c += (t == 33);
                yk_round += (t == in.round_wind34); yk_seat += (t == in.seat_wind34);
            } else {
                shun |= 1u << (t < 27 ? t : 31);
                junchan = junchan && (t_is_numterm(t) || t_is_numterm(t + 2));
                chanta = chanta && (t_is_terminal(t) || t_is_terminal(t + 2));
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (j < i && !b_kou(body, j) && b_tile(body, j) == t) n_eq++;
            }
        }
    }
    // melds
    if (e4m_n(in.ma) > 0) {
        junchan = junchan && !(in.ma.chi & E4M_NO_NUMTERM);
        chanta = chanta && !(in.ma.chi & E4M_NO_TERM);
        chanta_honor = chanta_honor || (e4m_types(in.ma) & MASK_HONORS) != 0ull;
    }
    // pinfu (yaku.rs:644-690) or fu (yaku.rs:561-642)
    const bool head_yakuhai = head >= 31 || head == in.round_wind34 || head == in.seat_wind34;
    bool pinfu = false;
    if (menzen && e4m_n(in.ma) == 0 && !any_kou && !head_yakuhai && wg >= 0 && !b_kou(body, wg)) {
        const int t = b_tile(body, wg);
        if (win == t) pinfu = (t % 9) != 6;
        else if (win == t + 2) pinfu = (t % 9) != 0;
    }
    int fu;
    if (pinfu) {
        han += 1; ym |= YB(14);
        fu = tsumo ? 20 : 30;
    } else {
        fu = 20;
        if (tsumo) fu += 2; else if (menzen) fu += 10;
        if (head == in.round_wind34) fu += 2;
        if (head == in.seat_wind34) fu += 2;
        if (head >= 31) fu += 2;
        if (wg < 0) fu += 2;
        else if (!b_kou(body, wg)) {
            const int t = b_tile(body, wg);
            if (win == t + 1 || (win == t + 2 && (t % 9 == 0)) || (win == t && (t % 9 == 6))) fu += 2;
        }
        fu += fu_body + e4m_fu(in.ma);
        if (fu == 20 && !tsumo) fu = 30;
        fu = (fu + 9) / 10 * 10;
    }
    // yakuhai, order P F C round seat (yaku.rs:356-386)
    if (yk_p > 0) { han += yk_p; ym |= YB(7); }
    if (yk_f > 0) { han += yk_f; ym |= YB(8); }
    if (yk_c > 0) { han += yk_c; ym |= YB(9); }
    if (yk_round > 0) { han += yk_round; ym |= YB(11); }
    if (yk_seat > 0) { han += yk_seat; ym |= YB(10); }
    {   // shousangen (yaku.rs:388-424)
        const bool a = yk_p > 0, b = yk_f > 0, c = yk_c > 0;
        const int dk = (int)a + b + c;
        const int dp = (head == 31) + (head == 32) + (head == 33);
        if (!(a && b && c) && dk == 2 && dp == 1) { han += 2; ym |= YB(23); }
    }
    if (n_kou + e4m_nonchi(in.ma) == 4) { han += 2; ym |= YB(21); }
    if (closed == 3) { han += 2; ym |= YB(22); }
    if (e4m_kan(in.ma) == 3) { han += 2; ym |= YB(20); }
    if (menzen) {  // iipeikou / ryanpeikou (yaku.rs:476-503)
        const int pairs = (n_eq == 6 || n_eq == 2) ? 2 : (n_eq >= 1 ? 1 : 0);
        if (pairs == 2) { han += 3; ym |= YB(28); } else if (pairs == 1) { han += 1; ym |= YB(13); }
    }
    {
        const uint32_t s_all = (shun & 0x07FFFFFFu) | (in.ma.chi & 0x07FFFFFFu);
        const uint32_t sm = s_all & 0x1FFu, sp = (s_all >> 9) & 0x1FFu, ss = (s_all >> 18) & 0x1FFu;
        const uint32_t ITT = (1u << 0) | (1u << 3) | (1u << 6);
        if ((sm & ITT) == ITT || (sp & ITT) == ITT || (ss & ITT) == ITT) { han += menzen ? 2 : 1; ym |= YB(16); }
        if ((sm & sp & ss & 0x7Fu) != 0u) { han += menzen ? 2 : 1; ym |= YB(17); }
        const uint32_t k_all = kou | (in.ma.kou & 0x07FFFFFFu);
        const uint32_t km = k_all & 0x1FFu, kp = (k_all >> 9) & 0x1FFu, ks = (k_all >> 18) & 0x1FFu;
        if ((km & kp & ks & 0x1FFu) != 0u) { han += 2; ym |= YB(19); }
    }
    if (hf & E4F_CHINITSU) { han += menzen ? 6 : 5; ym |= YB(29); }
    else if (hf & E4F_HONITSU) { han += menzen ? 3 : 2; ym |= YB(27); }
    if (hf & E4F_HONROUTOU) { han += 2; ym |= YB(24); }
    else if (junchan) { han += menzen ? 3 : 2; ym |= YB(26); }
    else if (chanta && chanta_honor) { han += menzen ? 2 : 1; ym |= YB(15); }
    o_han = han; o_fu = fu; o_yc = 0; o_kind = 0; o_ym = ym;
}

// One decision string of the reference's division search (agari.rs:96-141) over `h` = the hand without its head pair: bit (3 - lvl)
// of `path` is the decision at set lvl, 0 = koutsu (tried first), 1 = shuntsu; a string is canonical when the decisions behind the
// last set are 0.  body: 4 x 8 bit, each (koutsu << 6) | lowest type.
__device__ __forceinline__ bool e4_peel(PH h, int path, uint32_t& body, int& nb) {
    body = 0u;
    nb = 0;
    bool valid = true, done = false;
#pragma unroll
    for (int lvl = 0; lvl < 4; lvl++) {
        const int choice = (path >> (3 - lvl)) & 1;
        if (ph_empty(h)) done = true;
        if (!done && valid) {
            const int i = ph_first(h);
            const int s = t_suit(i);
            const uint32_t w = ph_get(h, s);
            const int sh3 = 3 * (i - 9 * s);
            const uint32_t c = (w >> sh3) & 7u;
            if (choice == 0) {
                if (c >= 3u) {
                    ph_addv(h, s, 0u - (3u << sh3));
                    body |= (uint32_t)((1 << 6) | i) << (8 * nb);
                    nb++;
                } else valid = false;
            } else {
                const bool seq_ok = i < 27 && (i - 9 * s) <= 6;
                if (seq_ok && ((w >> (sh3 + 3)) & 7u) > 0u && ((w >> (sh3 + 6)) & 7u) > 0u) {
                    ph_addv(h, s, 0u - (0111u << sh3));
                    body |= (uint32_t)i << (8 * nb);
                    nb++;
                } else valid = false;
            }
        } else if (done) {
            if (choice != 0) valid = false;
        }
    }
    if (valid && !ph_empty(h)) valid = false;
    return valid;
}

// HandEvaluator::calc core (hand_evaluator.rs:96-175) of the row's hand.  Must be called by every lane of the wave (rows that are
// not `on` idle through it); all inputs row-uniform, so is the result.
__device__ __forceinline__ E4Out e4_calc(const E4In& in, int r, int rb) {
    E4Out out;
    const PH& h = in.hand14;
    const uint32_t hf = e4_hand_flags(in);
    // ---- the standard form: one suit holds 2 mod 3 tiles (the pair's), the others 0 mod 3
    const int fa = field_sum(h.a), fb = field_sum(h.b), fc = field_sum(h.c), fd = field_sum(h.d);
    const int ta = fa % 3, tb = fb % 3, tc = fc % 3, td = fd % 3;
    const int n2 = (ta == 2) + (tb == 2) + (tc == 2) + (td == 2), n0 = (ta == 0) + (tb == 0) + (tc == 0) + (td == 0);
    const bool std_ok = in.on && n2 == 1 && n0 == 3;
    const int sp = ta == 2 ? 0 : (tb == 2 ? 1 : (tc == 2 ? 2 : 3));
    const uint32_t xw = ph_get(h, sp);
    const int j0 = pair_residue(xw);
    const int tot = fa + fb + fc + fd;
    const int nbody = tot >= 14 ? 4 : (tot >= 11 ? 3 : (tot >= 8 ? 2 : (tot >= 5 ? 1 : 0)));   // sets of a complete hand of `tot` tiles
    // ---- divisions: head candidates in ascending order, lane = decision string
    uint32_t vm0 = 0u, vm1 = 0u, vm2 = 0u, body0 = 0u, body1 = 0u, body2 = 0u;
#pragma unroll
    for (int hc = 0; hc < 3; hc++) {
        const int jr = j0 + 3 * hc;
        const bool ok_h = std_ok && jr < (sp == 3 ? 7 : 9) && ((xw >> (3 * jr)) & 7u) >= 2u;
        if (__ballot(ok_h)) {
            uint32_t body = 0u;
            bool valid = false;
            if (ok_h) {
                PH base = h;
                ph_addv(base, sp, 0u - (2u << (3 * jr)));
                int nb_;
                valid = e4_peel(base, r, body, nb_);
            }
            const uint32_t vm = e4_ballot(ok_h && valid, rb);
            if (hc == 0) { vm0 = vm; body0 = body; } else if (hc == 1) { vm1 = vm; body1 = body; } else { vm2 = vm; body2 = body; }
        }
    }
    const int D = __popc(vm0) + __popc(vm1) + __popc(vm2);
    // ---- candidates: lane = (division, winning group), three divisions per pass, in the reference's order (pair first, then the
    //      sets in division order: yaku.rs:275-296)
    uint32_t best_key = 0u;
    int b_han = 0, b_fu = 0, b_yc = 0, b_kind = 0;
    uint32_t b_ylo = 0u, b_yhi = 0u;
    int chc = 0;            // cursor over the division list: head candidate and its valid strings not yet evaluated
    uint32_t cm = vm0;
    if (cm == 0u) { chc = 1; cm = vm1; }
    if (cm == 0u) { chc = 2; cm = vm2; }
    int dbase = 0;
    const int dl = r >= 10 ? 2 : (r >= 5 ? 1 : 0), wgi = r - 5 * dl;
    while (__ballot(in.on && dbase < D)) {
        int lhc = chc;
        uint32_t lm = cm;
#pragma unroll
        for (int s = 0; s < 2; s++) {
            if (s < dl) {
                lm &= lm - 1u;
                if (lm == 0u && lhc < 2) { lhc += 1; lm = lhc == 1 ? vm1 : vm2; }
                if (lm == 0u && lhc < 2) { lhc += 1; lm = vm2; }
            }
        }
        const bool dvalid = in.on && r < 15 && dbase + dl < D && lm != 0u;
        const int path = dvalid ? __ffs((int)lm) - 1 : 0;
        const uint32_t s0 = (uint32_t)e4_bc((int)body0, rb + path), s1 = (uint32_t)e4_bc((int)body1, rb + path), s2 = (uint32_t)e4_bc((int)body2, rb + path);
        const uint32_t body = lhc == 0 ? s0 : (lhc == 1 ? s1 : s2);
        const int head = 9 * sp + j0 + 3 * lhc;
        const int wg = wgi - 1;
        bool hit = dvalid && wg < nbody;
        if (wg < 0) hit = hit && head == in.win34;
        else {
            const int t = b_tile(body, wg & 3);
            hit = hit && (b_kou(body, wg & 3) ? (t == in.win34) : (in.win34 >= t && in.win34 <= t + 2));
        }
        int c_han = 0, c_fu = 0, c_yc = 0, c_kind = 0;
        uint64_t c_ym = 0ull;
        if (__ballot(hit)) {
            if (hit) e4_candidate(in, hf, head, body, nbody, wg, c_han, c_fu, c_yc, c_kind, c_ym);
        }
        const uint32_t key = hit ? (((uint32_t)c_han << 20) | ((uint32_t)c_fu << 8) | (uint32_t)(255 - (5 * (dbase + dl) + wgi))) : 0u;
        const uint32_t mx = e4_rmax(key, rb);
        const uint32_t wb = e4_ballot(hit && key == mx, rb);
        const int wl = rb + (wb ? __ffs((int)wb) - 1 : 0);
        const int w_han = e4_bc(c_han, wl), w_fu = e4_bc(c_fu, wl), w_yc = e4_bc(c_yc, wl), w_kind = e4_bc(c_kind, wl);
        const uint32_t w_ylo = (uint32_t)e4_bc((int)(uint32_t)c_ym, wl), w_yhi = (uint32_t)e4_bc((int)(uint32_t)(c_ym >> 32), wl);
        if (mx > best_key) {
            best_key = mx;
            b_han = w_han; b_fu = w_fu; b_yc = w_yc; b_kind = w_kind; b_ylo = w_ylo; b_yhi = w_yhi;
        }
        // the cursor moves on by three divisions
#pragma unroll
        for (int s = 0; s < 3; s++) {
            cm &= cm - 1u;
            if (cm == 0u && chc < 2) { chc += 1; cm = chc == 1 ? vm1 : vm2; }
            if (cm == 0u && chc < 2) { chc += 1; cm = vm2; }
        }
        dbase += 3;
    }
    int han, fu, yc, kind;
    uint64_t ym;
    out.shape = true;
    if (D > 0) {
        han = b_han; fu = b_fu; yc = b_yc; kind = b_kind;
        ym = ((uint64_t)b_yhi << 32) | b_ylo;
        if ((best_key >> 8) == 0u) { han = 0; fu = 0; yc = 0; kind = 0; ym = 0ull; }  // no candidate contained the win tile
    } else if (is_kokushi(h)) {  // yaku.rs:237-255
        kind = 3; fu = 0;
        if (ph_cnt(h, in.win34) == 2) { han = 26; yc = 2; ym = YB(49); }
        else { han = 13; yc = 1; ym = YB(42); }
    } else if (is_chiitoi(h)) {  // yaku.rs:256-294 (quirk Q4: yakuman overwrites han, static added on top)
        kind = 2; fu = 25; han = 2; yc = 0; ym = YB(25);
        if (hf & E4F_TANYAO) { han += 1; ym |= YB(12); }
        if (hf & E4F_CHINITSU) { han += 6; ym |= YB(29); } else if (hf & E4F_HONITSU) { han += 3; ym |= YB(27); }
        if (hf & E4F_HONROUTOU) { han += 2; ym |= YB(24); }
        e4_yakuman(in, hf, 0, 0u, 0, -1, false, han, yc, ym);
        e4_static_yaku(in, han, ym);
    } else {
        out.shape = false;
        kind = 0; han = 0; fu = 0; yc = 0; ym = 0ull;
    }
    han &= 0xFF;  // YakuResult.han is u8 in the reference
    out.han = han; out.fu = fu; out.yakuman_count = yc; out.kind = kind; out.ym = ym;
    const bool is_oya = in.seat_wind34 == 27;
    const uint32_t scoring_han = (yc == 0 && han >= 13) ? 13u : (uint32_t)han;  // hand_evaluator.rs:144-148
    const ScoreOut sc = calc_score(scoring_han, (uint32_t)fu, is_oya, in.cf & CF_TSUMO, in.honba, in.sanma ? 3u : 4u);
    const bool has_yaku = (ym & ~YMASK_DORA) != 0ull;
    out.is_win = out.shape && (has_yaku || yc > 0) && han >= 1;
    out.yakuman = yc > 0;
    out.ron = sc.ron; out.tsumo_oya = sc.tsumo_oya; out.tsumo_ko = sc.tsumo_ko;
    if (!out.shape) { out.ron = 0u; out.tsumo_oya = 0u; out.tsumo_ko = 0u; }
    return out;
}

// The ordered yaku list (see yaku_list) by lanes: lane = position in the emission order of the result's kind, three positions per
// lane; a present id lands at the count of present ids before it.  dst: 20 bytes in LDS (zeroed by the caller).  Returns the
// list length (capped at 20).
__device__ __constant__ const uint8_t E4_ORDER[4][48] = {
    {2, 18, 30, 1, 5, 6, 4, 3, 31, 32, 33, 34, 12, 14, 7, 8, 9, 11, 10, 23, 21, 22, 20, 28, 13, 16, 17, 19, 29, 27, 24, 26, 15,
     255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255},
    {39, 41, 40, 44, 47, 45, 35, 36, 48, 38, 37, 50, 43, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255,
     255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255},
    {25, 12, 29, 27, 24, 39, 41, 40, 44, 47, 45, 35, 36, 48, 38, 37, 50, 43, 2, 18, 30, 1, 5, 6, 4, 3, 31, 32, 33, 34, 255, 255,
     255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255},
    {49, 42, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255,
     255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255, 255}};
__device__ __forceinline__ int e4_yaku_list(int kind, uint64_t ym, uint8_t* dst, int r, int rb) {
    int n = 0;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const uint32_t id = E4_ORDER[kind & 3][r + 16 * j];
        const bool present = id != 255u && ((ym >> (id & 63u)) & 1ull);
        const uint32_t b = e4_ballot(present, rb);
        const int pos = n + __popc(b & ((1u << r) - 1u));
        if (present && pos < 20) dst[pos] = (uint8_t)id;
        n += __popc(b);
    }
    return n < 20 ? n : 20;
}

}  // namespace rmj
