// One-wavefront-per-game state machine for gfx950.
//
// All 64 lanes of a wave own ONE game whose 640-byte record sits in LDS for the duration of
// the kernel.  Control flow is wave-uniform (every lane walks the same transition); the lanes
// split only inside the data-parallel helpers:
//   * waits / tenpai probes: lane = candidate tile (34 lanes) + __ballot
//   * legal-action enumeration (discards, pon pairs, chi combos, kakan): lane = candidate,
//     __ballot + prefix popcount gives each surviving candidate its slot in the ORDERED list
//   * hand sort / wall permutation: rank-by-counting, lane = element
//   * yaku: lane = candidate head (rmj_hand.hip.h)
//
// Reference semantics: riichienv-core/src/state/mod.rs (step, _resolve_discard, _resolve_kan,
// _accept_riichi, _deal_next, _initialize_next_round, _initialize_round, _trigger_ryukyoku,
// check_abortive_draw, _reveal_kan_dora), state/legal_actions.rs, riichienv-python/src/env.rs.
// NOTE: this file is included ONCE PER VARIANT by rmj_api.hip with
//   RMJ_NS    = namespace of the instantiation (rmj4 / rmj3)
//   RMJ_SANMA = 0 (state/, 4 seats, 136 tiles) or 1 (state_3p/, 3 seats, 108 tiles, kita, no chi)
// so that the seat count and the sanma switches are compile-time constants on each variant's hot path.
#include "rmj_common.hip.h"

namespace RMJ_NS {
using namespace rmj;
constexpr bool KSANMA = (RMJ_SANMA != 0);  // game_mode >= 3 (game_variant.rs:12-37)
constexpr int KNP = KSANMA ? 3 : 4;        // seats in play; seat 3 is inert in 3P

struct Ctx {
    GState& S;
    CEnv& E;
    WaveScratch& X;
    uint32_t g;
    int lane;
    uint8_t* W;       // this game's wall (global)
    uint64_t* Lg;     // this game's legal lists [4][64] in HBM (read for validation, written by finalize)
    // the next live-wall draw W[live_end-1], fetched as soon as the record is in LDS so that its latency is off the
    // critical path of deal_next (valid while live_end == pf_live_end; -1: not fetched)
    int pf_draw = -1, pf_live_end = -1;
    // Fast/slow split (k_step): the <FAST> instantiations of the transition functions cover the common transitions and
    // set `bail` as soon as they meet anything rare (yaku evaluation, settlement, kan, ryukyoku, next round, ...); the
    // kernel then re-runs the whole step from the untouched HBM record in the out-of-line full-featured path.
    bool bail = false;
    // >= 0: events are staged in X.evbuf (count so far), flushed by k_step once the fast path has succeeded; -1: every
    // event is stored to the ring at once (all other kernels, the full path, out-of-line bodies)
    int ev_stage = -1;
    // PState quarters of the record that this step may have modified (bit = seat; the 128 B of globals always are): the
    // fast path of k_step stores only those back.  0xF everywhere it is not tracked.
    uint32_t dirty = 0xFu;
};

// By-value view of a Ctx for out-of-line (rare-path) functions.  Passing Ctx& to a non-inlined function would
// make the caller's Ctx escape to scratch and turn every LDS access of the hot path into a flat access; passing
// the members by value keeps the caller's Ctx in registers (address spaces stay inferable).
struct CtxV {
    GState* S;
    WaveScratch* X;
    uint8_t* W;
    uint64_t* Lg;
    const Env* E;
    uint32_t g;
    int lane;
};
// Function arguments arrive in vector registers (the call ABI treats them as divergent); everything in a CtxV except
// `lane` is wave-uniform, so it is moved to scalar registers on entry: the out-of-line bodies then address memory and
// branch with scalar code like the inlined hot path does.
template <typename T>
__device__ __forceinline__ T* uni_ptr(T* p) {
    uint64_t x = (uint64_t)p;
    uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)x), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(x >> 32));
    return (T*)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ uint32_t uni(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
__device__ __forceinline__ uint64_t uni(uint64_t x) {
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(x >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)x);
}
// U(x): a value every lane of the wave holds alike (anything read from the record in LDS), moved to a scalar register.  The
// compiler cannot prove LDS loads uniform and would lower a branch on them with exec-mask save / restore instructions;
// a branch on U(x) is a plain scalar compare-and-branch (the step is bound by VALU + SALU issue).
template <typename T>
__device__ __forceinline__ T U(T x) { return (T)__builtin_amdgcn_readfirstlane((int)x); }
#undef CTX_FROM
#define CTX_FROM(v) Ctx c{*uni_ptr((v).S), *(CEnv*)uni_ptr((v).E), *uni_ptr((v).X), uni((v).g), (v).lane, uni_ptr((v).W), uni_ptr((v).Lg)}
__device__ __forceinline__ CtxV ctx_pack(const Ctx& c) {
    CtxV v;
    v.S = &c.S; v.X = &c.X; v.W = c.W; v.Lg = c.Lg; v.E = (const Env*)&c.E; v.g = c.g; v.lane = c.lane;
    return v;
}

// ---------------------------------------------------------------- events
// 8 wave-uniform dwords -> one 32-byte record (two 16-byte stores from lane 0; no stack object, no scratch)
__device__ __forceinline__ void emit_words(Ctx& c, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t w4, uint32_t w5,
                                           uint32_t w6, uint32_t w7) {
    if (c.E.skip_log) return;
    const uint32_t evc = (uint32_t)__builtin_amdgcn_readfirstlane((int)c.S.ev_count);  // wave-uniform: scalar address math
    const uint32_t idx = evc & c.E.ring_mask;
    c.S.ev_count = evc + 1;
    w7 = (w7 & 0x00FFFFFFu) | ((uint32_t)KNP << 24);  // pad byte = seats (formatter)
    if (c.ev_stage >= 0 && c.ev_stage < RMJ_EV_STAGE) {
        if (c.lane == 0) {
            uint4* b = reinterpret_cast<uint4*>(c.X.evbuf[c.ev_stage]);
            b[0] = make_uint4(w0, w1, w2, w3);
            b[1] = make_uint4(w4, w5, w6, w7);
            c.X.evidx[c.ev_stage] = idx;
        }
        c.ev_stage += 1;
        return;
    }
    uint4* dst = reinterpret_cast<uint4*>(c.E.events + (size_t)c.g * (c.E.ring_mask + 1u) + idx);
    if (c.lane == 0) {
        dst[0] = make_uint4(w0, w1, w2, w3);
        dst[1] = make_uint4(w4, w5, w6, w7);
    }
}
// staged events -> ring: lane = 2 * event + half, one 16-byte store per lane
__device__ __forceinline__ void flush_events(Ctx& c) {
    const int n = c.ev_stage;
    if (n > 0 && c.lane < 2 * n) {
        const int e = c.lane >> 1, h = c.lane & 1;
        uint4* dst = reinterpret_cast<uint4*>(c.E.events + (size_t)c.g * (c.E.ring_mask + 1u) + c.X.evidx[e]);
        dst[h] = reinterpret_cast<const uint4*>(c.X.evbuf[e])[h];
    }
}
__device__ __forceinline__ void emit_raw(Ctx& c, const RmjEvent& ev) {  // rare paths (struct built on the stack)
    uint32_t p[8];
    __builtin_memcpy(p, &ev, 32);  // (no type punning through a uint32_t*: strict aliasing)
    emit_words(c, p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7]);
}
__device__ inline RmjEvent ev_zero(uint8_t type) {
    RmjEvent e;
    __builtin_memset(&e, 0, sizeof(e));
    e.type = type;
    return e;
}
__device__ __forceinline__ void emit_simple(Ctx& c, uint8_t type, uint8_t actor = 0, uint8_t tile = 0, uint8_t flags = 0) {
    emit_words(c, (uint32_t)type | ((uint32_t)actor << 8) | ((uint32_t)tile << 24), 0, 0, 0, 0, 0, (uint32_t)flags, 0);
}
__device__ __forceinline__ void emit_meld(Ctx& c, uint8_t type, uint8_t actor, uint8_t target, uint8_t tile, uint64_t act) {
    uint32_t n = a_n(act);
    uint32_t cons = 0;
#pragma unroll
    for (int i = 0; i < 4; i++)
        if ((uint32_t)i < n) cons |= a_c(act, i) << (8 * i);
    emit_words(c, (uint32_t)type | ((uint32_t)actor << 8) | ((uint32_t)target << 16) | ((uint32_t)tile << 24), cons, 0, 0, 0, 0,
               (n << 4) & 0xFFu, 0);
}

__device__ __forceinline__ bool rule(const Ctx& c, uint32_t bit) { return (c.E.rule_bits & bit) != 0; }
// rank-sort the first n tiles of a hand, optionally dropping hand[drop] (hand.remove(idx); hand.sort(), state/mod.rs
// discard path).  Tiles live in registers (lane = slot) and are compared through v_readlane: no LDS round trips.
__device__ __forceinline__ void sort_hand(Ctx& c, PState& P, int n, int drop = -1) {
    const bool mine = c.lane < n && c.lane != drop;
    const int t = mine ? (int)P.hand[c.lane] : 0xFFFF;
    int r = 0;
#pragma unroll
    for (int k = 0; k < 14; k++) {  // stable rank: poked test states may repeat an id
        int tk = __builtin_amdgcn_readlane(t, k);
        r += (tk < t) || (tk == t && k < c.lane);
    }
    if (mine) P.hand[r] = (uint8_t)t;
    if (drop >= 0 && drop < n) P.hand_len = (uint8_t)(n - 1);
    wave_sync();
}
// hand.remove(drop); hand.sort() of the discard path when the hand is "n-1 sorted tiles + the drawn tile last" (always,
// unless a test poked an unsorted hand: checked, general sort then): every tile knows its new slot from the position of
// the removed tile and one comparison with the drawn tile; the drawn tile's slot is one ballot.  Same stable order as
// sort_hand (equal ids keep their relative order).
__device__ __forceinline__ void discard_sort(Ctx& c, PState& P, int n, int drop) {
    const int lane = c.lane;
    const int t = lane < n ? (int)P.hand[lane] : 0xFFFF;
    const int nxt = __builtin_amdgcn_update_dpp(0xFFFF, t, 0x101 /* row_shl:1 */, 0xf, 0xf, false);
    if (__ballot(lane < n - 2 && t > nxt)) {  // prefix not sorted
        sort_hand(c, P, n, drop);
        return;
    }
    const int d = __builtin_amdgcn_readlane(t, n - 1);
    const int before_d = __popcll(__ballot(lane < n - 1 && lane != drop && t <= d));
    int r = lane - (drop < lane ? 1 : 0) + ((drop != n - 1 && d < t) ? 1 : 0);
    if (lane == n - 1) r = before_d;
    if (lane < n && lane != drop) P.hand[r] = (uint8_t)t;
    P.hand_len = (uint8_t)(n - 1);
    wave_sync();
}
// remove hand[idx] keeping order
__device__ __forceinline__ void hand_remove_at(Ctx& c, PState& P, int idx) {
    int n = P.hand_len;
    int t = 0;
    if (c.lane > idx && c.lane < n) t = P.hand[c.lane];
    if (c.lane > idx && c.lane < n) P.hand[c.lane - 1] = (uint8_t)t;
    P.hand_len = (uint8_t)(n - 1);
    wave_sync();
}
__device__ __forceinline__ int hand_find(const Ctx& c, const PState& P, int tile) {  // position() of a 136-id, -1 if absent
    bool hit = c.lane < U((int)P.hand_len) && P.hand[c.lane] == tile;
    uint64_t b = __ballot(hit);
    return b ? (__ffsll((long long)b) - 1) : -1;
}

// ---------------------------------------------------------------- calc glue (HandEvaluator::new + calc)
struct Cond {
    uint32_t cf;
    uint32_t honba;
};
// seat's concealed tiles = hand minus `skip_idx` (-1: none); win tile added iff total == 13
__device__ __forceinline__ CalcOut seat_calc_impl(Ctx& c, int seat, int skip_idx, int win_tile, uint32_t cf, uint32_t honba, bool use_ura,
                                                  int kita_count) {
    GState& S = c.S;
    PState& P = S.p[seat];
    CalcIn in;
    in.ma = build_meld_agg(P);
    PH h = build_ph_wave(P, c.lane, skip_idx);
    int aka = in.ma.aka;
    for (int j = 0; j < P.hand_len; j++)
        if (j != skip_idx) aka += is_aka(P.hand[j]);
    int total = ph_total(h) + 3 * in.ma.n;
    int win34 = win_tile >> 2;
    if (total == 13) {
        ph_add(h, win34);
        aka += is_aka(win_tile);
    }
    CalcOut out;
    out.shape = false; out.is_win = false; out.yakuman = false; out.han = 0; out.fu = 0; out.yakuman_count = 0; out.kind = 0; out.ym = 0;
    out.ron = out.tsumo_oya = out.tsumo_ko = 0;
    if (!is_agari(h)) return out;
    // full histogram for dora counting
    PH full = h;
    for (int i = 0; i < in.ma.n; i++) {
        int nt = (P.meld_type[i] >= RMJ_MELD_DAIMINKAN) ? 4 : 3;
        for (int k = 0; k < nt; k++) ph_add(full, P.meld_tiles[i][k] >> 2);
    }
    int dora = 0, ura = 0;
    const bool sanma = KSANMA;
    for (int k = 0; k < S.n_dora; k++) {
        int nt = next_dora34(S.dora[k] >> 2, sanma);
        dora += ph_cnt(full, nt);
        if (sanma && nt == 30) dora += kita_count;  // hand_evaluator_3p.rs:110-116
    }
    if (use_ura)
        for (int k = 0; k < S.n_dora; k++) {  // _get_ura_indicators, state/mod.rs:2048-2057 ; 3P: pre-extracted W[9+2k]
            int idx = sanma ? 9 + 2 * k : 5 + 2 * k;
            if (sanma || idx < S.live_end) {
                int nt = next_dora34(c.W[idx] >> 2, sanma);
                ura += ph_cnt(full, nt);
                if (sanma && nt == 30) ura += kita_count;
            }
        }
    // the evaluation itself in row form (e4_calc): every row of the wave holds the same hand here, so all rows compute - and return -
    // the same result; its ~10 x shorter instruction stream is what matters to the wave that carries a Ron / Tsumo through the full path
    E4In e;
    e.on = true;
    e.hand14 = h;
    {   // the meld aggregate as packed words (MeldAgg -> E4Meld)
        E4Meld m = {0u, 0u, 0u, 0u, 0u};
        for (int i = 0; i < in.ma.n; i++) {
            const int t0 = in.ma.t0[i];
            const bool chi = in.ma.mtype[i] == RMJ_MELD_CHI;
            const uint32_t bit = t0 < 27 ? 1u << t0 : 0u;
            if (chi) m.chi |= bit; else m.kou |= bit;
            if (!(in.ma.mtypes[i] & MASK_NUMTERM)) m.chi |= E4M_NO_NUMTERM;
            if (!(in.ma.mtypes[i] & MASK_TERM)) m.chi |= E4M_NO_TERM;
            if (!chi && t0 >= 27 && t0 < 34) m.thi |= 1u << (8 + t0 - 27);
        }
        if (!in.ma.menzen) m.chi |= E4M_OPENED;
        m.tlo = (uint32_t)in.ma.types;
        m.thi |= (uint32_t)(in.ma.types >> 32) & 3u;
        m.cnt = (uint32_t)in.ma.n | ((uint32_t)in.ma.n_kan << 3) | ((uint32_t)in.ma.n_ankan << 6) | ((uint32_t)in.ma.n_nonchi << 9) |
                (((uint32_t)in.ma.fu & 0xFFu) << 12) | (((uint32_t)in.ma.aka & 0xFu) << 20);
        e.ma = m;
    }
    e.win34 = win34;
    e.cf = cf;
    e.dora = dora & 0xFF; e.aka = aka; e.ura = ura & 0xFF; e.nuki = sanma ? kita_count : 0;
    e.round_wind34 = 27 + (S.round_wind & 3);
    e.seat_wind34 = 27 + ((seat + KNP - S.oya) % KNP);
    e.sanma = sanma;
    e.honba = honba;
    const E4Out o = e4_calc(e, c.lane & 15, c.lane & 48);
    out.shape = o.shape; out.is_win = o.is_win; out.yakuman = o.yakuman; out.han = o.han; out.fu = o.fu; out.yakuman_count = o.yakuman_count;
    out.kind = o.kind; out.ym = o.ym; out.ron = o.ron; out.tsumo_oya = o.tsumo_oya; out.tsumo_ko = o.tsumo_ko;
    return out;
}
__device__ __noinline__ CalcOut ol_seat_calc(CtxV v, int seat, int skip_idx, int win_tile, uint32_t cf, uint32_t honba, bool use_ura,
                                             int kita_count) {
    CTX_FROM(v);
    return seat_calc_impl(c, uni(seat), uni(skip_idx), uni(win_tile), uni(cf), uni(honba), use_ura, uni(kita_count));
}
// kita_count: Conditions.kita_count — the reference passes it only at settlement and in the kita ron check
// (state_3p/mod.rs:635,926; sanma.rs:115), legality checks leave it 0.
__device__ __forceinline__ CalcOut seat_calc(Ctx& c, int seat, int skip_idx, int win_tile, uint32_t cf, uint32_t honba, bool use_ura,
                                             int kita_count = 0) {
    return ol_seat_calc(ctx_pack(c), seat, skip_idx, win_tile, cf, honba, use_ura, kita_count);
}
__device__ inline uint32_t base_cf(const PState& P) {
    uint32_t cf = 0;
    if (P.flags & PF_RIICHI_DECLARED) cf |= CF_RIICHI;
    if (P.flags & PF_DOUBLE_RIICHI) cf |= CF_DOUBLE_RIICHI;
    if (P.flags & PF_IPPATSU) cf |= CF_IPPATSU;
    return cf;
}

// Number of ISOLATED tiles of a wave-uniform hand: a kind held exactly once with nothing within two ranks in its suit
// (honors: held exactly once).  Such a tile belongs to no set, pair or taatsu.  lane = tile type.
__device__ __forceinline__ int isolated_tiles(const PH& h, int lane) {
    const int s = lane < 34 ? t_suit(lane) : 3, r = lane - 9 * s;
    const uint32_t x = ph_get(h, s);
    const uint32_t nz = (x | (x >> 1) | (x >> 2)) & O9_1;        // bit 3j set iff rank j is held
    const uint32_t win = (nz << 6) >> (3 * r);                    // rank r-2 -> bit 0, r-1 -> 3, r -> 6, r+1 -> 9, r+2 -> 12
    const bool alone = s == 3 || (win & 0x1209u) == 0u;
    return __popcll(__ballot(lane < 34 && ((x >> (3 * r)) & 7u) == 1u && alone));
}
// Fill the seat's wait cache from the histogram of its 13 tiles.  The table shanten goes first: a hand with a wait has
// shanten 0, so shanten > 0 means "no waits" without running the probe (most hands, most of the time), and the number
// itself is kept: one draw lowers it by at most one, which lets the riichi probe skip hands that were >= 2 away.
// In front of the tables sits a pure-ALU bound: four isolated tiles leave at most nine tiles for blocks, i.e. standard
// shanten >= 8 - 2*3 = 2, chiitoi shanten >= 6 - 4 = 2 (at most four pairs), and kokushi is excluded by its kind count -
// most hands of a random rollout stop there, without any table access.
// (4P tables also for a sanma hand: it has no 2m-8m, for which the 4P number is a lower bound of the 3P one.)
__device__ __forceinline__ uint64_t fill_waits13(Ctx& c, PState& P, const PH& h13) {
    int sh;
    uint64_t W = 0ull;
    const uint32_t T9 = 1u | (1u << 24);
    const uint32_t ha = U(h13.a), hb = U(h13.b), hc = U(h13.c), hd = U(h13.d);  // wave-uniform histogram words
    const int yaochu_kinds = __popc((ha | (ha >> 1) | (ha >> 2)) & T9) + __popc((hb | (hb >> 1) | (hb >> 2)) & T9) +
                             __popc((hc | (hc >> 1) | (hc >> 2)) & T9) + __popc((hd | (hd >> 1) | (hd >> 2)) & O7_1);
    // iso isolated tiles leave r = tiles - iso for blocks: 2 * mentsu + taatsu <= {6, 5, 4} for iso = {4, 5, >= 6} whatever
    // the number of melds, i.e. standard shanten >= {2, 3, 4}; a closed hand also has the exact chiitoi number (6 - pairs
    // + missing kinds) and a kokushi bound (13 - kinds - 1).  The bound is worth keeping as large as it is: every later
    // tedashi lowers it by one instead of recomputing (resolve_discard).
    const int iso = isolated_tiles(h13, c.lane);
    int lb = iso >= 6 ? 4 : (iso == 5 ? 3 : (iso == 4 ? 2 : 0));
    const int len3 = U((int)P.hand_len) / 3;
    if (len3 == 4) {
        const int koku = 12 - yaochu_kinds;
        lb = koku < lb ? koku : lb;
        if (lb > 2) {
            const int chi = sh_chiitoi(h13, false);
            lb = chi < lb ? chi : lb;
        }
    }
    if (lb >= 2) {
        sh = lb;  // a lower bound is all the users of sh13 need
    } else {
        sh = sh_shanten_wave(h13, len3, sh_tables_of(c.E), c.lane);
        if (sh <= 0) W = wave_waits(h13, c.lane);
    }
    P.waits13 = W;
    P.sh13 = (uint8_t)(sh < 0 ? 0 : sh);
    P.flags |= PF_WAITS_VALID;
    return W;
}
// cached get_waits of a seat's 13-tile hand (hand_evaluator.rs:196-213); 0 when the seat holds 14
__device__ __forceinline__ uint64_t seat_waits(Ctx& c, int seat) {
    PState& P = c.S.p[seat];
    if (P.hand_len + 3 * P.n_melds != 13) return 0ull;
    if (P.flags & PF_WAITS_VALID) return P.waits13;
    return fill_waits13(c, P, build_ph_wave(P, c.lane));
}
__device__ __forceinline__ void waits_invalidate(PState& P) { P.flags &= ~PF_WAITS_VALID; }
// cheap in-line win-shape probe so that the (large, out-of-line) yaku evaluation is entered only for complete hands
__device__ __noinline__ bool ol_is_agari(PH h) { return is_agari(h); }  // rare fallback: keep the big body out of line
__device__ __forceinline__ bool seat_shape(Ctx& c, int seat, int skip_idx, int win_tile) {
    PState& P = c.S.p[seat];
    PH h = build_ph_wave(P, c.lane, skip_idx);
    if (ph_total(h) + 3 * P.n_melds == 13) ph_add(h, win_tile >> 2);
    return ol_is_agari(h);
}

// ---------------------------------------------------------------- legal actions
__device__ __forceinline__ void put_legal(Ctx& c, int seat, int pos, uint64_t a) {
    if (c.lane == 0 && pos < RMJ_MAX_LEGAL) c.X.legal[seat][pos] = a;
}

// legal_actions.rs:254-508.  Writes the claim list (+Pass) for seat i; returns true iff seat i has claims.
__device__ __forceinline__ bool gen_claims(Ctx& c, int i, int pid, int tile, bool mark_missed = true) {
    GState& S = c.S;
    PState& P = S.p[i];
    const int lane = c.lane;
    const int tt = tile >> 2;
    const int hl = P.hand_len;
    int n = 0;
    uint64_t W = seat_waits(c, i);
    c.X.wout[i] = W;
    bool in_discards = (P.discard_type_mask >> tt) & 1ull;
    bool in_missed = (P.flags & PF_MISSED_DOUJUN) || ((P.flags & PF_RIICHI_DECLARED) && (P.flags & PF_MISSED_RIICHI));
    if (!in_discards && !in_missed) {
        bool furiten = (W & P.discard_type_mask) != 0ull || (P.flags & (PF_MISSED_RIICHI | PF_MISSED_DOUJUN));
        if (!furiten && ((W >> tt) & 1ull)) {
            uint32_t cf = base_cf(P);
            if (S.drawable_count == 0 && !S.is_rinshan) cf |= CF_HOUTEI;
            CalcOut r = seat_calc(c, i, -1, tile, cf, S.honba, false);
            if (r.is_win) {
                put_legal(c, i, n++, mk_action(RMJ_RON, tile, 0));
                S.ron_offer_mask |= (uint8_t)(1u << i);
            } else if (r.shape && mark_missed) {
                P.flags |= PF_MISSED_DOUJUN;  // state/mod.rs:1386-1389
            }
        }
    }
    const bool riichi = P.flags & PF_RIICHI_DECLARED;
    const bool kuikae = rule(c, RMJ_RULE_KUIKAE_FORBIDDEN);
    uint8_t ht = lane < hl ? P.hand[lane] : 0xFF;
    int hty = ht >> 2;
    if (!riichi && S.drawable_count > 0) {
        uint64_t mb = __ballot(lane < hl && hty == tt);
        int count = __popcll(mb);
        if (count >= 2 && hl >= 3) {
            bool ok = kuikae ? (hl - count) > 0 : (hl - 2) > 0;
            if (ok) {
                int i0 = __ffsll((long long)mb) - 1;
                uint64_t m1 = mb & (mb - 1);
                int i1 = __ffsll((long long)m1) - 1;
                put_legal(c, i, n++, mk_action(RMJ_PON, tile, 2, P.hand[i0], P.hand[i1]));
                if (count >= 3) {
                    uint64_t m2 = m1 & (m1 - 1);
                    int i2 = __ffsll((long long)m2) - 1;
                    put_legal(c, i, n++, mk_action(RMJ_PON, tile, 2, P.hand[i0], P.hand[i2]));
                    put_legal(c, i, n++, mk_action(RMJ_PON, tile, 2, P.hand[i1], P.hand[i2]));
                }
            }
        }
        if (count >= 3) {
            int i0 = __ffsll((long long)mb) - 1;
            uint64_t m1 = mb & (mb - 1);
            int i1 = __ffsll((long long)m1) - 1;
            uint64_t m2 = m1 & (m1 - 1);
            int i2 = __ffsll((long long)m2) - 1;
            put_legal(c, i, n++, mk_action(RMJ_DAIMINKAN, tile, 3, P.hand[i0], P.hand[i1], P.hand[i2]));
        }
    }
    // Chi: lane = pattern*16 + a*4 + b
    bool shimocha = !KSANMA && i == ((pid + 1) & 3);  // no Chi in 3P (state_3p/legal_actions.rs:386)
    if (!riichi && S.drawable_count > 0 && shimocha && hl >= 3 && tt < 27) {
        int r9 = tt % 9;
        int k = lane >> 4, a = (lane >> 2) & 3, b = lane & 3;
        // per-pattern type pair
        int ta = k == 0 ? tt - 2 : (k == 1 ? tt - 1 : tt + 1);
        int tb = k == 0 ? tt - 1 : (k == 1 ? tt + 1 : tt + 2);
        bool pat_ok = k == 0 ? (r9 >= 2) : (k == 1 ? (r9 >= 1 && r9 <= 7) : (k == 2 ? (r9 <= 6) : false));
        // ballots must be executed by all lanes: build the six type masks uniformly
        uint64_t m_m2 = __ballot(lane < hl && hty == tt - 2), m_m1 = __ballot(lane < hl && hty == tt - 1);
        uint64_t m_p1 = __ballot(lane < hl && hty == tt + 1), m_p2 = __ballot(lane < hl && hty == tt + 2);
        uint64_t m_0 = __ballot(lane < hl && hty == tt);
        uint64_t m_p3 = __ballot(lane < hl && hty == tt + 3), m_m3 = __ballot(lane < hl && hty == tt - 3);
        uint64_t ma = k == 0 ? m_m2 : (k == 1 ? m_m1 : m_p1);
        uint64_t mbb = k == 0 ? m_m1 : (k == 1 ? m_p1 : m_p2);
        // kuikae: forbidden = {tt} (+ tt+3 for pattern 2 if r9<=5, + tt-3 for pattern 0 if r9>=3)
        int forb = __popcll(m_0);
        if (k == 2 && r9 <= 5) forb += __popcll(m_p3);
        if (k == 0 && r9 >= 3) forb += __popcll(m_m3);
        bool kk_ok = kuikae ? (hl - 2 - forb) > 0 : (hl - 2) > 0;
        bool valid = lane < 48 && pat_ok && a < __popcll(ma) && b < __popcll(mbb) && kk_ok;
        uint64_t act = 0;
        if (valid) {
            uint64_t x = ma;
            for (int q = 0; q < a; q++) x &= x - 1;
            int ia = __ffsll((long long)x) - 1;
            uint64_t y = mbb;
            for (int q = 0; q < b; q++) y &= y - 1;
            int ib = __ffsll((long long)y) - 1;
            act = mk_action(RMJ_CHI, tile, 2, P.hand[ia], P.hand[ib]);
        }
        (void)ta; (void)tb;
        uint64_t vb = __ballot(valid);
        if (valid) {
            int pos = n + __popcll(vb & lanemask_lt(lane));
            if (pos < RMJ_MAX_LEGAL) c.X.legal[i][pos] = act;
        }
        n += __popcll(vb);
    }
    S.stale_n[i] = (uint8_t)(n > 62 ? 62 : n);
    if (n > 0) {
        put_legal(c, i, n, mk_action(RMJ_PASS, RMJ_TILE_NONE, 0));
        c.X.nl[i] = n + 1;
        return true;
    }
    c.X.nl[i] = 0;
    return false;
}

// bit j set iff HandEvaluator(hand minus hand[j]).is_tenpai()  (legal_actions.rs:77-131)
// Claim generation of _resolve_discard for ALL seats at once (legal_actions.rs:254-508; same lists, same order as three
// gen_claims calls).  The common case - nobody can claim - costs a handful of seat-parallel instructions:
//   A  lane = seat: refill stale wait caches (rare loop), B  lane = seat: ron eligibility, yaku only for candidates,
//   C  lane = 16*seat + hand slot: one ballot finds the pon/daiminkan material of every seat,
//   D  chi for the next seat only, E  lane = seat: Pass / list lengths / stale counts.
// Per-seat list positions run in LDS (X.nl), so the emission code exists once with a dynamic seat index.
template <bool FAST = false>
__device__ __forceinline__ uint32_t gen_claims_all(Ctx& c, int pid, int tile) {
    GState& S = c.S;
    const int lane = c.lane;
    pid = U(pid); tile = U(tile);
    const int tt = tile >> 2;
    if (lane < 4) c.X.nl[lane] = 0;
    // ---- A/B (lane = seat)
    const PState& Q = S.p[lane & 3];
    const bool other = lane < KNP && lane != pid;
    const uint32_t qfl = Q.flags;
    const bool holds13 = other && (Q.hand_len + 3 * Q.n_melds == 13);
    uint32_t need = (uint32_t)__ballot(holds13 && !(qfl & PF_WAITS_VALID)) & 0xFu;
    c.dirty |= need;
    while (need) {
        int i = __ffs((int)need) - 1;
        need &= need - 1u;
        PState& P = S.p[i];
        fill_waits13(c, P, build_ph_wave(P, lane));
    }
    wave_sync();
    PROF(c.X, lane, 16);
    const uint64_t W = holds13 ? Q.waits13 : 0ull;
    if (lane < 4) c.X.wout[lane] = W;
    const uint64_t dtm = Q.discard_type_mask;
    const bool in_discards = (dtm >> tt) & 1ull;
    const bool in_missed = (qfl & PF_MISSED_DOUJUN) || ((qfl & PF_RIICHI_DECLARED) && (qfl & PF_MISSED_RIICHI));
    const bool furiten = (W & dtm) != 0ull || (qfl & (PF_MISSED_RIICHI | PF_MISSED_DOUJUN));
    uint32_t ronm = (uint32_t)__ballot(other && !in_discards && !in_missed && !furiten && ((W >> tt) & 1ull)) & 0xFu;
    const uint32_t riichi_m = (uint32_t)__ballot(lane < 4 && (qfl & PF_RIICHI_DECLARED)) & 0xFu;
    if (FAST && ronm) { c.bail = true; return 0u; }
    while (ronm) {  // rare: a seat waits on this tile and is not furiten -> yaku check
        int i = __ffs((int)ronm) - 1;
        ronm &= ronm - 1u;
        PState& P = S.p[i];
        uint32_t cf = base_cf(P);
        if (S.drawable_count == 0 && !S.is_rinshan) cf |= CF_HOUTEI;
        CalcOut r = seat_calc(c, i, -1, tile, cf, S.honba, false);
        if (r.is_win) {
            put_legal(c, i, 0, mk_action(RMJ_RON, tile, 0));
            if (lane == 0) c.X.nl[i] = 1;
            S.ron_offer_mask |= (uint8_t)(1u << i);
        } else if (r.shape) {
            P.flags |= PF_MISSED_DOUJUN;  // state/mod.rs:1386-1389
        }
    }
    const bool can_call = U((int)S.drawable_count) > 0;
    const bool kuikae = rule(c, RMJ_RULE_KUIKAE_FORBIDDEN);
    PROF(c.X, lane, 17);
    // ---- C (lane = 16*seat + slot)
    {
        const int k = lane >> 4, slot = lane & 15;
        const PState& G = S.p[k];
        const bool m = k < KNP && k != pid && slot < G.hand_len && (G.hand[slot] >> 2) == tt;
        uint64_t mb = __ballot(m);
        if (!can_call) mb = 0ull;
#pragma unroll 1
        for (int i = 0; i < KNP; i++) {
            const uint32_t sm = (uint32_t)(mb >> (16 * i)) & 0xFFFFu;
            const int count = __popc(sm);
            if (count < 2 || ((riichi_m >> i) & 1u)) continue;
            PState& P = S.p[i];
            const int hl = U((int)P.hand_len);
            wave_sync();
            int n = U(c.X.nl[i]);
            const int i0 = __ffs((int)sm) - 1;
            const uint32_t m1 = sm & (sm - 1u);
            const int i1 = __ffs((int)m1) - 1;
            if (hl >= 3 && (kuikae ? (hl - count) > 0 : (hl - 2) > 0)) {
                put_legal(c, i, n++, mk_action(RMJ_PON, tile, 2, P.hand[i0], P.hand[i1]));
                if (count >= 3) {
                    const int i2 = __ffs((int)(m1 & (m1 - 1u))) - 1;
                    put_legal(c, i, n++, mk_action(RMJ_PON, tile, 2, P.hand[i0], P.hand[i2]));
                    put_legal(c, i, n++, mk_action(RMJ_PON, tile, 2, P.hand[i1], P.hand[i2]));
                }
            }
            if (count >= 3) {
                const int i2 = __ffs((int)(m1 & (m1 - 1u))) - 1;
                put_legal(c, i, n++, mk_action(RMJ_DAIMINKAN, tile, 3, P.hand[i0], P.hand[i1], P.hand[i2]));
            }
            if (lane == 0) c.X.nl[i] = n;
        }
    }
    PROF(c.X, lane, 18);
    // ---- D: Chi for the next seat (lane = pattern*16 + a*4 + b); no Chi in 3P (state_3p/legal_actions.rs:386)
    if (!KSANMA && can_call && tt < 27) {
        const int i = (pid + 1) & 3;
        PState& P = S.p[i];
        const int hl = U((int)P.hand_len);
        if (!((riichi_m >> i) & 1u) && hl >= 3) {
            const uint8_t ht = lane < hl ? P.hand[lane] : 0xFF;
            const int hty = ht >> 2;
            const int r9 = tt % 9;
            const uint64_t m_m2 = __ballot(lane < hl && hty == tt - 2), m_m1 = __ballot(lane < hl && hty == tt - 1);
            const uint64_t m_p1 = __ballot(lane < hl && hty == tt + 1), m_p2 = __ballot(lane < hl && hty == tt + 2);
            // a sequence needs two of the four neighbours, adjacent around the tile (and inside the suit: pat_ok below)
            if ((m_m2 && m_m1) || (m_m1 && m_p1) || (m_p1 && m_p2)) {
                const uint64_t m_0 = __ballot(lane < hl && hty == tt);
                const uint64_t m_p3 = __ballot(lane < hl && hty == tt + 3), m_m3 = __ballot(lane < hl && hty == tt - 3);
                const int k = lane >> 4, a = (lane >> 2) & 3, b = lane & 3;
                const bool pat_ok = k == 0 ? (r9 >= 2) : (k == 1 ? (r9 >= 1 && r9 <= 7) : (k == 2 ? (r9 <= 6) : false));
                const uint64_t ma = k == 0 ? m_m2 : (k == 1 ? m_m1 : m_p1);
                const uint64_t mbb = k == 0 ? m_m1 : (k == 1 ? m_p1 : m_p2);
                // kuikae: forbidden = {tt} (+ tt+3 for pattern 2 if r9<=5, + tt-3 for pattern 0 if r9>=3)
                int forb = __popcll(m_0);
                if (k == 2 && r9 <= 5) forb += __popcll(m_p3);
                if (k == 0 && r9 >= 3) forb += __popcll(m_m3);
                const bool kk_ok = kuikae ? (hl - 2 - forb) > 0 : (hl - 2) > 0;
                const bool valid = lane < 48 && pat_ok && a < __popcll(ma) && b < __popcll(mbb) && kk_ok;
                uint64_t act = 0;
                if (valid) {
                    uint64_t x = ma;
                    for (int q = 0; q < a; q++) x &= x - 1;
                    uint64_t y = mbb;
                    for (int q = 0; q < b; q++) y &= y - 1;
                    act = mk_action(RMJ_CHI, tile, 2, P.hand[__ffsll((long long)x) - 1], P.hand[__ffsll((long long)y) - 1]);
                }
                const uint64_t vb = __ballot(valid);
                wave_sync();
                const int n = U(c.X.nl[i]);
                if (valid) {
                    int pos = n + __popcll(vb & lanemask_lt(lane));
                    if (pos < RMJ_MAX_LEGAL) c.X.legal[i][pos] = act;
                }
                if (lane == 0) c.X.nl[i] = n + __popcll(vb);
            }
        }
    }
    wave_sync();
    PROF(c.X, lane, 19);
    // ---- E (lane = seat): Pass, lengths, stale counts
    int n = lane < 4 ? c.X.nl[lane] : 0;
    if (lane < 4) {
        S.stale_n[lane] = (uint8_t)(n > 62 ? 62 : n);
        if (n > 0) {
            if (n < RMJ_MAX_LEGAL) c.X.legal[lane][n] = mk_action(RMJ_PASS, RMJ_TILE_NONE, 0);
            c.X.nl[lane] = n + 1;
        }
    }
    PROF(c.X, lane, 27);
    return (uint32_t)__ballot(lane < 4 && n > 0) & 0xFu;
}

// sh13: shanten of the 13 tiles the seat held before the draw if known (else -1): one tile lowers it by at most one
__device__ __forceinline__ uint32_t tenpai_after_discard(Ctx& c, const PState& P, const PH& full, int sh13) {
    uint32_t out = 0;
    int hl = P.hand_len;
    if (hl + 3 * P.n_melds != 14) return 0;
    if (sh13 >= 2) return 0;
    // Sound prefilter: a tenpai 13-tile subset implies replacement number <= 1 for the 14 tiles
    // (swap the discard for the winning tile), i.e. shanten(14) <= 0.  Only then run the exact probes.
    // (4P tables; a sanma hand has no 2m-8m, for which the 4P number is a lower bound of the 3P one, so it is sound too)
    const ShantenTables T = sh_tables_of(c.E);
    if (sh_shanten_wave(full, hl / 3, T, c.lane) > 0) return 0;
    // Second sieve, lane = hand slot: the table shanten of the 13 tiles left after discarding slot j.  A hand with a wait
    // has shanten 0, so only the slots that pass can be tenpai; they alone get the exact probe (one to three instead of
    // up to fourteen probes - these hands were the slowest waves of a launch).
    bool maybe = false;
    if (c.lane < hl) {
        PH h = full;
        ph_sub(h, P.hand[c.lane] >> 2);
        maybe = sh_shanten(h, (hl - 1) / 3, false, T) <= 0;
    }
    const uint32_t cand = (uint32_t)__ballot(maybe);
    int prev_ty = -1;
    bool prev_res = false;
    for (int j = 0; j < hl; j++) {
        int ty = P.hand[j] >> 2;
        bool res = false;
        if ((cand >> j) & 1u) {
            if (ty == prev_ty) res = prev_res;
            else {
                PH h = full;
                ph_sub(h, ty);
                res = wave_waits(h, c.lane) != 0ull;
            }
            prev_ty = ty;
            prev_res = res;
        }
        if (res) out |= 1u << j;
    }
    return out;
}

// legal_actions.rs:11-252 (WaitAct branch) for the current player
template <bool FAST>
__device__ __forceinline__ void gen_act_legal(Ctx& c, int pid) {
    GState& S = c.S;
    pid = U(pid);
    PState& P = S.p[pid];
    const int lane = c.lane;
    const int hl = U((int)P.hand_len);
    const int nmelds = U((int)P.n_melds);
    int n = 0;
    const uint32_t pflags = U((uint32_t)P.flags);
    const bool r_decl = pflags & PF_RIICHI_DECLARED, r_stage = pflags & PF_RIICHI_STAGE;
    const int drawn_tile = U((int)S.drawn_tile);
    const bool drawn = drawn_tile != 0xFF;
    const int drawable = U((int)S.drawable_count);
    const bool first_turn = U((int)S.is_first_turn) != 0;
    const bool rinshan = U((int)S.is_rinshan) != 0;
    // waits of the acting seat: non-empty only for a (poked) 13-tile holder, state/mod.rs:220-225
    if (FAST) {
        if (hl + 3 * nmelds == 13) { c.bail = true; return; }
        c.X.wout[pid] = 0ull;
    } else {
        c.X.wout[pid] = seat_waits(c, pid);
    }
    // 1. Tsumo
    if (drawn && !r_stage) {
        int tile = drawn_tile;
        int idx = -1;  // rposition
        int same_type;  // tiles of the drawn type in the 14
        {
            const int ht0 = lane < hl ? (int)P.hand[lane] : -1;
            uint64_t b = __ballot(ht0 == tile);
            if (b) idx = 63 - __clzll((long long)b);
            same_type = __popcll(__ballot(ht0 >= 0 && (ht0 >> 2) == (tile >> 2)));
        }
        uint32_t cf = base_cf(P) | CF_TSUMO;
        if (drawable == 0 && !rinshan) cf |= CF_HAITEI;
        if (rinshan) cf |= CF_RINSHAN;
        if (first_turn && U((int)P.n_discards) == 0) cf |= CF_FIRST_TURN;  // quirk Q5
        // Win-shape probe.  is_agari(13 tiles + drawn) == "drawn type is a wait of the 13 tiles" as long as the drawn type
        // has < 4 copies among them (quirk Q7: get_waits skips a type already held four times - only a poked state with a
        // repeated id, like the reference's own actions/test_riichi_pass.py, holds five; it takes the direct probe), and
        // those waits are the seat's cached waits13: the cache describes the hand before the draw and survives a
        // tsumogiri.  Only an invalid cache costs a probe, and that probe refills it for the claim checks that follow.
        bool shape;
        if (idx >= 0 && same_type <= 4 && (hl - 1) + 3 * nmelds == 13) {
            uint64_t W13;
            if (pflags & PF_WAITS_VALID) W13 = P.waits13;
            else {
                // (hand_len is 14 here; fill_waits13 wants the mentsu count of the 13 tiles: same quotient)
                W13 = fill_waits13(c, P, build_ph_wave(P, lane, idx));
            }
            shape = (W13 >> (tile >> 2)) & 1ull;
        } else {
            if (FAST) { c.bail = true; return; }
            shape = seat_shape(c, pid, idx, tile);
        }
        if (shape) {
            if (FAST) { c.bail = true; return; }  // a complete hand: yaku evaluation lives in the full path
            CalcOut r = seat_calc(c, pid, idx, tile, cf, S.honba, false);
            if (r.is_win && (r.yakuman || r.han >= 1)) put_legal(c, pid, n++, mk_action(RMJ_TSUMO, tile, 0));
        }
    }
    PROF(c.X, lane, 9);
    TLF(c, 8);
    const PH full = build_ph_wave(P, lane);  // shared by the riichi probe and the kan checks
    // 2. Discard / Riichi
    uint8_t ht = lane < hl ? P.hand[lane] : 0xFF;
    const int nforb = U((int)P.n_forbidden);  // at most two entries (kuikae)
    const bool forb = (nforb > 0 && (P.forbidden[0] >> 2) == (ht >> 2)) || (nforb > 1 && (P.forbidden[1] >> 2) == (ht >> 2));
    if (r_decl) {
        if (drawn) put_legal(c, pid, n++, mk_action(RMJ_DISCARD, drawn_tile, 0));
    } else {
        uint32_t tp = 0;
        bool need_tp = r_stage;
        const bool all_closed = __ballot(lane < nmelds && P.meld_type[lane & 3] != RMJ_MELD_ANKAN) == 0ull;
        // quirk Q8: >= 4 in 4P, > 0 in 3P
        bool riichi_pre = !r_stage && U(P.score) >= 1000 && (KSANMA ? drawable > 0 : drawable >= 4) && all_closed;
        if (need_tp || riichi_pre) {
            // the cache (if valid) describes the 13 tiles without the drawn one: it survives a draw and a riichi declaration
            int sh13 = (drawn && (U((uint32_t)P.flags) & PF_WAITS_VALID)) ? U((int)P.sh13) : -1;
            tp = tenpai_after_discard(c, P, full, sh13);
        }
        bool ok = lane < hl && !forb && (!r_stage || ((tp >> lane) & 1u));
        uint64_t vb = __ballot(ok);
        if (ok) {
            int pos = n + __popcll(vb & lanemask_lt(lane));
            if (pos < RMJ_MAX_LEGAL) c.X.legal[pid][pos] = mk_action(RMJ_DISCARD, ht, 0);
        }
        n += __popcll(vb);
        if (riichi_pre && tp != 0u) put_legal(c, pid, n++, mk_action(RMJ_RIICHI, RMJ_TILE_NONE, 0));
    }
    PROF(c.X, lane, 10);
    TLF(c, 14);
    // 3. Kan
    if (drawable > 0 && drawn) {
        if (!r_decl && !r_stage) {
            // Ankan: types with 4 copies, ascending type.  lane = hand index, a type is reported by its first holder.
            int ty = ht >> 2;
            // wave-uniform gate from the histogram: some type is held four (or more) times
            const bool any4 = (((full.a | full.b | full.c) & O9_4) | (full.d & O7_4)) != 0u;
            uint64_t rb = 0ull;
            bool is4 = false;
            if (any4) {
                int cnt = 0, first = 1;
                for (int k = 0; k < hl; k++) {
                    int tk = P.hand[k] >> 2;
                    cnt += (tk == ty);
                    if (k < lane && tk == ty) first = 0;
                }
                is4 = lane < hl && cnt == 4 && first;
                rb = __ballot(is4);  // order by type: rank among reporting lanes by type value
            }
            if (rb) {
                int rank = 0;
                for (uint64_t q = rb; q; q &= q - 1) {
                    int l2 = __ffsll((long long)q) - 1;
                    int t2 = P.hand[l2] >> 2;
                    rank += (t2 < ty);
                }
                if (is4) {
                    uint32_t lo = (uint32_t)ty * 4u;
                    int pos = n + rank;
                    if (pos < RMJ_MAX_LEGAL) c.X.legal[pid][pos] = mk_action(RMJ_ANKAN, lo, 4, lo, lo + 1, lo + 2, lo + 3);
                }
                n += __popcll(rb);
            }
            // Kakan: meld order, then hand order
            for (int m = 0; m < nmelds; m++) {
                if (U((int)P.meld_type[m]) == RMJ_MELD_PON) {
                    int target = P.meld_tiles[m][0] >> 2;
                    bool hit = lane < hl && (ht >> 2) == target;
                    uint64_t kb = __ballot(hit);
                    if (hit) {
                        int pos = n + __popcll(kb & lanemask_lt(lane));
                        if (pos < RMJ_MAX_LEGAL)
                            c.X.legal[pid][pos] =
                                mk_action(RMJ_KAKAN, ht, 3, P.meld_tiles[m][0], P.meld_tiles[m][1], P.meld_tiles[m][2]);
                    }
                    n += __popcll(kb);
                }
            }
        } else if (r_decl) {
            int t = drawn_tile, t34 = t >> 2;
            if (ph_cnt(full, t34) == 4) {
                if (FAST) { c.bail = true; return; }  // ankan after riichi: two wait probes, full path
                PH pre = full;
                ph_sub(pre, t34);
                uint64_t wpre = 0, wpost = 0;
                if (ph_total(pre) + 3 * P.n_melds == 13) wpre = wave_waits(pre, lane);
                PH post = full;
                ph_sub(post, t34); ph_sub(post, t34); ph_sub(post, t34); ph_sub(post, t34);
                if (ph_total(post) + 3 * (P.n_melds + 1) == 13) wpost = wave_waits(post, lane);
                if (wpre == wpost && wpre != 0ull) {
                    uint32_t lo = (uint32_t)t34 * 4u;
                    put_legal(c, pid, n++, mk_action(RMJ_ANKAN, lo, 4, lo, lo + 1, lo + 2, lo + 3));
                }
            }
        }
    }
    PROF(c.X, lane, 11);
    TLF(c, 15);
    // 4. Kyushu kyuhai
    if (first_turn && !r_stage && U((int)(S.p[0].n_melds | S.p[1].n_melds | S.p[2].n_melds | S.p[3].n_melds)) == 0) {
        // kinds of terminals and honors in the hand, lane = hand slot: the 13 kinds as bits 0..12 of a ballot-free OR over the wave
        // (round 3: this was a scalar walk over the 14 tiles, 4 000 cycles of LDS round trips at every round start)
        uint32_t bit = 0u;
        if (lane < hl && is_terminal_tile136(ht)) {
            const int ty = ht >> 2;
            bit = 1u << (ty >= 27 ? 6 + (ty - 27) : 2 * (ty / 9) + (ty % 9 ? 1 : 0));
        }
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) bit |= (uint32_t)__shfl_xor((int)bit, off, 64);   // slots 0..15 hold every tile of the hand
        const uint32_t kinds = (uint32_t)__builtin_amdgcn_readfirstlane((int)bit);
        if (__popc(kinds) >= 9) put_legal(c, pid, n++, mk_action(RMJ_KYUSHU, RMJ_TILE_NONE, 0));
    }
    // 5. Kita (state_3p/sanma.rs:146-169): one action per North tile in hand, hand order
    if (KSANMA && drawn && drawable > 0) {
        bool hit = lane < hl && (ht >> 2) == 30;
        uint64_t kb = __ballot(hit);
        if (hit) {
            int pos = n + __popcll(kb & lanemask_lt(lane));
            if (pos < RMJ_MAX_LEGAL) c.X.legal[pid][pos] = mk_action(RMJ_KITA, ht, 0);
        }
        n += __popcll(kb);
    }
    c.X.nl[pid] = n > RMJ_MAX_LEGAL ? RMJ_MAX_LEGAL : n;
    PROF(c.X, lane, 12);
}


// current_claims.entry(i).or_default().push(Ron) (state/mod.rs:524-533, 665-673; sanma.rs:121-128): the Ron offer is
// APPENDED to whatever the seat still has in current_claims (see GState::stale_n).
__device__ __forceinline__ void offer_ron(Ctx& c, int i, int tile) {
    GState& S = c.S;
    int sn = S.stale_n[i];
    if (c.lane < sn) c.X.legal[i][c.lane] = c.Lg[i * RMJ_MAX_LEGAL + c.lane];
    wave_sync();
    put_legal(c, i, sn, mk_action(RMJ_RON, tile, 0));
    put_legal(c, i, sn + 1, mk_action(RMJ_PASS, RMJ_TILE_NONE, 0));
    c.X.nl[i] = sn + 2;
    S.stale_n[i] = (uint8_t)(sn + 1 > 62 ? 62 : sn + 1);
    S.ron_offer_mask |= (uint8_t)(1u << i);
}

// ---------------------------------------------------------------- transitions
__device__ void ol_init_next_round(CtxV v, bool oya_won, bool is_draw);
__device__ void ol_trigger_ryukyoku(CtxV v, int reason, int offender);
__device__ __forceinline__ void init_next_round(Ctx& c, bool oya_won, bool is_draw) { ol_init_next_round(ctx_pack(c), oya_won, is_draw); }
__device__ __forceinline__ void trigger_ryukyoku(Ctx& c, int reason, int offender) { ol_trigger_ryukyoku(ctx_pack(c), reason, offender); }

// state/mod.rs:2021-2046
__device__ __forceinline__ void reveal_kan_dora(Ctx& c) {
    GState& S = c.S;
    int count = S.n_dora;
    if (count < 5) {
        // 4P: tiles[4+2k-rinshan] = W[4+2k] if still inside the wall; 3P: pre-extracted W[8+2k], no bound check
        int widx = KSANMA ? 8 + 2 * count : 4 + 2 * count;
        if (KSANMA || widx < S.live_end) {
            uint8_t t = c.W[widx];
            S.dora[count] = t;
            S.n_dora = (uint8_t)(count + 1);
            emit_simple(c, RMJ_EV_DORA, 0, t);
        }
    }
}
__device__ __forceinline__ void flush_pending_kan_dora(Ctx& c) {
    for (int n = U((int)c.S.pending_kan_dora); n > 0; n--) {
        c.S.pending_kan_dora = (uint8_t)(n - 1);
        reveal_kan_dora(c);
    }
}
// state/mod.rs:1549-1567
__device__ __forceinline__ void accept_riichi(Ctx& c) {
    GState& S = c.S;
    const int p = U((int)S.riichi_pending);
    if (p != 0xFF) {
        c.dirty |= 1u << p;
        S.p[p].score -= 1000;
        S.p[p].score_delta -= 1000;
        S.riichi_sticks += 1;
        S.p[p].flags |= PF_RIICHI_DECLARED | PF_IPPATSU;
        emit_simple(c, RMJ_EV_REACH_ACCEPTED, (uint8_t)p);
        S.riichi_pending = 0xFF;
    }
}
// state/mod.rs:2071-2082
__device__ inline void process_end_game(Ctx& c) {
    c.S.is_done = 1;
    emit_simple(c, RMJ_EV_END_KYOKU);
    emit_simple(c, RMJ_EV_END_GAME);
}
// state/mod.rs:1569-1593
template <bool FAST = false>
__device__ __forceinline__ void deal_next(Ctx& c) {
    GState& S = c.S;
    S.is_rinshan = 0;
    const int drawable = U((int)S.drawable_count);
    if (drawable == 0) {
        if (FAST) { c.bail = true; return; }
        TLF(c, 1);     // everything of the step before the exhaustive draw (discard, claims)
        trigger_ryukyoku(c, RMJ_RK_EXHAUSTIVE, 0);
        return;
    }
    const int live_end = U((int)S.live_end);
    if (live_end > U((int)S.rinshan_count)) {
        const int le = live_end - 1;
        // FAST: nothing before this point moves live_end (kans bail), the prefetched tile is the draw
        uint8_t t = (FAST || c.pf_live_end == le + 1) ? (uint8_t)c.pf_draw : c.W[le];
        S.live_end = (uint8_t)le;
        S.drawable_count = (uint8_t)(drawable - 1);
        const int pid = U((int)S.current_player);
        c.dirty |= 1u << pid;
        PState& P = S.p[pid];
        const int hl = U((int)P.hand_len);
        if (hl < 14) { P.hand[hl] = t; P.hand_len = (uint8_t)(hl + 1); }
        S.drawn_tile = t;
        S.needs_tsumo = 0;
        S.phase = RMJ_WAIT_ACT;
        S.active_mask = (uint8_t)(1u << pid);
        emit_simple(c, RMJ_EV_TSUMO, (uint8_t)pid, t);
        P.n_forbidden = 0;
    }
}

// the build's seed -> wall definition (oracle/riichi_state.hpp build_wall); writes W[0..135] (already reversed):
// ids sorted by (key, id), key_i = splitmix64(hs + i * phi).  The rank of a tile is found with a 128-bucket counting
// pass on the top 7 key bits (LDS atomics + one wave scan) and an exact (key, id) comparison inside its own bucket
// (about one member on average) instead of 136 comparisons per tile: the round restart is the longest thing a wave
// ever does, and its tail sets the duration of the launch.
__device__ inline void shuffle_wall(Ctx& c) {
    GState& S = c.S;
    const int lane = c.lane;
    uint64_t hs = sm64(S.wall_seed + (uint64_t)S.hand_index);
    S.hand_index += 1;
    constexpr int N = KSANMA ? 108 : 136;  // 3P: ids without 2m-8m (types.rs:378-382)
    if (rule(c, RMJ_RULE_REFERENCE_RNG)) {   // the reference's own definition (rmj_refrng.hip.h); salt behind the wall, digest on demand
        uint8_t* w = c.X.maskbuf;            // scratch (free until finalize): key stream / indices in X.legal, w before the reversal in maskbuf
        const uint64_t salt = refrng_wall<N, KSANMA>(hs, lane, reinterpret_cast<uint32_t*>(&c.X.legal[0][0]), w);
        for (int i = lane; i < 144; i += 64) c.X.tiles[i] = i < N ? w[N - 1 - i] : (i >= 136 ? (uint8_t)(salt >> (8 * (i - 136))) : (uint8_t)0);
        if (lane == 0) S.wall_meta = 1;
        wave_sync();
        return;
    }
    if (lane < 8) c.X.tiles[136 + lane] = 0;
    // scratch (free until finalize): grouped keys in X.legal[0..135], bucket counters behind them, source index in maskbuf
    uint64_t* gk = &c.X.legal[0][0];
    uint32_t* cnt = reinterpret_cast<uint32_t*>(&c.X.legal[0][0] + 144);  // 129 counters (+1 sentinel)
    uint8_t* gi = c.X.maskbuf;
    for (int q = lane; q < 130; q += 64) cnt[q] = 0u;
    uint64_t key[3];
    uint32_t pos[3];
#pragma unroll
    for (int q = 0; q < 3; q++) {
        const int i = lane + 64 * q;
        key[q] = sm64(hs + (uint64_t)i * 0x9E3779B97F4A7C15ull);
    }
    wave_sync();
#pragma unroll
    for (int q = 0; q < 3; q++) {
        const int i = lane + 64 * q;
        pos[q] = i < N ? atomicAdd(&cnt[(uint32_t)(key[q] >> 57)], 1u) : 0u;
    }
    wave_sync();
    {   // exclusive prefix sum over the 128 buckets (two per lane), written back in place; cnt[128] = N
        const uint32_t v0 = cnt[2 * lane], v1 = cnt[2 * lane + 1];
        uint32_t incl = v0 + v1;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            uint32_t up = (uint32_t)__shfl_up((int)incl, off, 64);
            if (lane >= off) incl += up;
        }
        const uint32_t excl = incl - (v0 + v1);
        wave_sync();
        cnt[2 * lane] = excl;
        cnt[2 * lane + 1] = excl + v0;
        if (lane == 63) cnt[128] = incl;
    }
    wave_sync();
#pragma unroll
    for (int q = 0; q < 3; q++) {
        const int i = lane + 64 * q;
        if (i < N) {
            const uint32_t slot = cnt[(uint32_t)(key[q] >> 57)] + pos[q];
            gk[slot] = key[q];
            gi[slot] = (uint8_t)i;
        }
    }
    wave_sync();
#pragma unroll
    for (int q = 0; q < 3; q++) {
        const int sl = lane + 64 * q;
        if (sl < N) {
            const uint64_t k = gk[sl];
            const int i = gi[sl];
            const uint32_t bkt = (uint32_t)(k >> 57);
            const int lo = (int)cnt[bkt], hi = (int)cnt[bkt + 1];
            int r = lo;
            for (int t = lo; t < hi; t++) {
                const uint64_t kt = gk[t];
                const int it = gi[t];
                r += (kt < k) || (kt == k && it < i);
            }
            const int id = (KSANMA && i >= 4) ? i + 28 : i;  // i-th id of the tile universe
            c.X.tiles[N - 1 - r] = (uint8_t)id;               // w[r] = id, then reverse
        }
    }
    wave_sync();
}

// state/mod.rs:1695-1844.  Wall must already be in c.X.tiles (reversed orientation W).
__device__ inline void init_round(Ctx& c, int oya, int round_wind, int honba, uint32_t kyotaku, const int32_t* scores) {
    GState& S = c.S;
    const int lane = c.lane;
    S.oya = (uint8_t)oya;
    S.kyoku_idx = (uint8_t)oya;
    S.current_player = (uint8_t)oya;
    S.honba = (uint8_t)honba;
    S.riichi_sticks = kyotaku;
    S.round_wind = (uint8_t)round_wind;
    for (int p = 0; p < 4; p++) {  // PlayerState::reset_round, state/player.rs:66-86
        PState& P = S.p[p];
        P.hand_len = 0; P.n_melds = 0; P.n_discards = 0;
        P.flags = PF_NAGASHI;
        P.pao37 = 0xFF; P.pao50 = 0xFF;
        P.n_forbidden = 0;
        P.riichi_decl_idx = 0xFF; P.riichi_sutehai = 0xFF; P.last_tedashi = 0xFF;
        P.score_delta = 0;
        P.discard_from_hand_bits = 0; P.discard_is_riichi_bits = 0;
        P.discard_type_mask = 0;
        P.n_kita = 0;
        if (scores && p < KNP) P.score = scores[p];
    }
    S.is_done = 0;
    S.pending_kan_pid = 0xFF;
    S.pending_kan_action = 0;
    S.is_rinshan = 0;
    S.rinshan_count = 0;
    S.pending_kan_dora = 0;
    S.is_first_turn = 1;
    S.riichi_pending = 0xFF;
    S.turn_count = 0;
    S.needs_tsumo = 1;
    S.last_discard_pid = 0xFF;
    S.last_discard_tile = 0;
    S.ron_offer_mask = 0;
    S.win_mask = 0;  // win_results.clear() (state/mod.rs:1729)
    for (int p = 0; p < 4; p++) S.stale_n[p] = 0;
    // publish the wall to HBM (W) and deal from LDS
    for (int i = lane; i < RMJ_WALL_STRIDE / 4; i += 64)
        reinterpret_cast<uint32_t*>(c.W)[i] = reinterpret_cast<const uint32_t*>(c.X.tiles)[i];   // [136..143]: the salt (or 0)
    const int np = KNP;
    const int total = KSANMA ? 108 : 136;
    S.wall_total = (uint8_t)total;
    S.n_dora = 1;
    S.dora[0] = c.X.tiles[KSANMA ? 8 : 4];  // state_3p/wall.rs:104-112
    // deal: pop #n = X.tiles[total - 1 - n]; three rounds of 4 tiles per seat from the dealer, then one each
    if (lane < 13 * np) {
        int n = lane, p, pos;
        if (n < 12 * np) {
            int r = n / (4 * np), rem = n - r * 4 * np;
            int idx = rem >> 2, k = rem & 3;
            p = (idx + oya) % np;
            pos = r * 4 + k;
        } else {
            p = ((n - 12 * np) + oya) % np;
            pos = 12;
        }
        S.p[p].hand[pos] = c.X.tiles[total - 1 - n];
    }
    wave_sync();
    for (int p = 0; p < np; p++) {
        S.p[p].hand_len = 13;
        sort_hand(c, S.p[p], 13);
    }
    S.live_end = (uint8_t)(total - 13 * np);
    S.drawable_count = (uint8_t)(S.live_end - 14);
    TLF(c, 6);         // record reset, wall to HBM, deal, four sorts
    if (!c.E.skip_log) {
        RmjEvent e = ev_zero(RMJ_EV_START_KYOKU);
        e.actor = (uint8_t)oya;
        e.target = (uint8_t)(oya + 1);
        e.tile = S.dora[0];
        e.consumed[0] = (uint8_t)(round_wind & 3);
        e.consumed[1] = (uint8_t)honba;
        e.consumed[2] = (uint8_t)(kyotaku & 0xFF);
        e.consumed[3] = (uint8_t)((kyotaku >> 8) & 0xFF);
        for (int p = 0; p < 4; p++) e.deltas[p] = p < np ? S.p[p].score : 0;
        emit_raw(c, e);
        for (int half = 0; half < 2; half++) {
            RmjEvent t = ev_zero(RMJ_EV_TEHAI);
            t.actor = (uint8_t)half;
            uint8_t* pl = reinterpret_cast<uint8_t*>(&t) + 4;
            for (int k = 0; k < 13; k++) {
                pl[k] = S.p[2 * half].hand[k];
                pl[13 + k] = (2 * half + 1 < np) ? S.p[2 * half + 1].hand[k] : 0;
            }
            emit_raw(c, t);
        }
    }
    TLF(c, 7);         // start_kyoku + tehai events
    S.current_player = (uint8_t)oya;
    S.phase = RMJ_WAIT_ACT;
    S.active_mask = (uint8_t)(1u << oya);
    {
        uint8_t t = c.X.tiles[--S.live_end];
        S.drawable_count -= 1;
        PState& P = S.p[oya];
        P.hand[P.hand_len++] = t;
        S.drawn_tile = t;
        S.needs_tsumo = 0;
        emit_simple(c, RMJ_EV_TSUMO, (uint8_t)oya, t);
    }
}

// state/mod.rs:1595-1688
__device__ __noinline__ void ol_init_next_round(CtxV v, bool oya_won, bool is_draw) {
    CTX_FROM(v);
    GState& S = c.S;
    if (S.is_done) return;
    const int np = KNP;
    const int32_t goal = KSANMA ? 40000 : 30000;  // state_3p/mod.rs:1524,1553,1561
    int32_t sc[4] = {0, 0, 0, 0};
    bool neg = false;
    int32_t max_score = S.p[0].score;
    for (int p = 0; p < np; p++) {
        sc[p] = S.p[p].score;
        neg = neg || sc[p] < 0;
        max_score = max(max_score, sc[p]);
    }
    if (neg) { process_end_game(c); return; }
    int oya = S.oya;
    int32_t ds = sc[oya];
    bool top = true;
    for (int seat = 0; seat < np; seat++) top = top && (seat == oya || ds > sc[seat] || (ds == sc[seat] && oya <= seat));
    uint32_t gm = c.E.game_mode;
    bool last_regular = false;
    if (gm == 1 || gm == 4) last_regular = S.round_wind == 0 && oya == np - 1;
    if (gm == 2 || gm == 5) last_regular = S.round_wind == 1 && oya == np - 1;
    if (oya_won && last_regular && top && ds >= goal) { process_end_game(c); return; }
    int next_honba = S.honba, next_oya = oya, next_rw = S.round_wind;
    if (oya_won) {
        next_honba = min(next_honba + 1, 255);
    } else {
        next_honba = is_draw ? min(next_honba + 1, 255) : 0;
        next_oya = (next_oya + 1) % np;
        if (next_oya == 0) next_rw += 1;
    }
    bool end = false;
    if (gm == 1 || gm == 4) end = next_rw >= 1 && (max_score >= goal || next_rw > 1);
    else if (gm == 2 || gm == 5) end = next_rw >= 2 && (max_score >= goal || next_rw > 2);
    else if (gm == 0 || gm == 3) end = true;
    else end = next_rw >= 1;
    if (end) { process_end_game(c); return; }
    emit_simple(c, RMJ_EV_END_KYOKU);
    uint32_t sticks = S.riichi_sticks;
    TLF(c, 4);         // next-round decision + end_kyoku
    shuffle_wall(c);
    TLF(c, 5);
    init_round(c, next_oya, next_rw, next_honba, sticks, sc);
    TLF(c, 9);
}

// tenpai of a seat at exhaustive draw: HandEvaluator::is_tenpai (hand_evaluator.rs:178-194)
__device__ inline bool seat_tenpai(Ctx& c, int seat) {
    return seat_waits(c, seat) != 0ull;
}

// state/mod.rs:1846-1968
__device__ __noinline__ void ol_trigger_ryukyoku(CtxV v, int reason, int offender) {
    CTX_FROM(v);
    reason = uni(reason); offender = uni(offender);
    GState& S = c.S;
    accept_riichi(c);
    const int np = KNP;
    bool tenpai[4] = {false, false, false, false};
    int final_reason = reason;
    uint32_t nagashi = 0;
    if (reason == RMJ_RK_EXHAUSTIVE) {
        for (int i = 0; i < np; i++) tenpai[i] = seat_tenpai(c, i);
        TLF(c, 2);     // accept_riichi + the seats' tenpai
        for (int i = 0; i < np; i++)
            if (S.p[i].flags & PF_NAGASHI) nagashi |= 1u << i;
        if (nagashi) {
            final_reason = RMJ_RK_NAGASHI;
            for (int w = 0; w < np; w++) {
                if (!((nagashi >> w) & 1u)) continue;
                bool is_oya = w == S.oya;
                ScoreOut s = calc_score(5, 30, is_oya, true, 0, np);
                for (int i = 0; i < np; i++) {
                    if (i == w) continue;
                    int32_t pay = is_oya ? (int32_t)s.tsumo_ko : (i == S.oya ? (int32_t)s.tsumo_oya : (int32_t)s.tsumo_ko);
                    S.p[i].score -= pay; S.p[i].score_delta -= pay;
                    S.p[w].score += pay; S.p[w].score_delta += pay;
                }
            }
        } else {
            int num_tp = tenpai[0] + tenpai[1] + tenpai[2] + tenpai[3];
            if (num_tp > 0 && num_tp < np) {
                const int32_t pool = KSANMA ? 2000 : 3000;  // state_3p/game_mode.rs:39-41
                int32_t pk = pool / num_tp, pn = pool / (np - num_tp);
                for (int i = 0; i < np; i++) {
                    int32_t d = tenpai[i] ? pk : -pn;
                    S.p[i].score += d;
                    S.p[i].score_delta = d;
                }
            }
        }
    } else if (reason == RMJ_RK_ILLEGAL) {
        int pid = offender;
        if (pid == S.oya) {
            int32_t penalty = 4000 * (np - 1), each = penalty / (np - 1);
            for (int i = 0; i < np; i++) {
                if (i == pid) { S.p[i].score -= penalty; S.p[i].score_delta = -penalty; }
                else { S.p[i].score += each; S.p[i].score_delta = each; }
            }
        } else {
            int32_t total = 4000 + 2000 * (np - 2);
            for (int i = 0; i < np; i++) {
                if (i == pid) { S.p[i].score -= total; S.p[i].score_delta = -total; }
                else if (i == S.oya) { S.p[i].score += 4000; S.p[i].score_delta = 4000; }
                else { S.p[i].score += 2000; S.p[i].score_delta = 2000; }
            }
        }
    }
    bool renchan;
    if (final_reason == RMJ_RK_EXHAUSTIVE) renchan = tenpai[S.oya];
    else if (final_reason == RMJ_RK_NAGASHI) renchan = (nagashi >> S.oya) & 1u;
    else renchan = true;
    if (!c.E.skip_log) {
        RmjEvent e = ev_zero(RMJ_EV_RYUKYOKU);
        e.flags = (uint8_t)final_reason;
        e.actor = (uint8_t)offender;
        for (int i = 0; i < np; i++) e.deltas[i] = S.p[i].score_delta;
        emit_raw(c, e);
    }
    TLF(c, 3);         // payments + ryukyoku event
    init_next_round(c, renchan, true);
}

// state/mod.rs:1970-2019
template <bool FAST = false>
__device__ __forceinline__ bool check_abortive_draw(Ctx& c) {
    GState& S = c.S;
    const int lane = c.lane;
    // lane = seat*4 + meld slot (16 lanes); seat facts are taken from slot 0 of each seat
    const int p = (lane >> 2) & 3, m = lane & 3;
    const PState& P = S.p[p];
    const bool in = lane < 16, seat_lane = in && m == 0;
    const int nm = P.n_melds;
    const uint32_t turns_ok = (uint32_t)__ballot(seat_lane && P.n_discards == 1);
    const uint32_t has_melds = (uint32_t)__ballot(seat_lane && nm != 0);
    const uint32_t riichi = (uint32_t)__ballot(seat_lane && (P.flags & PF_RIICHI_DECLARED));
    const uint32_t kan = (uint32_t)__ballot(in && m < nm && P.meld_type[m] >= RMJ_MELD_DAIMINKAN);
    const uint32_t all_seats = 0x1111u;
    if (!KSANMA && turns_ok == all_seats && has_melds == 0u) {  // sufuurenta / suucha riichi: disabled in 3P (state_3p/mod.rs:1861-1888)
        const int first = U((int)S.p[0].discards[0]) >> 2;
        if (first >= 27 && first <= 30) {
            uint32_t same = (uint32_t)__ballot(seat_lane && (P.discards[0] >> 2) == first);
            if (same == all_seats) {
                if (FAST) { c.bail = true; return true; }
                trigger_ryukyoku(c, RMJ_RK_SUFUURENTA, 0);
                return true;
            }
        }
    }
    if (__popc(kan) == 4) {  // suukansansen: four kans by at least two players
        int owner = (__ffs((int)kan) - 1) >> 2;
        if (kan & ~(0xFu << (4 * owner))) {
            if (FAST) { c.bail = true; return true; }
            trigger_ryukyoku(c, RMJ_RK_SUUKANSANSEN, 0);
            return true;
        }
    }
    if (!KSANMA && riichi == all_seats) {
        if (FAST) { c.bail = true; return true; }
        trigger_ryukyoku(c, RMJ_RK_SUUCHA_RIICHI, 0);
        return true;
    }
    return false;
}

// pao bookkeeping, state/mod.rs:1228-1259 / 1443-1472
__device__ inline void pao_check(Ctx& c, int claimer, int discarder, int tile) {
    PState& C = c.S.p[claimer];
    int tv = tile >> 2;
    int nd = 0, nw = 0;
    for (int m = 0; m < C.n_melds; m++) {
        int t = C.meld_tiles[m][0] >> 2;
        if (C.meld_type[m] != RMJ_MELD_CHI) {
            nd += (t >= 31 && t <= 33);
            nw += (t >= 27 && t <= 30);
        }
    }
    if (tv >= 31 && tv <= 33) {
        if (nd == 3) C.pao37 = (uint8_t)discarder;
    } else if (tv >= 27 && tv <= 30) {
        if (nw == 4) C.pao50 = (uint8_t)discarder;
    }
}

__device__ inline void push_meld(PState& P, int type, uint32_t t0, uint32_t t1, uint32_t t2, uint32_t t3, int n, int from, int called) {
    int m = P.n_melds;
    if (m >= 4) return;
    uint32_t v[4] = {t0, t1, t2, n == 4 ? t3 : 0xFFFFu};
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            if (v[j] > v[j + 1]) { uint32_t t = v[j]; v[j] = v[j + 1]; v[j + 1] = t; }
    for (int i = 0; i < 4; i++) P.meld_tiles[m][i] = (i < n) ? (uint8_t)v[i] : 0;
    P.meld_type[m] = (uint8_t)type;
    P.meld_from[m] = (uint8_t)from;
    P.meld_called[m] = (uint8_t)called;
    P.n_melds = (uint8_t)(m + 1);
}

// remove each consume tile from the hand (position lookup per tile, order preserved)
__device__ __forceinline__ void hand_remove_tiles(Ctx& c, PState& P, uint64_t act) {
    waits_invalidate(P);
    uint32_t n = a_n(act);
    for (uint32_t k = 0; k < n && k < 4; k++) {
        int idx = hand_find(c, P, (int)a_c(act, k));
        if (idx >= 0) hand_remove_at(c, P, idx);
    }
}

// state/mod.rs:1415-1547
__device__ void ol_resolve_kan(CtxV v, int pid, uint64_t action);
__device__ __forceinline__ void resolve_kan(Ctx& c, int pid, uint64_t action) { ol_resolve_kan(ctx_pack(c), pid, action); }
__device__ __noinline__ void ol_resolve_kan(CtxV v, int pid, uint64_t action) {
    CTX_FROM(v);
    pid = uni(pid); action = uni(action);
    GState& S = c.S;
    PState& P = S.p[pid];
    uint32_t ty = a_type(action);
    if (ty != RMJ_KAKAN) {
        hand_remove_tiles(c, P, action);
        if (ty == RMJ_ANKAN) {
            push_meld(P, RMJ_MELD_ANKAN, a_c(action, 0), a_c(action, 1), a_c(action, 2), a_c(action, 3), (int)min(a_n(action), 4u), 0xFF, 0xFF);
        } else {
            int discarder = S.last_discard_pid, tile = S.last_discard_tile;
            push_meld(P, RMJ_MELD_DAIMINKAN, a_c(action, 0), a_c(action, 1), a_c(action, 2), (uint32_t)tile, 4, discarder, tile);
            pao_check(c, pid, discarder, tile);
        }
    }
    S.is_first_turn = 0;
    for (int p = 0; p < 4; p++) S.p[p].flags &= ~PF_IPPATSU;
    if (S.drawable_count > 0) {
        uint8_t t = c.W[S.rinshan_count];
        S.rinshan_count += 1;
        S.drawable_count -= 1;
        if (P.hand_len < 14) P.hand[P.hand_len++] = t;
        S.drawn_tile = t;
        S.is_rinshan = 1;
        if (ty == RMJ_ANKAN) {
            uint32_t tile = a_tile(action) != RMJ_TILE_NONE ? a_tile(action) : a_c(action, 0);
            emit_meld(c, RMJ_EV_ANKAN, (uint8_t)pid, 0, (uint8_t)tile, action);
        } else if (ty == RMJ_DAIMINKAN) {
            emit_meld(c, RMJ_EV_DAIMINKAN, (uint8_t)pid, S.last_discard_pid, S.last_discard_tile, action);
        }
        flush_pending_kan_dora(c);
        if (ty == RMJ_ANKAN) reveal_kan_dora(c);
        else S.pending_kan_dora += 1;
        emit_simple(c, RMJ_EV_TSUMO, (uint8_t)pid, t);
        S.phase = RMJ_WAIT_ACT;
        S.active_mask = (uint8_t)(1u << pid);
    }
}

// state/mod.rs:1317-1413
template <bool FAST>
__device__ __forceinline__ void resolve_discard_tail(Ctx& c, int pid);
template <bool FAST = false>
__device__ __forceinline__ void resolve_discard(Ctx& c, int pid, int tile, bool tsumogiri) {
    GState& S = c.S;
    pid = U(pid); tile = U(tile); tsumogiri = U((int)tsumogiri) != 0;
    PState& P = S.p[pid];
    if (FAST && U((int)S.pending_kan_dora) > 0) { c.bail = true; return; }
    if (KSANMA) { S.pending_kan_pid = 0xFF; S.pending_kan_action = 0; }  // quirk Q11 (state_3p/mod.rs:1224-1227)
    S.is_rinshan = 0;
    uint32_t fl = U((uint32_t)P.flags);  // one LDS read / one write for the whole sequence of flag updates
    fl &= ~(uint32_t)PF_IPPATSU;
    const bool stage = fl & PF_RIICHI_STAGE;
    int nd = U((int)P.n_discards);
    if (nd < RMJ_MAX_DISCARDS) {
        P.discards[nd] = (uint8_t)tile;
        if (!tsumogiri) P.discard_from_hand_bits |= 1u << nd;
        if (stage) P.discard_is_riichi_bits |= 1u << nd;
        nd += 1;
        P.n_discards = (uint8_t)nd;
    }
    P.discard_type_mask |= 1ull << (tile >> 2);
    S.last_discard_pid = (uint8_t)pid;
    S.last_discard_tile = (uint8_t)tile;
    S.drawn_tile = 0xFF;
    if (!tsumogiri) {
        P.last_tedashi = (uint8_t)tile;
        // The 13-tile hand changed by one tile (a tsumogiri leaves it, and the cache, intact): its shanten moved by at
        // most one.  A cached lower bound >= 3 therefore stays a bound >= 2 after losing one: still "no waits" (waits13
        // is 0 for every sh13 >= 1) and still "no Riichi after the next draw" (tenpai_after_discard).  A bound of 2 is
        // recomputed instead of kept as 1: measured, the riichi probe it would no longer spare costs more than the refill.
        const int lb = U((int)P.sh13);
        if ((fl & PF_WAITS_VALID) && lb >= 3) P.sh13 = (uint8_t)(lb - 1);
        else fl &= ~(uint32_t)PF_WAITS_VALID;
    }
    S.needs_tsumo = 1;
    if (stage) {
        fl |= PF_RIICHI_DECLARED;
        if (U((int)S.is_first_turn)) fl |= PF_DOUBLE_RIICHI;
        P.riichi_decl_idx = (uint8_t)(nd - 1);
        fl &= ~(uint32_t)PF_RIICHI_STAGE;
        S.riichi_pending = (uint8_t)pid;
    }
    fl &= ~(uint32_t)PF_MISSED_DOUJUN;
    if (!is_terminal_tile136(tile)) fl &= ~(uint32_t)PF_NAGASHI;
    P.flags = (uint8_t)fl;
    flush_pending_kan_dora(c);
    emit_simple(c, RMJ_EV_DAHAI, (uint8_t)pid, (uint8_t)tile, tsumogiri ? 1 : 0);
    S.active_mask = 0;
    S.ron_offer_mask = 0;
    PROF(c.X, c.lane, 4);
    resolve_discard_tail<FAST>(c, pid);
}
// ... from the claims on (also the full path's entry behind a discard made by k_step4's tier 0: STEP_F_CONT_CLAIMS)
template <bool FAST>
__device__ __forceinline__ void resolve_discard_tail(Ctx& c, int pid) {
    GState& S = c.S;
    const int tile = U((int)S.last_discard_tile);
    const uint32_t claim_active = gen_claims_all<FAST>(c, pid, tile);
    if (FAST && c.bail) return;
    PROF(c.X, c.lane, 5);
    if (claim_active) {
        S.phase = RMJ_WAIT_RESPONSE;
        S.active_mask = (uint8_t)claim_active;
    } else {
        if (U((int)S.riichi_pending) != 0xFF) accept_riichi(c);
        bool abort_ = check_abortive_draw<FAST>(c);
        PROF(c.X, c.lane, 23);
        if (!abort_) {
            S.turn_count = U(S.turn_count) + 1u;
            S.current_player = (uint8_t)(pid + 1 == KNP ? 0 : pid + 1);
            deal_next<FAST>(c);
            PROF(c.X, c.lane, 24);
            if (U(S.turn_count) >= (uint32_t)KNP) S.is_first_turn = 0;  // re-read: deal_next may have started a new round
        }
    }
    PROF(c.X, c.lane, 6);
}

__device__ inline int yakuman_val(const Ctx& c, int yid) {
    if (yid == 47 && rule(c, RMJ_RULE_JUNSEI_CHUUREN_DOUBLE)) return 2;
    if (yid == 48 && rule(c, RMJ_RULE_SUUANKOU_TANKI_DOUBLE)) return 2;
    if (yid == 49 && rule(c, RMJ_RULE_KOKUSHI13_DOUBLE)) return 2;
    if (yid == 50 && rule(c, RMJ_RULE_DAISUUSHII_DOUBLE)) return 2;
    return 1;
}
// double-yakuman cap, state/mod.rs:720-745 / 1005-1030
__device__ inline void cap_double(const Ctx& c, CalcOut& r, bool is_oya, bool tsumo, uint32_t honba) {
    if (r.yakuman && r.han > 13) {
        int cap = 0;
        if (((r.ym >> 47) & 1) && !rule(c, RMJ_RULE_JUNSEI_CHUUREN_DOUBLE)) cap += 13;
        if (((r.ym >> 48) & 1) && !rule(c, RMJ_RULE_SUUANKOU_TANKI_DOUBLE)) cap += 13;
        if (((r.ym >> 49) & 1) && !rule(c, RMJ_RULE_KOKUSHI13_DOUBLE)) cap += 13;
        if (((r.ym >> 50) & 1) && !rule(c, RMJ_RULE_DAISUUSHII_DOUBLE)) cap += 13;
        if (cap > 0) {
            int h = r.han > cap ? r.han - cap : 0;
            r.han = h < 13 ? 13 : h;
            ScoreOut s = calc_score((uint32_t)r.han, 0, is_oya, tsumo, honba, (uint32_t)KNP);
            r.ron = s.ron; r.tsumo_oya = s.tsumo_oya; r.tsumo_ko = s.tsumo_ko;
        }
    }
}
// sums yakuman values over the (unordered) yakuman id set; pao liability for 37 / 50
__device__ inline void yakuman_totals(const Ctx& c, const CalcOut& r, const PState& P, int& total_val, int& pao_val, int& pao_payer) {
    total_val = 0; pao_val = 0;
    const int ids[15] = {35, 36, 37, 38, 39, 40, 41, 42, 43, 44, 45, 47, 48, 49, 50};
    // reference iterates res.yaku in emission order; pao_payer = liable seat of the LAST matching id in that order.
    // Emission order puts 37 before 50 (ORDER_YAKUMAN), and a hand cannot hold both, so order is immaterial here.
    for (int k = 0; k < 15; k++) {
        int y = ids[k];
        if (!((r.ym >> y) & 1ull)) continue;
        int v = yakuman_val(c, y);
        total_val += v;
        if (y == 37 && P.pao37 != 0xFF) { pao_val += v; pao_payer = P.pao37; }
        if (y == 50 && P.pao50 != 0xFF) { pao_val += v; pao_payer = P.pao50; }
    }
}

// win_results.insert(seat, val) (state/mod.rs:855-863, 1100-1107): the capped result with its ordered yaku list and the pao payer
__device__ inline void record_win(Ctx& c, int seat, const CalcOut& r) {
    GState& S = c.S;
    const PState& P = S.p[seat];
    if (c.lane == 0) {
        RmjWinResult w;
        __builtin_memset(&w, 0, sizeof(w));
        w.is_win = r.is_win; w.yakuman = r.yakuman; w.has_win_shape = r.shape;
        w.n_yaku = (uint8_t)yaku_list(r.kind, r.ym, w.yaku, 20);
        w.han = (uint32_t)r.han; w.fu = (uint32_t)r.fu;
        w.ron_agari = r.ron; w.tsumo_agari_oya = r.tsumo_oya; w.tsumo_agari_ko = r.tsumo_ko;
        w.pao_payer = -1;
        if (((r.ym >> 37) & 1ull) && P.pao37 != 0xFF) w.pao_payer = (int8_t)P.pao37;
        else if (((r.ym >> 50) & 1ull) && P.pao50 != 0xFF) w.pao_payer = (int8_t)P.pao50;
        c.E.win[(size_t)c.g * 4 + seat] = w;
    }
    S.win_mask |= (uint8_t)(1u << seat);
}

__device__ inline void emit_hora(Ctx& c, int actor, int target, const int32_t* deltas, bool tsumo, bool riichi) {
    if (c.E.skip_log) return;
    GState& S = c.S;
    RmjEvent e = ev_zero(RMJ_EV_HORA);
    e.actor = (uint8_t)actor;
    e.target = (uint8_t)target;
    for (int i = 0; i < 4; i++) e.deltas[i] = deltas[i];
    e.flags = tsumo ? 1 : 0;
    int nu = 0;
    if (riichi)
        for (int k = 0; k < S.n_dora; k++) {
            int idx = KSANMA ? 9 + 2 * k : 5 + 2 * k;
            if ((KSANMA || idx < S.live_end) && nu < 5) e.ura[nu++] = c.W[idx];
        }
    e.n_ura = (uint8_t)nu;
    emit_raw(c, e);
}

// state_3p/sanma.rs:171-204
__device__ inline void resolve_kita_rinshan(Ctx& c, int pid) {
    GState& S = c.S;
    if (S.drawable_count > 0) {
        flush_pending_kan_dora(c);
        uint8_t t = c.W[S.rinshan_count];  // draw_rinshan_tile (state_3p/wall.rs:117-124)
        S.rinshan_count += 1;
        S.drawable_count -= 1;
        PState& P = S.p[pid];
        if (P.hand_len < 14) P.hand[P.hand_len++] = t;
        S.drawn_tile = t;
        S.is_rinshan = 1;
        emit_simple(c, RMJ_EV_TSUMO, (uint8_t)pid, t);  // no new dora for kita
        S.phase = RMJ_WAIT_ACT;
        S.active_mask = (uint8_t)(1u << pid);
    }
}
// state_3p/sanma.rs:9-144.  <FAST>: part of the fast path of k_step (a Kita is a common action of 3P games: 3 % of the steps
// of a random rollout); it hands over to the full path when a seat could rob the tile (yaku evaluation).
template <bool FAST = false>
__device__ __forceinline__ void handle_kita(Ctx& c, int pid, uint64_t act) {
    GState& S = c.S;
    PState& P = S.p[pid];
    const int lane = c.lane;
    c.dirty = 0xFu;
    if (FAST && U((int)S.pending_kan_dora) > 0) { c.bail = true; return; }
    int tile;
    if (a_tile(act) != RMJ_TILE_NONE && (a_tile(act) >> 2) == 30) tile = (int)a_tile(act);
    else {
        uint64_t nb = __ballot(lane < P.hand_len && (P.hand[lane] >> 2) == 30);
        if (nb) tile = P.hand[__ffsll((long long)nb) - 1];
        else tile = a_tile(act) != RMJ_TILE_NONE ? (int)a_tile(act) : (a_n(act) ? (int)a_c(act, 0) : 0);
    }
    int idx = hand_find(c, P, tile);
    if (idx >= 0) hand_remove_at(c, P, idx);
    waits_invalidate(P);
    if (P.n_kita < 4) P.kita[P.n_kita++] = (uint8_t)tile;
    S.is_first_turn = 0;
    emit_simple(c, RMJ_EV_KITA, (uint8_t)pid, (uint8_t)tile);
    flush_pending_kan_dora(c);
    uint32_t ronners = 0;
    S.ron_offer_mask = 0;
    for (int i = 0; i < 4; i++) {
        c.X.nl[i] = 0;
        if (i == pid || i >= KNP) continue;
        PState& Q = S.p[i];
        uint64_t W = seat_waits(c, i);
        c.X.wout[i] = W;
        bool furiten = (W & Q.discard_type_mask) != 0ull || (Q.flags & (PF_MISSED_RIICHI | PF_MISSED_DOUJUN));
        if (furiten || !((W >> (tile >> 2)) & 1ull)) continue;
        if (FAST) { c.bail = true; return; }
        CalcOut r = seat_calc(c, i, -1, tile, base_cf(Q), S.honba, false, Q.n_kita);  // chankan: false (sanma.rs:106)
        if (r.is_win && (r.yakuman || r.han >= 1)) {
            ronners |= 1u << i;
            offer_ron(c, i, tile);
        }
    }
    if (ronners) {
        S.phase = RMJ_WAIT_RESPONSE;
        S.active_mask = (uint8_t)ronners;
        S.last_discard_pid = (uint8_t)pid;
        S.last_discard_tile = (uint8_t)tile;
        S.pending_kan_pid = (uint8_t)pid;
        S.pending_kan_action = act;
    } else {
        for (int p = 0; p < 4; p++) S.p[p].flags &= ~PF_IPPATSU;
        resolve_kita_rinshan(c, pid);
    }
}

// WaitAct actions other than Discard (Kyushu, Riichi, Ankan, Kakan, Tsumo, Kita): once per kyoku or rarer, so out of
// line - the hot Discard path keeps its registers and its instruction-cache footprint.  Returns bit0 = "continue with
// _resolve_discard" (Riichi declared together with a tile, unreachable through validation), bit1 = tsumogiri,
// bits 8.. = tile.
__device__ __noinline__ uint32_t ol_wait_act_other(CtxV v, int pid, uint64_t act) {
    CTX_FROM(v);
    pid = uni(pid); act = uni(act);
    GState& S = c.S;
    const int lane = c.lane;
    (void)lane;
    PState& P = S.p[pid];
    const uint32_t ty = a_type(act);
    bool do_discard = false, d_tsumogiri = false;
    int d_tile = 0;
    if (ty == RMJ_KYUSHU) {
        trigger_ryukyoku(c, RMJ_RK_KYUSHU, 0);
    } else if (ty == RMJ_RIICHI) {
        if (P.score >= 1000 && (KSANMA ? S.drawable_count > 0 : S.drawable_count >= 4) &&
            !(P.flags & (PF_RIICHI_DECLARED | PF_RIICHI_STAGE))) {
            P.flags |= PF_RIICHI_STAGE;
            emit_simple(c, RMJ_EV_REACH, (uint8_t)pid);
            if (a_tile(act) != RMJ_TILE_NONE) {  // unreachable through validation (quirk Q13), kept for parity
                int t = (int)a_tile(act);
                bool tsumogiri = S.drawn_tile != 0xFF && S.drawn_tile == t;
                P.riichi_sutehai = (uint8_t)t;
                if (!tsumogiri) P.last_tedashi = (uint8_t)t;
                int idx = hand_find(c, P, t);
                if (idx >= 0) sort_hand(c, P, P.hand_len, idx);
                do_discard = true; d_tile = t; d_tsumogiri = tsumogiri;
            }
        }
    } else if (ty == RMJ_ANKAN) {
        int tile = a_tile(act) != RMJ_TILE_NONE ? (int)a_tile(act) : (a_n(act) ? (int)a_c(act, 0) : 0);
        uint32_t ronners = 0;
        if (rule(c, RMJ_RULE_RON_ON_ANKAN_KOKUSHI)) {
            S.ron_offer_mask = 0;
            for (int i = 0; i < 4; i++) {
                c.X.nl[i] = 0;
                if (i == pid || i >= KNP) continue;
                PState& Q = S.p[i];
                if ((Q.discard_type_mask >> (tile >> 2)) & 1ull) continue;
                uint32_t cf = CF_CHANKAN | ((Q.flags & PF_RIICHI_DECLARED) ? CF_RIICHI : 0u);
                if (!seat_shape(c, i, -1, tile)) continue;
                CalcOut r = seat_calc(c, i, -1, tile, cf, 0, false);
                if (r.is_win && ((r.ym >> 42) & 1ull || (r.ym >> 49) & 1ull)) {
                    ronners |= 1u << i;
                    offer_ron(c, i, tile);
                    c.X.wout[i] = seat_waits(c, i);   // the observation of the offered seat carries its waits (like the kakan branch)
                }
            }
        }
        if (ronners) {
            S.pending_kan_pid = (uint8_t)pid;
            S.pending_kan_action = act;
            S.phase = RMJ_WAIT_RESPONSE;
            S.active_mask = (uint8_t)ronners;
            S.last_discard_pid = (uint8_t)pid;
            S.last_discard_tile = (uint8_t)tile;
        } else {
            resolve_kan(c, pid, act);
        }
    } else if (ty == RMJ_KAKAN) {
        int tile = a_tile(act) != RMJ_TILE_NONE ? (int)a_tile(act) : (a_n(act) ? (int)a_c(act, 0) : 0);
        int idx = hand_find(c, P, tile);
        if (idx >= 0) hand_remove_at(c, P, idx);
        waits_invalidate(P);
        for (int m = 0; m < P.n_melds; m++)
            if (P.meld_type[m] == RMJ_MELD_PON && (P.meld_tiles[m][0] >> 2) == (tile >> 2)) {
                P.meld_type[m] = RMJ_MELD_KAKAN;
                // insert keeping ascending ids
                uint32_t v[4] = {P.meld_tiles[m][0], P.meld_tiles[m][1], P.meld_tiles[m][2], (uint32_t)tile};
                for (int a = 0; a < 4; a++)
                    for (int b = 0; b < 3; b++)
                        if (v[b] > v[b + 1]) { uint32_t t = v[b]; v[b] = v[b + 1]; v[b + 1] = t; }
                for (int a = 0; a < 4; a++) P.meld_tiles[m][a] = (uint8_t)v[a];
                break;
            }
        emit_meld(c, RMJ_EV_KAKAN, (uint8_t)pid, 0, (uint8_t)tile, act);
        flush_pending_kan_dora(c);
        uint32_t ronners = 0;
        S.ron_offer_mask = 0;
        for (int i = 0; i < 4; i++) {
            c.X.nl[i] = 0;
            if (i == pid || i >= KNP) continue;
            PState& Q = S.p[i];
            uint64_t W = seat_waits(c, i);
            c.X.wout[i] = W;
            bool furiten = (W & Q.discard_type_mask) != 0ull || (Q.flags & (PF_MISSED_RIICHI | PF_MISSED_DOUJUN));
            if (furiten || !((W >> (tile >> 2)) & 1ull)) continue;
            uint32_t cf = base_cf(Q) | CF_CHANKAN;
            CalcOut r = seat_calc(c, i, -1, tile, cf, S.honba, false);
            if (r.is_win && (r.yakuman || r.han >= 1)) {
                ronners |= 1u << i;
                offer_ron(c, i, tile);
            }
        }
        if (ronners) {
            S.pending_kan_pid = (uint8_t)pid;
            S.pending_kan_action = act;
            S.phase = RMJ_WAIT_RESPONSE;
            S.active_mask = (uint8_t)ronners;
            S.last_discard_pid = (uint8_t)pid;
            S.last_discard_tile = (uint8_t)tile;
        } else {
            resolve_kan(c, pid, act);
        }
    } else if (ty == RMJ_TSUMO) {
        uint32_t cf = base_cf(P) | CF_TSUMO;
        if (S.drawable_count == 0 && !S.is_rinshan) cf |= CF_HAITEI;
        if (S.is_rinshan) cf |= CF_RINSHAN;
        bool no_melds = (S.p[0].n_melds | S.p[1].n_melds | S.p[2].n_melds | S.p[3].n_melds) == 0;
        if (S.is_first_turn && no_melds) cf |= CF_FIRST_TURN;  // quirk Q5 (settlement form)
        int win_tile = S.drawn_tile != 0xFF ? S.drawn_tile : 0;
        bool riichi = P.flags & PF_RIICHI_DECLARED;
        CalcOut res = seat_calc(c, pid, -1, win_tile, cf, S.honba, riichi, KSANMA ? P.n_kita : 0);
        cap_double(c, res, pid == S.oya, true, S.honba);
        if (res.is_win) {
            int32_t deltas[4] = {0, 0, 0, 0};
            int32_t total_win = 0;
            int pao_payer = -1, pao_val = 0, total_val = 0;
            if (res.yakuman) yakuman_totals(c, res, P, total_val, pao_val, pao_payer);
            if (pao_val > 0) {
                // state_3p/mod.rs:713-721: (np-1)*16000 for the dealer, 16000+(np-2)*8000 otherwise
                const int np = KNP;
                int32_t unit = pid == S.oya ? (np - 1) * 16000 : 16000 + (np - 2) * 8000;
                int32_t honba_total = (int32_t)S.honba * (np - 1) * 100;
                if (pao_payer >= 0) {
                    if (rule(c, RMJ_RULE_PAO_LIABILITY_ONLY)) {
                        int32_t pao_amt = pao_val * unit + honba_total;
                        int32_t non = total_val - pao_val;
                        deltas[pao_payer] -= pao_amt;
                        total_win += pao_amt;
                        if (non > 0)
                            for (int i = 0; i < np; i++)
                                if (i != pid) {
                                    int32_t pay = (pid == S.oya) ? non * 16000 : (i == S.oya ? non * 16000 : non * 8000);
                                    deltas[i] -= pay;
                                    total_win += pay;
                                }
                    } else {
                        int32_t full = total_val * unit + honba_total;
                        deltas[pao_payer] -= full;
                        total_win += full;
                    }
                }
            } else {
                for (int i = 0; i < KNP; i++)
                    if (i != pid) {
                        int32_t pay = (pid == S.oya) ? (int32_t)res.tsumo_ko : (i == S.oya ? (int32_t)res.tsumo_oya : (int32_t)res.tsumo_ko);
                        deltas[i] = -pay;
                        total_win += pay;
                    }
            }
            total_win += (int32_t)(S.riichi_sticks * 1000u);
            S.riichi_sticks = 0;
            deltas[pid] += total_win;
            for (int i = 0; i < 4; i++) { S.p[i].score += deltas[i]; S.p[i].score_delta = deltas[i]; }
            record_win(c, pid, res);
            emit_hora(c, pid, pid, deltas, true, riichi);
            init_next_round(c, pid == S.oya, false);
        } else {
            S.current_player = (uint8_t)((S.current_player + 1) % KNP);
            deal_next(c);
        }
    } else if (ty == RMJ_KITA && KSANMA) {
        handle_kita(c, pid, act);
    }
    return (do_discard ? 1u : 0u) | (d_tsumogiri ? 2u : 0u) | ((uint32_t)d_tile << 8);
}

// Ron settlement of a WaitResponse step (state/mod.rs:945-1142): once per kyoku at most, out of line.
__device__ __noinline__ void ol_settle_ron(CtxV v, uint32_t ron_mask) {
    CTX_FROM(v);
    ron_mask = uni(ron_mask);
    GState& S = c.S;
        if (!KSANMA && __popc(ron_mask) >= 3 && rule(c, RMJ_RULE_SANCHAHO_DRAW)) { trigger_ryukyoku(c, RMJ_RK_SANCHAHO, 0); return; }
        int target = S.last_discard_pid != 0xFF ? S.last_discard_pid : S.current_player;
        int win_tile = S.last_discard_pid != 0xFF ? S.last_discard_tile : 0;
        int32_t total_d[4] = {0, 0, 0, 0};
        bool oya_won = false, deposit_taken = false, honba_taken = false;
        for (int dist = 1; dist < KNP; dist++) {  // winners sorted by distance from the discarder (state/mod.rs:954)
            int w = (target + dist) % KNP;
            if (!((ron_mask >> w) & 1u)) continue;
            PState& Wp = S.p[w];
            uint32_t ron_honba = 0;
            if (!honba_taken) { honba_taken = true; ron_honba = S.honba; }
            uint32_t cf = base_cf(Wp);
            if (S.drawable_count == 0 && !S.is_rinshan) cf |= CF_HOUTEI;
            // a pending kita is a chankan-style claim but awards no chankan yaku (state_3p/mod.rs:896-902)
            if (S.pending_kan_pid != 0xFF && a_type(S.pending_kan_action) != RMJ_KITA) cf |= CF_CHANKAN;
            bool riichi = Wp.flags & PF_RIICHI_DECLARED;
            CalcOut res = seat_calc(c, w, -1, win_tile, cf, ron_honba, riichi, KSANMA ? Wp.n_kita : 0);
            cap_double(c, res, w == S.oya, false, ron_honba);
            if (res.is_win) {
                int32_t score = (int32_t)res.ron;
                int pao_payer = target;
                int32_t pao_amt = 0;
                if (res.yakuman) {
                    int total_val = 0, pao_val = 0, pp = -1;
                    yakuman_totals(c, res, Wp, total_val, pao_val, pp);
                    if (pp >= 0) {
                        pao_payer = pp;
                        int32_t unit = (w == S.oya) ? 48000 : 32000;
                        int32_t honba_ron = (int32_t)ron_honba * (KNP - 1) * 100;
                        int32_t split_base = rule(c, RMJ_RULE_PAO_LIABILITY_ONLY) ? pao_val * unit : total_val * unit;
                        pao_amt = split_base / 2 + honba_ron;
                    }
                }
                int32_t this_d[4] = {0, 0, 0, 0};
                this_d[w] += score;
                this_d[pao_payer] -= pao_amt;
                this_d[target] -= score - pao_amt;
                total_d[w] += score;
                total_d[pao_payer] -= pao_amt;
                total_d[target] -= score - pao_amt;
                if (!deposit_taken) {
                    int32_t sp = (int32_t)(S.riichi_sticks * 1000u);
                    total_d[w] += sp;
                    this_d[w] += sp;
                    S.riichi_sticks = 0;
                    deposit_taken = true;
                }
                if (w == S.oya) oya_won = true;
                record_win(c, w, res);
                emit_hora(c, w, target, this_d, false, riichi);
            }
        }
        for (int i = 0; i < 4; i++) { S.p[i].score += total_d[i]; S.p[i].score_delta = total_d[i]; }
        init_next_round(c, oya_won, false);
}

// ---------------------------------------------------------------- step (state/mod.rs:330-1315)
// acts_in: canonical packed actions (a_canon), RMJ_NO_ACTION for a silent seat.  `trusted`: they were taken from the
// stored legal lists by the device policy (legal by construction), so validation is skipped.
// `mine`: lane p (< 4) holds seat p's action; a seat's action is pulled out with v_readlane where it is needed instead
// of keeping four 64-bit values in scalar registers for the whole step (SGPR pressure of the hot path).
__device__ __forceinline__ uint64_t act_at(uint64_t mine, int p) {
    return (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)mine, p) |
           ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(mine >> 32), p) << 32);
}
template <bool FAST = false>
__device__ __forceinline__ void step_game(Ctx& c, const uint64_t mine, bool trusted = false) {
    GState& S = c.S;
    const int lane = c.lane;
    if (U((int)S.is_done)) return;
    S.step_count += 1;
    const int phase = U((int)S.phase);
    // ---- validation against the stored legal lists
    for (int pid = 0; pid < 4; pid++) {
        if (trusted) continue;
        const uint64_t a_pid = act_at(mine, pid);
        if (a_pid == RMJ_NO_ACTION) continue;
        int n = U((int)S.nlegal[pid]);
        // whose list counts: the reference regenerates the sender's legal actions (state/mod.rs:339-402) - in WaitAct those of the
        // current player whether active_players names it or not (a poked state may not: tests/test_riichi_autoplay.py; its list is
        // kept by finalize_outputs without being published), in WaitResponse those of the seats that were offered something
        bool active = phase == RMJ_WAIT_ACT ? pid == U((int)S.current_player) : (((U((uint32_t)S.active_mask) >> pid) & 1u) != 0u);
        bool valid;
        if (!active || n == 0) {
            // _get_legal_actions_internal for any other seat: [] in WaitAct, [Pass] in WaitResponse
            valid = phase == RMJ_WAIT_RESPONSE && a_match(mk_action(RMJ_PASS, RMJ_TILE_NONE, 0), a_pid);
        } else {
            bool hit = lane < n && a_match(c.Lg[pid * RMJ_MAX_LEGAL + lane], a_pid);
            valid = __ballot(hit) != 0ull;
        }
        if (!valid) {
            if (FAST) { c.bail = true; return; }
            S.last_error_pid = (uint8_t)pid;
            trigger_ryukyoku(c, RMJ_RK_ILLEGAL, pid);
            return;
        }
    }
    PROF(c.X, lane, 2);
    if (phase == RMJ_WAIT_ACT) {
        const int pid = U((int)S.current_player);
        const uint64_t act = act_at(mine, pid);
        if (act == RMJ_NO_ACTION) return;
        PState& P = S.p[pid];
        const uint32_t ty = a_type(act);
        // The Discard branch and the (validation-unreachable) Riichi-with-tile branch both end in _resolve_discard;
        // they share ONE inlined copy of it below (code size = instruction-cache footprint of the hot path).
        bool do_discard = false, d_tsumogiri = false;
        int d_tile = 0;
        if (ty == RMJ_DISCARD) {
            if (a_tile(act) == RMJ_TILE_NONE) return;
            if (FAST) c.dirty = 1u << pid;  // the discard path tracks the seats it touches (resolve_discard, claims, next draw)
            int tile = (int)a_tile(act);
            bool tsumogiri = false, valid = false;
            const int drawn = U((int)S.drawn_tile);
            if (drawn != 0xFF && drawn == tile) { tsumogiri = true; valid = true; }
            PROF(c.X, lane, 3);
            int idx = hand_find(c, P, tile);
            PROF(c.X, lane, 20);
            if (idx >= 0) {
                discard_sort(c, P, U((int)P.hand_len), idx);  // hand.remove(idx); hand.sort()
                PROF(c.X, lane, 22);
                valid = true;
            }
            do_discard = valid; d_tile = tile; d_tsumogiri = tsumogiri;
        } else if (FAST && KSANMA && ty == RMJ_KITA) {
            handle_kita<true>(c, pid, act);
            return;
        } else {
            if (FAST) { c.bail = true; return; }  // Riichi, kans, Tsumo, Kyushu: full path
            uint32_t r = ol_wait_act_other(ctx_pack(c), pid, act);
            do_discard = r & 1u; d_tsumogiri = (r >> 1) & 1u; d_tile = (int)(r >> 8);
        }
        PROF(c.X, lane, 3);
        if (do_discard) resolve_discard<FAST>(c, pid, d_tile, d_tsumogiri);
        return;
    }
    // ---- WaitResponse (state/mod.rs:900-1314)
#if defined(RMJ_PROFILE) || defined(RMJ_CUTS)
    struct ProfTail { Ctx& c; __device__ ~ProfTail() { PROF(c.X, c.lane, 7); } } prof_tail{c};
#endif
    // lane = seat (active_players is in ascending seat order, state/mod.rs:1378-1395)
    const bool has = lane < 4 && mine != RMJ_NO_ACTION;
    const uint32_t my_ty = a_type(mine);
    const uint32_t act_m = S.active_mask;
    const bool is_act = has && ((act_m >> lane) & 1u);
    const uint32_t roned = (uint32_t)__ballot(has && my_ty == RMJ_RON) & 0xFu;
    if (lane < 4 && (((uint32_t)S.ron_offer_mask & ~roned) >> lane) & 1u) {  // a Ron offer that was not taken
        uint32_t fl = S.p[lane].flags | PF_MISSED_DOUJUN;
        if (fl & PF_RIICHI_DECLARED) fl |= PF_MISSED_RIICHI;
        S.p[lane].flags = (uint8_t)fl;
    }
    const uint32_t ron_mask = roned & act_m;
    const uint32_t pon_m = (uint32_t)__ballot(is_act && (my_ty == RMJ_PON || my_ty == RMJ_DAIMINKAN)) & 0xFu;
    const uint32_t chi_m = KSANMA ? 0u : ((uint32_t)__ballot(is_act && my_ty == RMJ_CHI) & 0xFu);
    // the first Pon/Daiminkan wins over a Chi; among equals the lower seat (the reference's replace-if-better scan)
    const int claimer = pon_m ? __ffs((int)pon_m) - 1 : (chi_m ? __ffs((int)chi_m) - 1 : -1);
    const uint64_t claim = claimer >= 0 ? act_at(mine, claimer) : 0ull;
    if (ron_mask) {
        if (FAST) { c.bail = true; return; }
        ol_settle_ron(ctx_pack(c), ron_mask);
    } else if (claimer >= 0) {
        PState& C = S.p[claimer];
        accept_riichi(c);
        S.is_rinshan = 0;
        S.is_first_turn = 0;
        C.flags &= ~PF_MISSED_DOUJUN;
        const int ldp = U((int)S.last_discard_pid);
        if (ldp != 0xFF) S.p[ldp].flags &= ~PF_NAGASHI;
        for (int p = 0; p < 4; p++) S.p[p].flags &= ~PF_IPPATSU;
        uint32_t ty = a_type(claim);
        if (FAST && ty == RMJ_DAIMINKAN) { c.bail = true; return; }
        if (ty == RMJ_DAIMINKAN) {
            S.current_player = (uint8_t)claimer;
            S.active_mask = (uint8_t)(1u << claimer);
            C.n_forbidden = 0;
            resolve_kan(c, claimer, claim);
            return;
        }
        hand_remove_tiles(c, C, claim);
        int discarder = U((int)S.last_discard_pid), tile = U((int)S.last_discard_tile);
        push_meld(C, ty == RMJ_PON ? RMJ_MELD_PON : RMJ_MELD_CHI, a_c(claim, 0), a_c(claim, 1), (uint32_t)tile, 0, 3, discarder, tile);
        emit_meld(c, ty == RMJ_PON ? RMJ_EV_PON : RMJ_EV_CHI, (uint8_t)claimer, (uint8_t)discarder, (uint8_t)tile, claim);
        if (ty == RMJ_PON) pao_check(c, claimer, discarder, tile);
        S.current_player = (uint8_t)claimer;
        S.phase = RMJ_WAIT_ACT;
        S.active_mask = (uint8_t)(1u << claimer);
        C.n_forbidden = 0;
        if (ty == RMJ_PON) {
            C.forbidden[0] = (uint8_t)tile;
            C.n_forbidden = 1;
        } else {
            C.forbidden[0] = (uint8_t)tile;
            C.n_forbidden = 1;
            int t34 = tile >> 2;
            int x = (int)a_c(claim, 0) >> 2, y = (int)a_c(claim, 1) >> 2;
            int lo = min(x, y), hi = max(x, y);
            if (lo == t34 + 1 && hi == t34 + 2) {
                if (t34 % 9 <= 5) { C.forbidden[1] = (uint8_t)((t34 + 3) * 4); C.n_forbidden = 2; }
            } else if (t34 >= 2 && hi == t34 - 1 && lo == t34 - 2 && t34 % 9 >= 3) {
                C.forbidden[1] = (uint8_t)((t34 - 3) * 4);
                C.n_forbidden = 2;
            }
        }
        S.needs_tsumo = 0;
        S.drawn_tile = 0xFF;
    } else {
        S.active_mask = 0;
        S.ron_offer_mask = 0;
        for (int p = 0; p < 4; p++) S.stale_n[p] = 0;  // current_claims.clear() (state/mod.rs:1299)
        if (FAST && S.pending_kan_pid != 0xFF) { c.bail = true; return; }
        if (S.pending_kan_pid != 0xFF) {
            int pk = S.pending_kan_pid;
            uint64_t pa = S.pending_kan_action;
            S.pending_kan_pid = 0xFF;
            S.pending_kan_action = 0;
            if (a_type(pa) == RMJ_KITA) {  // state_3p/mod.rs:1201-1209
                for (int p = 0; p < 4; p++) S.p[p].flags &= ~PF_IPPATSU;
                resolve_kita_rinshan(c, pk);
            } else {
                resolve_kan(c, pk, pa);
            }
        } else {
            accept_riichi(c);
            S.turn_count += 1;
            S.current_player = (uint8_t)((S.current_player + 1) % KNP);
            deal_next<FAST>(c);
            if (S.turn_count >= (uint32_t)KNP) S.is_first_turn = 0;
        }
    }
}

// ---------------------------------------------------------------- MJAI event ingestion (row N1)
// apply_mjai_event: state/event_handler.rs:18-330 (4P), state_3p/event_handler.rs:18-362 (3P).  `ev` = up to three
// binary records of one game (a start_kyoku is START_KYOKU + two TEHAI records, like the emitted log).  This is the
// reference's replay state machine, NOT step(): no validation, no wall (a zero placeholder), melds keep from_who = -1,
// discards carry no tsumogiri/riichi flags, is_first_turn is never cleared, scoring never happens.
// Deviation: meld tiles are stored sorted (the reference keeps [called, consumed...] order on this path).
__device__ inline void apply_remove_first(Ctx& c, PState& P, int tile) {
    int idx = hand_find(c, P, tile);
    if (idx >= 0) hand_remove_at(c, P, idx);
}
// current_claims of a discard / kita (event_handler.rs:129-155, state_3p/event_handler.rs:320-352)
__device__ inline void apply_claims(Ctx& c, int actor, int tile, bool ron_only) {
    GState& S = c.S;
    uint32_t claim_active = 0;
    S.ron_offer_mask = 0;
    for (int i = 0; i < 4; i++) {
        c.X.nl[i] = 0;
        c.X.wout[i] = 0;
        S.stale_n[i] = 0;
    }
    for (int i = 0; i < KNP; i++) {
        if (i == actor) continue;
        bool any = gen_claims(c, i, actor, tile, false);  // the `missed` result is dropped on this path
        if (ron_only) {  // kita: only Ron survives the filter (no Pass either)
            wave_sync();
            any = any && a_type(c.X.legal[i][0]) == RMJ_RON;
            c.X.nl[i] = any ? 1 : 0;
            S.stale_n[i] = (uint8_t)(any ? 1 : 0);
        }
        if (any) claim_active |= 1u << i;
    }
    if (claim_active) {
        S.phase = RMJ_WAIT_RESPONSE;
        S.active_mask = (uint8_t)claim_active;
    } else {
        S.phase = RMJ_WAIT_ACT;
        S.active_mask = 0;
        S.current_player = 0xFF;
    }
}
__device__ inline void apply_event(Ctx& c, const RmjEvent* ev) {
    GState& S = c.S;
    const int lane = c.lane;
    const uint32_t ty = ev[0].type;
    const int actor = ev[0].actor & 3;
    const int tile = ev[0].tile;
    PState& P = S.p[actor];
    // Replay semantics (RMJ_EVF_REPLAY_PASS in `pad`; KyokuStepIterator::_collect_pass_observations, replay/mod.rs:129-177): a
    // seat that was offered Ron on the last discard and does not win with this event has passed - same-turn furiten,
    // permanent in riichi.
    if ((ev[0].pad & 1u) && S.phase == RMJ_WAIT_RESPONSE && S.pending_kan_pid == 0xFF) {
        uint32_t m = (uint32_t)S.ron_offer_mask & (uint32_t)S.active_mask & 0xFu;
        if (ty == RMJ_EV_HORA) m &= ~(1u << actor);
        for (int i = 0; i < KNP; i++)
            if ((m >> i) & 1u) {
                PState& Q = S.p[i];
                Q.flags |= PF_MISSED_DOUJUN | ((Q.flags & PF_RIICHI_DECLARED) ? PF_MISSED_RIICHI : 0);
            }
    }
    // ... and the walker's discard (apply_log_action, state/event_handler.rs:391-392) ends the discarder's same-turn furiten
    if ((ev[0].pad & 1u) && ty == RMJ_EV_DAHAI) P.flags &= ~PF_MISSED_DOUJUN;
    // ... and the walker knows a replacement draw (apply_log_action: is_after_kan, state/event_handler.rs:415, :428-430, :565, :658,
    // state_3p/event_handler.rs:606 - kita too): the tile dealt after a kan is a rinshan draw, any other deal is not
    if (ty == RMJ_EV_START_KYOKU) S.replay_after_kan = 0;
    if (ev[0].pad & 1u) {
        if (ty == RMJ_EV_DAHAI || ty == RMJ_EV_PON || ty == RMJ_EV_CHI) S.replay_after_kan = 0;
        else if (ty == RMJ_EV_DAIMINKAN || ty == RMJ_EV_ANKAN || ty == RMJ_EV_KAKAN || (KSANMA && ty == RMJ_EV_KITA)) S.replay_after_kan = 1;
        else if (ty == RMJ_EV_TSUMO) { S.is_rinshan = S.replay_after_kan; S.replay_after_kan = 0; }
    }
    switch (ty) {
        case RMJ_EV_START_GAME:  // env.rs:56-72 + event_handler.rs:20-25
            S.ev_base = S.ev_count;
            S.current_player = 0xFF;
            S.active_mask = 0;
            break;
        case RMJ_EV_START_KYOKU: {  // event_handler.rs:26-103
            S.honba = ev[0].consumed[1];
            S.riichi_sticks = (uint32_t)ev[0].consumed[2] | ((uint32_t)ev[0].consumed[3] << 8);
            S.round_wind = ev[0].consumed[0] & 3;
            S.oya = ev[0].actor;
            S.kyoku_idx = ev[0].target ? (uint8_t)(ev[0].target - 1) : 0;
            S.current_player = 0xFF;
            S.turn_count = 0;
            S.is_done = 0;
            S.needs_tsumo = 1;
            S.phase = RMJ_WAIT_ACT;
            S.active_mask = 0;
            S.last_discard_pid = 0xFF; S.last_discard_tile = 0;
            S.ron_offer_mask = 0;
            S.pending_kan_pid = 0xFF; S.pending_kan_action = 0;
            S.is_rinshan = 0;
            S.is_first_turn = 1;
            S.riichi_pending = 0xFF;
            S.drawn_tile = 0xFF;
            S.last_error_pid = 0xFF;
            const int total = KSANMA ? 108 : 136;
            S.wall_total = (uint8_t)total;
            S.live_end = (uint8_t)(total - 13 * KNP);
            S.rinshan_count = 0;
            S.pending_kan_dora = 0;
            S.drawable_count = (uint8_t)(S.live_end - 14);
            S.n_dora = 1;
            S.dora[0] = (uint8_t)tile;
            S.win_mask = 0;
            for (int i = lane; i < RMJ_WALL_STRIDE / 4; i += 64) reinterpret_cast<uint32_t*>(c.W)[i] = 0u;  // placeholder wall
            S.wall_meta = 0;   // wall_digest.clear(); salt.clear() (event_handler.rs:81-82)
            for (int p = 0; p < 4; p++) {  // PlayerState::reset_round, state/player.rs:66-86
                PState& Q = S.p[p];
                Q.hand_len = 0; Q.n_melds = 0; Q.n_discards = 0;
                Q.flags = PF_NAGASHI;
                Q.pao37 = 0xFF; Q.pao50 = 0xFF;
                Q.n_forbidden = 0;
                Q.riichi_decl_idx = 0xFF; Q.riichi_sutehai = 0xFF; Q.last_tedashi = 0xFF;
                Q.score_delta = 0;
                Q.discard_from_hand_bits = 0; Q.discard_is_riichi_bits = 0;
                Q.discard_type_mask = 0;
                Q.n_kita = 0;
                S.stale_n[p] = 0;
                if (p < KNP) Q.score = ev[0].deltas[p];
            }
            wave_sync();
            if (lane < 52) {  // lane = 13*seat + slot; records 1/2 carry seats 0,1 / 2,3 (26 payload bytes each)
                const int seat = lane / 13, slot = lane - 13 * seat;
                const uint8_t* pl = reinterpret_cast<const uint8_t*>(&ev[1 + (seat >> 1)]) + 4;
                if (seat < KNP) S.p[seat].hand[slot] = pl[13 * (seat & 1) + slot];
            }
            wave_sync();
            for (int p = 0; p < KNP; p++) {
                S.p[p].hand_len = 13;
                sort_hand(c, S.p[p], 13);
            }
            break;
        }
        case RMJ_EV_TSUMO:  // :104-118 (the whole hand is re-sorted: the drawn tile does not stay last on this path)
            S.current_player = (uint8_t)actor;
            S.drawn_tile = (uint8_t)tile;
            waits_invalidate(P);
            if (P.hand_len < 14) P.hand[P.hand_len++] = (uint8_t)tile;
            wave_sync();
            sort_hand(c, P, P.hand_len);
            P.n_forbidden = 0;
            if (S.live_end > S.rinshan_count) {
                S.live_end -= 1;
                if (S.drawable_count > 0) S.drawable_count -= 1;
            }
            S.phase = RMJ_WAIT_ACT;
            S.active_mask = (uint8_t)(1u << actor);
            S.needs_tsumo = 0;
            break;
        case RMJ_EV_DAHAI: {  // :119-156
            S.current_player = (uint8_t)actor;
            apply_remove_first(c, P, tile);
            waits_invalidate(P);
            if (P.n_discards < RMJ_MAX_DISCARDS) P.discards[P.n_discards++] = (uint8_t)tile;
            P.discard_type_mask |= 1ull << (tile >> 2);
            S.last_discard_pid = (uint8_t)actor;
            S.last_discard_tile = (uint8_t)tile;
            S.drawn_tile = 0xFF;
            if (P.flags & PF_RIICHI_STAGE) P.flags = (uint8_t)((P.flags | PF_RIICHI_DECLARED) & ~PF_RIICHI_STAGE);
            wave_sync();
            apply_claims(c, actor, tile, false);
            S.needs_tsumo = 1;
            break;
        }
        case RMJ_EV_PON:
        case RMJ_EV_CHI: {  // :157-238 ; 3P chi: no kuikae bookkeeping (state_3p/event_handler.rs:195-222)
            S.current_player = (uint8_t)actor;
            const int c1 = ev[0].consumed[0], c2 = ev[0].consumed[1];
            apply_remove_first(c, P, c1);
            apply_remove_first(c, P, c2);
            waits_invalidate(P);
            push_meld(P, ty == RMJ_EV_PON ? RMJ_MELD_PON : RMJ_MELD_CHI, (uint32_t)tile, (uint32_t)c1, (uint32_t)c2, 0, 3, 0xFF, tile);
            S.drawn_tile = 0xFF;
            S.phase = RMJ_WAIT_ACT;
            S.active_mask = (uint8_t)(1u << actor);
            S.needs_tsumo = 0;
            if (ty == RMJ_EV_PON || !KSANMA) {
                P.n_forbidden = 0;
                if (rule(c, RMJ_RULE_KUIKAE_FORBIDDEN)) {
                    P.forbidden[0] = (uint8_t)tile;
                    P.n_forbidden = 1;
                    if (ty == RMJ_EV_CHI) {
                        int t34 = tile >> 2, a = c1 >> 2, b = c2 >> 2;
                        int lo = min(a, b), hi = max(a, b);
                        if (lo == t34 + 1 && hi == t34 + 2) {
                            if (t34 % 9 <= 5) { P.forbidden[1] = (uint8_t)((t34 + 3) * 4); P.n_forbidden = 2; }
                        } else if (t34 >= 2 && hi == t34 - 1 && lo == t34 - 2 && t34 % 9 >= 3) {
                            P.forbidden[1] = (uint8_t)((t34 - 3) * 4);
                            P.n_forbidden = 2;
                        }
                    }
                }
            }
            break;
        }
        case RMJ_EV_DAIMINKAN: {  // :239-268
            S.current_player = (uint8_t)actor;
            const int n = (ev[0].flags >> 4) & 15;
            for (int k = 0; k < n && k < 3; k++) apply_remove_first(c, P, ev[0].consumed[k]);
            waits_invalidate(P);
            push_meld(P, RMJ_MELD_DAIMINKAN, (uint32_t)tile, ev[0].consumed[0], ev[0].consumed[1], ev[0].consumed[2], 4, 0xFF, tile);
            S.phase = RMJ_WAIT_ACT;
            S.active_mask = (uint8_t)(1u << actor);
            S.needs_tsumo = 1;
            break;
        }
        case RMJ_EV_ANKAN: {  // :269-289
            const int n = (ev[0].flags >> 4) & 15;
            for (int k = 0; k < n && k < 4; k++) apply_remove_first(c, P, ev[0].consumed[k]);
            waits_invalidate(P);
            push_meld(P, RMJ_MELD_ANKAN, ev[0].consumed[0], ev[0].consumed[1], ev[0].consumed[2], ev[0].consumed[3], 4, 0xFF, 0xFF);
            S.current_player = (uint8_t)actor;
            S.phase = RMJ_WAIT_ACT;
            S.active_mask = (uint8_t)(1u << actor);
            S.needs_tsumo = 1;
            break;
        }
        case RMJ_EV_KAKAN: {  // :290-305
            apply_remove_first(c, P, tile);
            waits_invalidate(P);
            for (int m = 0; m < P.n_melds; m++)
                if (P.meld_type[m] == RMJ_MELD_PON && (P.meld_tiles[m][0] >> 2) == (tile >> 2)) {
                    P.meld_type[m] = RMJ_MELD_KAKAN;
                    uint32_t v[4] = {P.meld_tiles[m][0], P.meld_tiles[m][1], P.meld_tiles[m][2], (uint32_t)tile};
                    for (int a = 0; a < 4; a++)
                        for (int b = 0; b < 3; b++)
                            if (v[b] > v[b + 1]) { uint32_t t = v[b]; v[b] = v[b + 1]; v[b + 1] = t; }
                    for (int a = 0; a < 4; a++) P.meld_tiles[m][a] = (uint8_t)v[a];
                    break;
                }
            S.current_player = (uint8_t)actor;
            S.phase = RMJ_WAIT_ACT;
            S.active_mask = (uint8_t)(1u << actor);
            S.needs_tsumo = 1;
            break;
        }
        case RMJ_EV_REACH: P.flags |= PF_RIICHI_STAGE; break;  // :306-310
        case RMJ_EV_REACH_ACCEPTED:                             // :311-315
            P.flags |= PF_RIICHI_DECLARED;
            S.riichi_sticks += 1;
            P.score -= 1000;
            break;
        case RMJ_EV_DORA:  // :316-319
            if (S.n_dora < 5) S.dora[S.n_dora++] = (uint8_t)tile;
            break;
        case RMJ_EV_KITA: {  // 4P: ignored; 3P: state_3p/event_handler.rs:308-357
            if (!KSANMA) break;
            uint64_t nb = __ballot(lane < P.hand_len && (P.hand[lane] >> 2) == 30);
            S.current_player = (uint8_t)actor;
            for (int i = 0; i < 4; i++) { c.X.nl[i] = 0; c.X.wout[i] = 0; S.stale_n[i] = 0; }
            S.ron_offer_mask = 0;
            S.active_mask = 0;
            if (nb) {
                int idx = __ffsll((long long)nb) - 1;
                int kt = P.hand[idx];
                hand_remove_at(c, P, idx);
                waits_invalidate(P);
                if (P.n_kita < 4) P.kita[P.n_kita++] = (uint8_t)kt;
                wave_sync();
                apply_claims(c, actor, kt, true);
            } else {
                S.phase = RMJ_WAIT_ACT;
                S.current_player = 0xFF;
            }
            S.needs_tsumo = 1;
            break;
        }
        case RMJ_EV_HORA:
        case RMJ_EV_RYUKYOKU:
        case RMJ_EV_END_KYOKU: S.is_done = 1; break;  // :323-325
        default: break;
    }
    wave_sync();
}

// After a transition: produce the observation-side outputs for the new state
// (get_observations(active_players), env.rs:870-871 -> state/mod.rs:189-263; mask: observation/python.rs:98-111)
template <bool FAST = false>
__device__ __forceinline__ void finalize_outputs(Ctx& c, bool claims_fresh, bool observe = true, bool all_rows = false) {
    GState& S = c.S;
    const int lane = c.lane;
    const int phase = U((int)S.phase);
    if (!FAST) S.tp_seat = 0xFF;   // whatever the rich tier 0 cached for a riichi-stage list (GState::tp_*) is void after a full-path publication
    if (U((int)S.is_done)) {
        for (int p = 0; p < 4; p++) { c.X.nl[p] = 0; c.X.wout[p] = 0; }
    } else if (!FAST && U((int)S.active_mask) == 0) {
        // nobody to act (only reachable through rmj_apply_events: between a discard nobody can claim and the next tsumo)
        for (int p = 0; p < 4; p++) { c.X.nl[p] = 0; c.X.wout[p] = 0; }
    } else if (phase == RMJ_WAIT_ACT) {
        for (int p = 0; p < 4; p++) { c.X.nl[p] = 0; c.X.wout[p] = 0; }
        gen_act_legal<FAST>(c, U((int)S.current_player));
        if (FAST && c.bail) return;
        if (!FAST) {
            // a poked state may list seats as active that are not the current player (tests/test_riichi_autoplay.py sets
            // current_player without active_players): their observation has no actions but still carries their waits
            const uint32_t others = U((uint32_t)S.active_mask) & ~(1u << U((int)S.current_player)) & 0xFu;
            for (int p = 0; p < 4; p++)
                if ((others >> p) & 1u) c.X.wout[p] = seat_waits(c, p);
        }
    } else if (!claims_fresh) {
        // WaitResponse that was not produced in this launch (e.g. after rmj_poke_state): rebuild claims
        if (S.pending_kan_pid == 0xFF && S.last_discard_pid != 0xFF) {
            uint32_t want = S.active_mask, got = 0;
            S.ron_offer_mask = 0;
            for (int i = 0; i < 4; i++) {
                c.X.nl[i] = 0;
                c.X.wout[i] = 0;
                if (i == S.last_discard_pid || !((want >> i) & 1u)) continue;
                if (gen_claims(c, i, S.last_discard_pid, S.last_discard_tile)) got |= 1u << i;
            }
            (void)got;
        }
    }
    wave_sync();
    PROF(c.X, lane, 13);
    TLF(c, 10);        // (finalize) the acting seat's list
    // masks + list publication (only the seats that act have a list; a step usually has one).  Mask rows of seats that
    // are not to act are all zero and stay so: only the rows of the seats that had a list before this step (S.nlegal
    // still holds the previous publication) or have one now are rewritten - 82 B per row in 16-bit units.
    const uint32_t am = (uint32_t)__builtin_amdgcn_readfirstlane((int)S.active_mask);
    // a current player that active_players does not name (poked states only): its list is stored for the validation of what it sends,
    // like the reference regenerates it on demand, but nothing of it is published (its seat has no observation)
    const uint32_t hidden = (!FAST && phase == RMJ_WAIT_ACT && !U((int)S.is_done) && am != 0u) ? ((1u << U((int)S.current_player)) & ~am & 0xFu) : 0u;
    // (all_rows: the last step of a fused rollout - its quiet steps published no mask rows, so every row is rewritten)
    const uint32_t rows = all_rows ? 0xFu : ((((uint32_t)__ballot(lane < 4 && S.nlegal[lane & 3] != 0)) | am) & 0xFu);
    for (int i = lane; i < (4 * 82 + 3) / 4; i += 64) reinterpret_cast<uint32_t*>(c.X.maskbuf)[i] = 0u;
    wave_sync();
    for (uint32_t m = am | hidden; m; m &= m - 1u) {
        const int p = __ffs((int)m) - 1;
        const int n = U(c.X.nl[p]);
        if (lane < n) {
            uint64_t a = c.X.legal[p][lane];
            c.Lg[p * RMJ_MAX_LEGAL + lane] = a;
            int id = KSANMA ? a_encode_3p(a) : a_encode(a);  // 60 ids in 3P (observation_3p/python.rs:100-112)
            if (id >= 0 && id < (KSANMA ? 60 : 82) && ((am >> p) & 1u)) c.X.maskbuf[p * 82 + id] = 1;
        }
    }
    wave_sync();
    uint16_t* mout = reinterpret_cast<uint16_t*>(c.E.mask + (size_t)c.g * 328);
    for (uint32_t m = rows; m; m &= m - 1u) {
        const int p = __ffs((int)m) - 1;
        if (lane < 41) mout[41 * p + lane] = reinterpret_cast<const uint16_t*>(c.X.maskbuf)[41 * p + lane];
    }
    if (lane < 4) {
        const bool acts = (am >> lane) & 1u;
        const int n = acts ? c.X.nl[lane] : 0;
        c.E.nlegal[(size_t)c.g * 4 + lane] = (uint8_t)n;
        S.nlegal[lane] = ((hidden >> lane) & 1u) ? (uint8_t)c.X.nl[lane] : (uint8_t)n;
        c.E.waits[(size_t)c.g * 4 + lane] = acts ? c.X.wout[lane] : 0ull;
        if (observe && acts && !S.is_done) {  // get_observation advances the seat's event cursor (state/mod.rs:211-218)
            S.obs_from[lane] = S.obs_upto[lane];
            S.obs_upto[lane] = S.ev_count;
        }
    }
    if (lane == 0) c.E.status[c.g] = (uint32_t)S.active_mask | ((uint32_t)S.phase << 8) | ((uint32_t)S.is_done << 16);
    PROF(c.X, lane, 14);
}

}  // namespace RMJ_NS
