// Sequence (transformer) features of an Observation, row N3: observation/sequence_features.rs (4P only, like the
// reference).  One wave per game.  The reference computes them from the MJAI strings an Observation carries; here the
// same quantities are read from the binary event ring and the record, over the events of the CURRENT ROUND (from the
// last start_kyoku on) - the progression is exactly GameState::round_seq_progression (state/mod.rs:73, 1781, 2150-2161),
// the snapshot the reference attaches with enable_seq_caching (state/mod.rs:257-260).
#pragma once
#include "rmj_common.hip.h"

namespace rmj {

// sequence_features.rs:46-72
__device__ __forceinline__ uint32_t seq_kan37(uint32_t t) {
    if (t == 16u) return 0u;
    if (t == 52u) return 10u;
    if (t == 88u) return 20u;
    const uint32_t tt = t >> 2;
    return tt <= 8u ? tt + 1u : (tt <= 17u ? tt + 2u : (tt <= 33u ? tt + 3u : 0u));
}
__device__ __forceinline__ bool seq_red(uint32_t t) { return t == 16u || t == 52u || t == 88u; }
// sequence_features.rs:93-133 (c0, c1 = the two tiles from the hand)
__device__ __forceinline__ uint32_t seq_chi(uint32_t c0, uint32_t c1, uint32_t called) {
    const uint32_t lo = min(min(c0, c1), called) >> 2;
    const uint32_t suit = lo / 9u, seq_start = lo - suit * 9u;
    const uint32_t call_pos = (called >> 2) - suit * 9u - seq_start;
    const bool has_red = seq_red(c0) || seq_red(c1) || seq_red(called);
    const bool five = seq_start <= 4u && 4u <= seq_start + 2u;
    // sequences starting at rank 2, 3, 4 (0-based) contain the five and own six slots, the others three
    const uint32_t offset = 3u * seq_start + 3u * (seq_start > 2u ? min(seq_start - 2u, 3u) : 0u);
    return suit * 30u + offset + ((five && has_red) ? 3u + call_pos : call_pos);
}
// sequence_features.rs:144-184
__device__ __forceinline__ uint32_t seq_pon(uint32_t c0, uint32_t c1, uint32_t called) {
    const uint32_t ct = called >> 2, suit = ct / 9u;
    if (suit == 3u) return 33u + (ct - 27u);
    const uint32_t rank = ct - suit * 9u;
    if (rank == 4u) return suit * 11u + 4u + (seq_red(called) ? 2u : ((seq_red(c0) || seq_red(c1)) ? 1u : 0u));
    return suit * 11u + (rank < 4u ? rank : rank + 2u);
}
__device__ __forceinline__ uint32_t seq_rel(uint32_t actor, uint32_t target) { return (target + 3u - actor + 4u) & 3u; }  // :188-190

// State of a forward scan over a range of the binary event ring (the reference walks the MJAI strings of
// Observation.events): progression entries (sequence_features.rs:213-314), the actor of the last dahai / kakan (:825-835),
// the pending-reach rule, the range's start_kyoku, and - for one seat - get_drawn_tile (:410-435: the seat's last tsumo
// with no dahai / chi / pon / daiminkan after it; forward form: a stop event clears it, an own tsumo sets it).
struct SeqScan {
    uint32_t n_prog = 0;
    int last_da = -1;
    int pending = -1;
    int64_t sk_idx = -1;     // ring index of the first start_kyoku in the range
    int drawn = -1;          // tile id of the drawn-tile token, -1 = none
};
// scans events [from, to); writes at most `cap` progression tuples to `prog`; seat < 0: no drawn-tile tracking
__device__ __forceinline__ void seq_scan_range(const RmjEvent* ring, uint32_t mask, int64_t from, int64_t to, int lane, uint16_t* prog,
                                               uint32_t cap, int seat, SeqScan& q) {
    for (int64_t base = from; base < to; base += 64) {
        const int64_t idx = base + lane;
        uint32_t ty = RMJ_EV_NONE, actor = 0, target = 0, tile = 0, c0 = 0, c1 = 0, fl = 0;
        if (idx < to) {
            const RmjEvent& e = ring[(uint32_t)idx & mask];
            ty = e.type; actor = e.actor & 3u; target = e.target & 3u; tile = e.tile; c0 = e.consumed[0]; c1 = e.consumed[1]; fl = e.flags;
        }
        const bool is_reach = ty == RMJ_EV_REACH, is_dahai = ty == RMJ_EV_DAHAI;
        // liqi: the dahai consumes a pending reach of its own actor (:240-245).  pending before this event = actor of the
        // last reach before it, unless that actor has discarded since; before the first reach of the chunk: the carry.
        const uint64_t lt = lanemask_lt(lane);
        uint64_t reach_of[4], dahai_of[4];
        for (int a = 0; a < 4; a++) {
            reach_of[a] = __ballot(is_reach && (int)actor == a);
            dahai_of[a] = __ballot(is_dahai && (int)actor == a);
        }
        const uint64_t reach_any = reach_of[0] | reach_of[1] | reach_of[2] | reach_of[3];
        uint32_t liqi = 0;
        if (is_dahai) {
            const uint64_t rb = reach_any & lt;
            if (rb) {
                const int j = 63 - __clzll((long long)rb);
                const bool mine = (reach_of[actor] >> j) & 1ull;
                const uint64_t between = lt & ~((2ull << j) - 1ull);  // lanes j+1 .. lane-1
                liqi = mine && (dahai_of[actor] & between) == 0ull;
            } else {
                liqi = q.pending == (int)actor && (dahai_of[actor] & lt) == 0ull;
            }
        }
        {   // carry out of the chunk
            if (reach_any) {
                const int j = 63 - __clzll((long long)reach_any);
                int a = 0;
                for (int k = 1; k < 4; k++)
                    if ((reach_of[k] >> j) & 1ull) a = k;
                const uint64_t after = j == 63 ? 0ull : ~((2ull << j) - 1ull);
                q.pending = (dahai_of[a] & after) ? -1 : a;
            } else if (q.pending >= 0 && dahai_of[q.pending]) {
                q.pending = -1;
            }
        }
        uint32_t e0 = 0, e1 = 0, e2 = 2, e3 = 2, e4 = 4;
        bool emit = true;
        switch (ty) {
            case RMJ_EV_START_KYOKU: e0 = 4; e1 = 0; break;
            case RMJ_EV_DAHAI: e0 = actor; e1 = 1u + seq_kan37(tile); e2 = fl & 1u; e3 = liqi; break;
            case RMJ_EV_CHI: e0 = actor; e1 = 38u + seq_chi(c0, c1, tile); e4 = seq_rel(actor, target); break;
            case RMJ_EV_PON: e0 = actor; e1 = 128u + seq_pon(c0, c1, tile); e4 = seq_rel(actor, target); break;
            case RMJ_EV_DAIMINKAN: e0 = actor; e1 = 168u + seq_kan37(tile); e4 = seq_rel(actor, target); break;
            case RMJ_EV_ANKAN: e0 = actor; e1 = 205u + (c0 >> 2); break;
            case RMJ_EV_KAKAN: e0 = actor; e1 = 239u + seq_kan37(tile); break;
            default: emit = false; break;
        }
        const uint64_t eb = __ballot(emit);
        const uint32_t pos = q.n_prog + (uint32_t)__popcll(eb & lt);
        if (emit && pos < cap) {
            uint16_t* d = prog + pos * 5;
            d[0] = (uint16_t)e0; d[1] = (uint16_t)e1; d[2] = (uint16_t)e2; d[3] = (uint16_t)e3; d[4] = (uint16_t)e4;
        }
        q.n_prog += (uint32_t)__popcll(eb);
        const uint64_t db = __ballot(ty == RMJ_EV_DAHAI || ty == RMJ_EV_KAKAN);
        if (db) q.last_da = __shfl((int)actor, 63 - __clzll((long long)db), 64);
        const uint64_t kb = __ballot(ty == RMJ_EV_START_KYOKU);
        if (kb && q.sk_idx < 0) q.sk_idx = base + (__ffsll((long long)kb) - 1);  // parse_start_kyoku_info takes the first (:476-489)
        if (seat >= 0) {
            const uint64_t stop = __ballot(ty == RMJ_EV_DAHAI || ty == RMJ_EV_CHI || ty == RMJ_EV_PON || ty == RMJ_EV_DAIMINKAN);
            const uint64_t own = __ballot(ty == RMJ_EV_TSUMO && (int)actor == seat);
            const int js = stop ? 63 - __clzll((long long)stop) : -1, jt = own ? 63 - __clzll((long long)own) : -1;
            if (jt > js) q.drawn = __shfl((int)tile, jt, 64);
            else if (js >= 0) q.drawn = -1;
        }
    }
}

// One seat's sparse tokens (:331-378), numeric features (:447-471) and candidates (:697-813).  drawn_raw: the tile of the
// seat's drawn-tile token as a 136-id (< 0: none); round-start numbers and the last discarder come from the caller's scan.
struct SeqSeatOut {
    uint16_t* sparse; uint8_t* n_sparse; float* numeric; uint16_t* cand; uint8_t* n_cand;
};
__device__ __forceinline__ uint32_t seq_used_common(const GState& S) {
    uint32_t used_common = S.n_dora;
    for (int q = 0; q < 4; q++) {
        used_common += S.p[q].n_discards;
        for (int m = 0; m < S.p[q].n_melds; m++)
            used_common += (S.p[q].meld_type[m] == RMJ_MELD_CHI || S.p[q].meld_type[m] == RMJ_MELD_PON) ? 3u : 4u;
    }
    return used_common;
}
__device__ __forceinline__ void seq_emit_seat(const GState& S, const Env& E, uint32_t g, int p, int lane, int game_style, int drawn_raw,
                                              int32_t st_honba, int32_t st_kyotaku, const int32_t* st_score, int last_da,
                                              uint32_t used_common, const SeqSeatOut& O) {
    const PState& P = S.p[p];
    const bool has_drawn = drawn_raw >= 0;
    // The reference reads the drawn tile back from the tsumo event's MJAI name (:421), i.e. as the id mjai_to_tid gives
    // that name (parser.rs:336-385: copy 0 of the type; a plain five is copy 1, a red five its own id), and compares
    // THAT id with the candidate's tile (:816-822): "tsumogiri" marks the candidate holding the name's canonical copy.
    const uint32_t dt = (uint32_t)(has_drawn ? drawn_raw : 0), dty = dt >> 2;
    const uint32_t drawn = !has_drawn ? 0xFFFFu : (seq_red(dt) ? dt : dty * 4u + ((dty == 4u || dty == 13u || dty == 22u) ? 1u : 0u));
    const uint32_t hl = P.hand_len, nd = S.n_dora < 5 ? S.n_dora : 5;
    const uint32_t used = used_common + hl;  // only the own hand is visible (state/mod.rs:192-199)
    const uint32_t remaining = 136u > 14u + used ? 136u - 14u - used : 0u;
    const uint32_t n_tok = 5u + nd + hl + (has_drawn ? 1u : 0u);
    uint32_t tok = 441u;  // :20
    if (lane == 0) tok = game_style > 0 ? 1u : 0u;
    else if (lane == 1) tok = 2u + (uint32_t)p;
    else if (lane == 2) tok = 6u + (S.round_wind < 2 ? S.round_wind : 2u);
    else if (lane == 3) tok = 9u + (S.oya < 3 ? S.oya : 3u);
    else if (lane == 4) tok = 13u + (remaining < 69u ? remaining : 69u);
    else if ((uint32_t)lane < 5u + nd) tok = 83u + (uint32_t)(lane - 5) * 37u + seq_kan37(S.dora[lane - 5]);
    else if ((uint32_t)lane < 5u + nd + hl) tok = 268u + P.hand[lane - 5 - nd];
    else if ((uint32_t)lane == 5u + nd + hl && has_drawn) tok = 404u + seq_kan37(drawn);
    if (lane < RMJ_SEQ_SPARSE) O.sparse[lane] = (uint16_t)tok;
    if (lane == 0) *O.n_sparse = (uint8_t)(n_tok < RMJ_SEQ_SPARSE ? n_tok : RMJ_SEQ_SPARSE);
    if (lane < 12) {
        float v;
        if (lane == 0) v = (float)S.honba;
        else if (lane == 1) v = (float)S.riichi_sticks;
        else if (lane < 6) v = (float)S.p[(p + lane - 2) & 3].score;
        else if (lane == 6) v = (float)st_honba;
        else if (lane == 7) v = (float)st_kyotaku;
        else v = (float)st_score[(p + lane - 8) & 3];
        O.numeric[lane] = v;
    }
    const bool acts = ((S.active_mask >> p) & 1u) && !S.is_done;
    const int nl = acts ? (int)S.nlegal[p] : 0;
    uint32_t c0 = 279, c1 = 2, c2 = 2, c3 = 3;  // padding tuple (:36)
    bool emit = false;
    if (lane < nl) {
        const uint64_t a = E.legal[((size_t)g * 4 + p) * RMJ_MAX_LEGAL + lane];
        const uint32_t ty = a_type(a), tile = a_tile(a), n = a_n(a);
        const uint32_t rel = last_da >= 0 ? seq_rel((uint32_t)p, (uint32_t)last_da) : 0u;
        switch (ty) {
            case RMJ_DISCARD: emit = tile != RMJ_TILE_NONE; c0 = seq_kan37(tile); c1 = (has_drawn && drawn == tile) ? 1u : 0u; break;
            case RMJ_ANKAN: emit = n > 0; c0 = 37u + (a_c(a, 0) >> 2); break;
            case RMJ_KAKAN: emit = tile != RMJ_TILE_NONE || n > 0; c0 = 71u + seq_kan37(tile != RMJ_TILE_NONE ? tile : a_c(a, 0)); break;
            case RMJ_TSUMO: emit = true; c0 = 108; break;
            case RMJ_KYUSHU: emit = true; c0 = 109; break;
            case RMJ_PASS: emit = true; c0 = 110; break;
            case RMJ_CHI: emit = tile != RMJ_TILE_NONE && n >= 2 && last_da >= 0; c0 = 111u + seq_chi(a_c(a, 0), a_c(a, 1), tile); c3 = rel; break;
            case RMJ_PON: emit = tile != RMJ_TILE_NONE && n >= 2 && last_da >= 0; c0 = 201u + seq_pon(a_c(a, 0), a_c(a, 1), tile); c3 = rel; break;
            case RMJ_DAIMINKAN: emit = tile != RMJ_TILE_NONE && last_da >= 0; c0 = 241u + seq_kan37(tile); c3 = rel; break;
            case RMJ_RON: emit = last_da >= 0; c0 = 278; c3 = rel; break;
            default: break;  // Riichi (:739-745), Kita (:811): no candidate
        }
    }
    const uint64_t cb = __ballot(emit);
    const int n_cand = __popcll(cb);
    if (emit) {
        uint16_t* d = O.cand + __popcll(cb & lanemask_lt(lane)) * 4;
        d[0] = (uint16_t)c0; d[1] = (uint16_t)c1; d[2] = (uint16_t)c2; d[3] = (uint16_t)c3;
    }
    if (lane >= n_cand) {
        uint16_t* d = O.cand + lane * 4;
        d[0] = 279; d[1] = 2; d[2] = 2; d[3] = 3;
    }
    if (lane == 0) *O.n_cand = (uint8_t)n_cand;
}

__global__ __launch_bounds__(64) void k_encode_seq(Env E, int game_style, RmjSeqBuffers O) {
    __shared__ GState st;
    const int lane = threadIdx.x & 63;
    const uint32_t g = blockIdx.x;
    if (lane < (int)(sizeof(GState) / 16)) reinterpret_cast<uint4*>(&st)[lane] = reinterpret_cast<const uint4*>(E.core + g)[lane];
    wave_sync();
    const GState& S = st;
    const uint32_t mask = E.ring_mask, R = mask + 1u;
    const RmjEvent* ring = E.events + (size_t)g * R;
    // stream positions as int64 from the current game's first record (GState::ev_base; congruent to the u32 positions mod 2^32)
    const int64_t eb = (int64_t)S.ev_base, ec = eb + (int64_t)(uint32_t)(S.ev_count - S.ev_base), lo = ec - eb > (int64_t)R ? ec - (int64_t)R : eb;
    // ---- the round's events: [start, ec).  start = the last START_KYOKU still in the ring (-1: overwritten)
    int64_t start = -1;
    for (int64_t hi = ec; hi > lo && start < 0; hi -= 64) {
        const int64_t idx = hi - 1 - lane;
        const bool is = idx >= lo && ring[(uint32_t)idx & mask].type == RMJ_EV_START_KYOKU;
        const uint64_t b = __ballot(is);
        if (b) start = hi - 1 - (__ffsll((long long)b) - 1);
    }
    // ---- progression (sequence_features.rs:213-314), the actor of the last dahai / kakan (:825-835), round-start numbers
    uint16_t* prog = O.progression + (size_t)g * RMJ_SEQ_PROG * 5;
    int32_t st_honba = S.honba, st_kyotaku = (int32_t)S.riichi_sticks;
    int32_t st_score[4] = {S.p[0].score, S.p[1].score, S.p[2].score, S.p[3].score};  // :490 fallback
    SeqScan q;
    if (start >= 0) {
        const RmjEvent& sk = ring[(uint32_t)start & mask];
        st_honba = sk.consumed[1];
        st_kyotaku = (int32_t)sk.consumed[2] | ((int32_t)sk.consumed[3] << 8);
        for (int i = 0; i < 4; i++) st_score[i] = sk.deltas[i];
        seq_scan_range(ring, mask, start, ec, lane, prog, RMJ_SEQ_PROG, -1, q);
    }
    uint32_t n_prog = q.n_prog;
    const int last_da = q.last_da;
    if (n_prog > RMJ_SEQ_PROG) n_prog = RMJ_SEQ_PROG;
    for (uint32_t i = n_prog + lane; i < RMJ_SEQ_PROG; i += 64) {  // padding tuple (:28)
        uint16_t* d = prog + i * 5;
        d[0] = 4; d[1] = 276; d[2] = 2; d[3] = 2; d[4] = 4;
    }
    if (lane == 0) O.n_progression[g] = start >= 0 ? (uint16_t)n_prog : (uint16_t)0xFFFF;
    // ---- per seat: sparse tokens (:331-378), numeric (:447-471), candidates (:697-813)
    const uint32_t used_common = seq_used_common(S);
    for (int p = 0; p < 4; p++) {
        // the seat's last own tsumo with nothing but reach / kan / dora events after it (:410-435) = the record's drawn tile
        // while the seat is the one to act on it
        const bool has_drawn = S.drawn_tile != 0xFF && S.current_player == p;
        const SeqSeatOut so{O.sparse + ((size_t)g * 4 + p) * RMJ_SEQ_SPARSE, O.n_sparse + (size_t)g * 4 + p,
                            O.numeric + ((size_t)g * 4 + p) * 12, O.candidates + ((size_t)g * 4 + p) * RMJ_SEQ_CAND * 4,
                            O.n_candidates + (size_t)g * 4 + p};
        seq_emit_seat(S, E, g, p, lane, game_style, has_drawn ? (int)S.drawn_tile : -1, st_honba, st_kyotaku, st_score, last_da,
                      used_common, so);
    }
}

// The same features over the events of ONE OBSERVATION, as the reference's live environment feeds them: Observation.events
// is the seat's log since its previous observation (state/mod.rs:211-218), so the progression holds only that delta, the
// drawn-tile token exists only if the delta still contains the seat's tsumo (not after its own reach), the round-start
// numbers are those of a start_kyoku inside the delta and fall back to the current ones otherwise (:490), and the last
// discarder is searched in the delta.  The delta of seat p is [obs_from[p], obs_upto[p]) of the record (advanced by
// every publication of observations for an acting seat); seats that are not to act get empty outputs.
__global__ __launch_bounds__(64) void k_encode_seq_delta(Env E, int game_style, RmjSeqDeltaBuffers O) {
    __shared__ GState st;
    const int lane = threadIdx.x & 63;
    const uint32_t g = blockIdx.x;
    if (lane < (int)(sizeof(GState) / 16)) reinterpret_cast<uint4*>(&st)[lane] = reinterpret_cast<const uint4*>(E.core + g)[lane];
    wave_sync();
    const GState& S = st;
    const uint32_t mask = E.ring_mask, R = mask + 1u;
    const RmjEvent* ring = E.events + (size_t)g * R;
    // stream positions as int64 from the current game's first record (GState::ev_base; congruent to the u32 positions mod 2^32)
    const int64_t eb = (int64_t)S.ev_base, ec = eb + (int64_t)(uint32_t)(S.ev_count - S.ev_base), lo = ec - eb > (int64_t)R ? ec - (int64_t)R : eb;
    const uint32_t used_common = seq_used_common(S);
    for (int p = 0; p < 4; p++) {
        uint16_t* prog = O.progression + ((size_t)g * 4 + p) * RMJ_SEQ_DELTA_PROG * 5;
        const bool acts = ((S.active_mask >> p) & 1u) && !S.is_done;
        const int64_t from = eb + (int64_t)(uint32_t)(S.obs_from[p] - S.ev_base), to = eb + (int64_t)(uint32_t)(S.obs_upto[p] - S.ev_base);
        const bool lost = acts && from < lo;   // the ring no longer holds the whole delta
        SeqScan q;
        int32_t st_honba = S.honba, st_kyotaku = (int32_t)S.riichi_sticks;
        int32_t st_score[4] = {S.p[0].score, S.p[1].score, S.p[2].score, S.p[3].score};
        if (acts && !lost) {
            seq_scan_range(ring, mask, from, to, lane, prog, RMJ_SEQ_DELTA_PROG, p, q);
            if (q.sk_idx >= 0) {
                const RmjEvent& sk = ring[(uint32_t)q.sk_idx & mask];
                st_honba = sk.consumed[1];
                st_kyotaku = (int32_t)sk.consumed[2] | ((int32_t)sk.consumed[3] << 8);
                for (int i = 0; i < 4; i++) st_score[i] = sk.deltas[i];
            }
        }
        uint32_t n_prog = q.n_prog > RMJ_SEQ_DELTA_PROG ? RMJ_SEQ_DELTA_PROG : q.n_prog;
        for (uint32_t i = n_prog + lane; i < RMJ_SEQ_DELTA_PROG; i += 64) {
            uint16_t* d = prog + i * 5;
            d[0] = 4; d[1] = 276; d[2] = 2; d[3] = 2; d[4] = 4;
        }
        if (lane == 0) O.n_progression[(size_t)g * 4 + p] = lost ? (uint16_t)0xFFFF : (uint16_t)n_prog;
        const SeqSeatOut so{O.sparse + ((size_t)g * 4 + p) * RMJ_SEQ_SPARSE, O.n_sparse + (size_t)g * 4 + p,
                            O.numeric + ((size_t)g * 4 + p) * 12, O.candidates + ((size_t)g * 4 + p) * RMJ_SEQ_CAND * 4,
                            O.n_candidates + (size_t)g * 4 + p};
        seq_emit_seat(S, E, g, p, lane, game_style, q.drawn, st_honba, st_kyotaku, st_score, q.last_da, used_common, so);
    }
}

}  // namespace rmj
