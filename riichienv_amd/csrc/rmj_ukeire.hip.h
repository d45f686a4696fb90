// The pair-dense ukeire walk (round 5): calculate_effective_tiles(_3p)_with_discard and calculate_best_ukeire(_3p), shanten.rs:265-405 /
// :488-626, for ONE wave-uniform hand - the same numbers as the walk in rmj_shanten.hip.h (kept as the general 3P path and for A/B),
// rearranged so that the wave's lanes are (discard, draw) PAIRS instead of one discard per iteration with lane = draw:
//
//   * draws outside the discard's suit: two discards per pass (lanes 0..31 / 32..63; at most 27 such draws per discard), the pair's
//     replacement number is ONE entry of a merge of two vectors that exist before the pass (journal_r04.md section 6): the draw's side
//     (the suit with the tile drawn, alone or merged with its partner suit) sits in the draw's type lane, the discard's side (the
//     suit without the tile, merged with its partner suit or with the other half) in a lane of the discard - both fetched by bpermute;
//   * draws inside the discard's suit: seven discards per pass (7 x 9 lanes), one table lookup each;
//   * every merge the hand needs besides the four of its base (draw side x 34, discard side x 2 x #discards <= 28) runs in ONE
//     lane-parallel call (62 lanes);
//   * cost vectors with 6-bit fields (ShantenTables::v6): the entry of a merge is two adds and a minimum over ten fields once one side
//     is stored reversed; the perfect hash is two dependent reads keyed by the packed counts (ShantenTables::r2) instead of nine;
//   * seven pairs / kokushi numbers of a pair's hand from the hand's own counts by O(1) deltas;
//   * a pair's verdict lands as a bit of a ballot; the bits of one discard are counted (effective tiles) or weighted by the tiles left
//     (ukeire) in the discard's type lane, the maximum over discards is taken once at the end.
#pragma once
#include "rmj_shanten.hip.h"   // (which includes this file at its end: sh_ukeire_both hands its fast case over)

namespace rmj {

struct S6 {
    uint32_t lo, hi;   // five 6-bit fields each: costs (k = 0..4) without / with the pair
};
#define S6_FILL 0x0F3CF3CFu   /* 15 in every field */

__device__ __forceinline__ S6 s6_vec(uint32_t word, int q, const ShantenTables& T) {
    const bool suit = q < 3;
    const uint32_t e = T.r2[(suit ? SH_R2_HI9 : SH_R2_HI7) + (word & 0x7FFFu)];
    const uint32_t s = e >> 20;
    const uint32_t l = T.r2[(suit ? SH_R2_LO9 + (s << 12) : SH_R2_LO7 + (s << 6)) + (word >> 15)];
    uint32_t idx = (e & 0xFFFFFu) + l;
    const uint32_t n = suit ? (uint32_t)SH_SUIT_ENTRIES : (uint32_t)SH_HONOR_ENTRIES;
    idx = idx < n ? idx : n - 1u;
    const uint64_t v = T.v6[(suit ? 0u : (uint32_t)SH_SUIT_ENTRIES) + idx];
    return S6{(uint32_t)v, (uint32_t)(v >> 32)};
}
__device__ __forceinline__ uint32_t s6_rev5(uint32_t x) {   // field j <- field 4 - j
    return ((x & 0x3Fu) << 24) | ((x >> 24) & 0x3Fu) | ((x & 0xFC0u) << 12) | ((x >> 12) & 0xFC0u) | (x & 0x3F000u);
}
// b as the second operand of an entry: reversed and with the halves swapped (a's cost without the pair meets b's cost with it)
__device__ __forceinline__ S6 s6_rev(const S6& b) { return S6{s6_rev5(b.hi), s6_rev5(b.lo)}; }
// ... for the entry (pair, m): field k1 = b[1 - p1][m - k1], 15 where m - k1 < 0
__device__ __forceinline__ S6 s6_at(const S6& r, int m) {
    const uint32_t sh = 6u * (uint32_t)(4 - m);
    const uint32_t fill = S6_FILL & ~((1u << (6u * (uint32_t)(m + 1))) - 1u);
    return S6{(r.lo >> sh) | fill, (r.hi >> sh) | fill};
}
__device__ __forceinline__ uint32_t s6_min5(uint32_t s, uint32_t best) {
    best = min(min(__builtin_amdgcn_ubfe(s, 0u, 6u), __builtin_amdgcn_ubfe(s, 6u, 6u)), best);
    best = min(min(__builtin_amdgcn_ubfe(s, 12u, 6u), __builtin_amdgcn_ubfe(s, 18u, 6u)), best);
    return min(s >> 24, best);
}
// entry (pair, m) of merge(a, b) with y = s6_at(s6_rev(b), m); capped at 15 like sh_merge_entry
__device__ __forceinline__ int s6_entry(const S6& a, const S6& y) { return (int)s6_min5(a.hi + y.hi, s6_min5(a.lo + y.lo, 15u)); }

// the full (min,+) merge of two vectors
__device__ __forceinline__ S6 s6_merge(const S6& a, const S6& b) {
    const uint32_t rl = s6_rev5(b.lo), rh = s6_rev5(b.hi);
    S6 o{0u, 0u};
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const uint32_t sh = 6u * (uint32_t)(4 - k);
        const uint32_t big = S6_FILL & ~((1u << (6u * (uint32_t)(k + 1))) - 1u);   // fields 0..k take part; 15 beyond (a sum there is >= 15, the cap, and < 64)
        const uint32_t yl = (rl >> sh) | big, yh = (rh >> sh) | big;              // b[.][k - k1] at field k1
        const uint32_t e0 = s6_min5(a.lo + yl, 15u);
        const uint32_t e1 = s6_min5(a.hi + yl, s6_min5(a.lo + yh, 15u));
        o.lo |= e0 << (6u * k);
        o.hi |= e1 << (6u * k);
    }
    return o;
}
__device__ __forceinline__ uint32_t s6_row_or16(uint32_t v) { return sh_row_or16(v); }
// Up to four merges at once, one per 16-lane row (row-uniform inputs): lane i < 10 of a row computes entry i, the row ORs them
// together; valid in lane 15 of every row.
__device__ __forceinline__ S6 s6_merge_rows(const S6& a, const S6& b, int lane) {
    const int i = lane & 15;
    const int p = i >= 5 ? 1 : 0, k = i < 10 ? i - 5 * p : 0;
    const uint32_t sh = 6u * (uint32_t)(4 - k), big = S6_FILL & ~((1u << (6u * (uint32_t)(k + 1))) - 1u);
    const uint32_t yl = (s6_rev5(b.lo) >> sh) | big, yh = (s6_rev5(b.hi) >> sh) | big;
    uint32_t e = s6_min5(a.lo + (p ? yh : yl), 15u);
    if (p) e = s6_min5(a.hi + yl, e);
    e = i < 10 ? e << (6u * (uint32_t)k) : 0u;
    return S6{s6_row_or16(p ? 0u : e), s6_row_or16(p ? e : 0u)};
}
__device__ __forceinline__ S6 s6_rl(const S6& v, int src) {
    return S6{(uint32_t)__builtin_amdgcn_readlane((int)v.lo, src), (uint32_t)__builtin_amdgcn_readlane((int)v.hi, src)};
}
__device__ __forceinline__ S6 s6_pull(const S6& v, int src_lane) {   // per-lane source
    return S6{(uint32_t)__builtin_amdgcn_ds_bpermute(src_lane << 2, (int)v.lo), (uint32_t)__builtin_amdgcn_ds_bpermute(src_lane << 2, (int)v.hi)};
}
__device__ __forceinline__ S6 s6_sel(bool c, const S6& x, const S6& y) { return S6{c ? x.lo : y.lo, c ? x.hi : y.hi}; }
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false));
    const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)v, 15), b = (uint32_t)__builtin_amdgcn_readlane((int)v, 31);
    const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)v, 47), d = (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
    return max(max(a, b), max(c, d));
}
// shanten.rs:228-241 / :454-468 from the replacement number and the seven pairs / kokushi statistics of the hand: kinds held twice,
// kinds held (shanten.rs:198-211), terminal kinds held, terminal kinds held twice (:213-226)
__device__ __forceinline__ int uk_finish(int rep, bool len4, int pairs, int kinds, int kk, int kp) {
    const int s = rep - 1;
    const int chi = 6 - pairs + (kinds < 7 ? 7 - kinds : 0);
    const int kok = 13 - kk - (kp > 0 ? 1 : 0);
    const int s1 = chi < s ? chi : s;
    const int s2 = (s1 > 0 && kok < s1) ? kok : s1;
    return (s > 0 && len4) ? s2 : s;
}

// Preconditions (the caller, sh_ukeire_both, checks them): at most 14 tiles; 3P: h holds no 2m..8m and leaves three honor slots
// free, so that every hand of the walk relocates both manzu terminals (see there).  lane = tile type for my_vis and the outputs.
__device__ inline void sh_ukeire_dense(const ShantenTables& T, const PH& h, uint32_t my_vis, bool sm, int lane, bool want_eff, bool want_uke,
                                       uint32_t& eff, uint32_t& uke, int cur_in, int* nsh_out, int* cur_out) {
    const int t = lane;
    const bool t_in = t < 34;
    auto suit_of = [&](int x) -> int { return (sm && (x == 0 || x == 8)) ? 3 : t_suit(x); };   // the suit whose vector a type changes
    auto type_ok = [&](int x) -> bool { return !sm || x == 0 || x >= 8; };                          // SANMA_VALID_TILE_TYPES (shanten.rs:244-247)
    auto is_term = [&](int x) -> bool { const int r = x - 9 * t_suit(x); return x >= 27 || r == 0 || r == 8; };
    auto word = [&](const PH& y, int qq) -> uint32_t {                                             // suit qq of hand y as the tables see it
        if (!sm) return ph_get(y, qq);
        return qq == 0 ? 0u : (qq == 3 ? sh_relocate_3p(y).d : ph_get(y, qq));
    };
    const int total = ph_total(h);
    const int q = t_in ? suit_of(t) : 0;
    // the hand's seven pairs / kokushi statistics
    int pairs_h, kinds_h, kk_h, kp_h;
    {
        const uint32_t a = sm ? (h.a & (7u | (7u << 24))) : h.a;
        const uint32_t T9 = 1u | (1u << 24);
        const uint32_t ha = a | (a >> 1) | (a >> 2), hb = h.b | (h.b >> 1) | (h.b >> 2), hc = h.c | (h.c >> 1) | (h.c >> 2), hd = h.d | (h.d >> 1) | (h.d >> 2);
        const uint32_t pa = (a >> 1) | (a >> 2), pb = (h.b >> 1) | (h.b >> 2), pc = (h.c >> 1) | (h.c >> 2), pd = (h.d >> 1) | (h.d >> 2);
        kinds_h = __popc(ha & O9_1) + __popc(hb & O9_1) + __popc(hc & O9_1) + __popc(hd & O7_1);
        pairs_h = __popc(pa & O9_1) + __popc(pb & O9_1) + __popc(pc & O9_1) + __popc(pd & O7_1);
        const uint32_t fa = h.a | (h.a >> 1) | (h.a >> 2), qa = (h.a >> 1) | (h.a >> 2);   // kokushi counts 1m / 9m of the unmasked word
        kk_h = __popc(fa & T9) + __popc(hb & T9) + __popc(hc & T9) + __popc(hd & O7_1);
        kp_h = __popc(qa & T9) + __popc(pb & T9) + __popc(pc & T9) + __popc(pd & O7_1);
    }
    // this lane's type: held count, what discarding one of it leaves, what drawing one adds
    const int ct = t_in ? ph_cnt(h, t) : 0;
    const bool term = t_in && is_term(t);
    const int d_pairs = pairs_h - (ct == 2), d_kinds = kinds_h - (ct == 1), d_kk = kk_h - (term && ct == 1), d_kp = kp_h - (term && ct == 2);
    const uint32_t dinfo = (uint32_t)d_pairs | ((uint32_t)d_kinds << 4) | ((uint32_t)d_kk << 8) | ((uint32_t)d_kp << 12);
    auto fin = [&](int rep, bool len4, uint32_t di, int c_before, bool tm) -> int {   // di: the statistics before the draw; the draw's type held c_before times
        return uk_finish(rep, len4, (int)(di & 15u) + (c_before == 1), (int)((di >> 4) & 15u) + (c_before == 0), (int)((di >> 8) & 15u) + (tm && c_before == 0),
                         (int)((di >> 12) & 15u) + (tm && c_before == 1));
    };
    const uint32_t hinfo = (uint32_t)pairs_h | ((uint32_t)kinds_h << 4) | ((uint32_t)kk_h << 8) | ((uint32_t)kp_h << 12);

    // the hand's suit vectors and their pair merges
    const S6 mine = s6_vec(word(h, lane & 3), lane & 3, T);
    const S6 v0 = s6_rl(mine, 0), v1 = s6_rl(mine, 1), v2 = s6_rl(mine, 2), v3 = s6_rl(mine, 3);
    S6 ab, cd;
    {
        const bool r0 = (lane >> 4) == 0;
        const S6 M = s6_merge_rows(s6_sel(r0, v0, v2), s6_sel(r0, v1, v3), lane);
        ab = s6_rl(M, 15);
        cd = s6_rl(M, 31);
    }
    const int m_h = total / 3 > 4 ? 4 : total / 3;
    const int cur = cur_in != -99 ? cur_in : uk_finish(s6_entry(ab, s6_at(s6_rev(cd), m_h)), total / 3 >= 4, pairs_h, kinds_h, kk_h, kp_h);
    if (cur_out) *cur_out = cur;
    // O_q: the three other suits of h merged (row q computes O_q); every lane keeps the one of its type's suit, reversed
    S6 Orev;
    {
        const int rw = lane >> 4;
        const S6 pa = rw == 0 ? v1 : (rw == 1 ? v0 : (rw == 2 ? v3 : v2));
        const S6 Om = s6_merge_rows(pa, rw < 2 ? cd : ab, lane);
        const S6 O0 = s6_rl(Om, 15), O1 = s6_rl(Om, 31), O2 = s6_rl(Om, 47), O3 = s6_rl(Om, 63);
        Orev = s6_rev(q == 0 ? O0 : (q == 1 ? O1 : (q == 2 ? O2 : O3)));
    }
    const bool drawable = t_in && type_ok(t) && ct < 4;
    S6 vt{0u, 0u};                       // this type's suit with one tile of the type drawn
    if (drawable) {
        PH hp = h;
        ph_add(hp, t);
        vt = s6_vec(word(hp, q), q, T);
    }
    eff = 0;
    uke = 0;
    if (want_eff && total % 3 == 1) {
        const int m13 = (total + 1) / 3 > 4 ? 4 : (total + 1) / 3;
        const bool f = drawable && fin(s6_entry(vt, s6_at(Orev, m13)), (total + 1) / 3 >= 4, hinfo, ct, term) < cur;
        eff = (uint32_t)__popcll(__ballot(f));
    }
    if (want_eff && total % 3 == 0) eff = 0xFFFFFFFFu;
    const bool eff_loop = want_eff && total % 3 == 2;
    if (!eff_loop && !want_uke) return;
    const int m_loop = total / 3 > 4 ? 4 : total / 3;            // hands of the walk hold total tiles again (h - d + t)
    const int m_sub = (total - 1) / 3 > 4 ? 4 : (total - 1) / 3;
    const bool len4 = total / 3 >= 4;
    // shanten after discarding one tile of this lane's type
    int nsh_l = 127;
    S6 nv{0u, 0u};
    if (t_in && ct > 0) {
        PH sub = h;
        ph_sub(sub, t);
        nv = s6_vec(word(sub, q), q, T);
        nsh_l = uk_finish(s6_entry(nv, s6_at(Orev, m_sub)), (total - 1) / 3 >= 4, d_pairs, d_kinds, d_kk, d_kp);
    }
    if (nsh_out) *nsh_out = nsh_l;
    const uint64_t cand = __ballot(nsh_l <= cur);
    if (!cand) return;
    const int ncand = __popcll(cand);
    const bool is_c = (cand >> lane) & 1ull;
    const int ci = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(cand >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)cand, 0u));   // this discard's number
    // lanes 34 + c / 48 + c receive discard c: its type and its suit without the tile
    int rd;
    S6 rv;
    {
        const int dst_h = (is_c ? 34 + ci : 63) << 2, dst_w = (is_c ? 48 + ci : 63) << 2;
        const int d1 = __builtin_amdgcn_ds_permute(dst_h, t), l1 = __builtin_amdgcn_ds_permute(dst_h, (int)nv.lo), h1 = __builtin_amdgcn_ds_permute(dst_h, (int)nv.hi);
        const int d2 = __builtin_amdgcn_ds_permute(dst_w, t), l2 = __builtin_amdgcn_ds_permute(dst_w, (int)nv.lo), h2 = __builtin_amdgcn_ds_permute(dst_w, (int)nv.hi);
        const bool w = lane >= 48;
        rd = w ? d2 : d1;
        rv = S6{(uint32_t)(w ? l2 : l1), (uint32_t)(w ? h2 : h1)};
    }
    // one lane-parallel merge: lanes 0..33 the draw side for the other half (suit with the tile drawn (+) its partner suit), lanes
    // 34.. the discard side for the other half (suit without the tile (+) its partner suit), lanes 48.. for the partner suit (suit
    // without the tile (+) the other half); the discard sides are kept reversed for the entry (pair, m_loop)
    S6 X;
    {
        const bool draw_side = lane < 34;
        const int qx = draw_side ? q : suit_of((rd < 0 || rd > 33) ? 0 : rd);
        const S6 part = qx == 0 ? v1 : (qx == 1 ? v0 : (qx == 2 ? v3 : v2));
        const S6 othr = qx < 2 ? cd : ab;
        const S6 M = s6_merge(draw_side ? vt : rv, lane >= 48 ? othr : part);
        X = draw_side ? M : s6_at(s6_rev(M), m_loop);
    }
    const S6 OrM = s6_at(Orev, m_loop);
    // tiles of the type left to draw: 4 - visible - held, both saturating (shanten.rs:380-389); a pair whose draw is its discard never counts
    int w_t = 4 - (int)my_vis;
    w_t = w_t < 0 ? 0 : w_t;
    w_t -= ct;
    w_t = (t_in && w_t > 0) ? w_t : 0;
    const uint32_t tinfo = (uint32_t)ct | ((uint32_t)term << 3) | ((uint32_t)w_t << 4) | ((uint32_t)drawable << 7);
    uint32_t effc = 0, ukec = 0;   // lane = discard: draws that lower the shanten / their tiles left
    // ---- draws outside the discard's suit, two discards per pass
    {
        uint64_t cs = cand;
        const int hf = lane >> 5, j = lane & 31;
        for (int it = 0; cs; it++) {
            const int da = __ffsll((long long)cs) - 1;
            cs &= cs - 1ull;
            const bool two = cs != 0ull;
            const int db = two ? __ffsll((long long)cs) - 1 : da;
            cs &= cs - 1ull;   // (0 stays 0)
            const int qa = suit_of(da), qb = suit_of(db);
            const int qd = hf ? qb : qa;
            const uint32_t di = hf ? (uint32_t)__builtin_amdgcn_readlane((int)dinfo, db) : (uint32_t)__builtin_amdgcn_readlane((int)dinfo, da);
            const int nsh = hf ? __builtin_amdgcn_readlane(nsh_l, db) : __builtin_amdgcn_readlane(nsh_l, da);
            int tt = j + ((qd < 3 && j >= 9 * qd) ? 9 : 0);
            bool ok = (hf == 0 || two) && tt < 34 && (qd < 3 || j < 27);
            tt = ok ? tt : 0;
            const int qt = suit_of(tt);
            const uint32_t ti = (uint32_t)__builtin_amdgcn_ds_bpermute(tt << 2, (int)tinfo);
            ok = ok && qt != qd && ((ti >> 7) & 1u);
            const bool partner = (qt ^ 1) == qd;
            const S6 a = s6_sel(partner, s6_pull(vt, tt), s6_pull(X, tt));
            const S6 b = s6_pull(X, (partner ? 48 : 34) + 2 * it + hf);
            const int c_t = (int)(ti & 7u);
            const bool f = ok && fin(s6_entry(a, b), len4, di, c_t, (ti >> 3) & 1u) < nsh;
            const uint64_t fb = __ballot(f);
            const bool mine = is_c && (ci >> 1) == it;
            const bool up = ci & 1;
            if (eff_loop) effc += mine ? (uint32_t)__popc(up ? (uint32_t)(fb >> 32) : (uint32_t)fb) : 0u;
            if (want_uke) {
                const uint32_t w = f ? (ti >> 4) & 7u : 0u;
                const uint64_t b1 = __ballot(w & 1u), b2 = __ballot(w & 2u), b4 = __ballot(w & 4u);
                const uint32_t u = (uint32_t)__popc(up ? (uint32_t)(b1 >> 32) : (uint32_t)b1) + 2u * (uint32_t)__popc(up ? (uint32_t)(b2 >> 32) : (uint32_t)b2) +
                                   4u * (uint32_t)__popc(up ? (uint32_t)(b4 >> 32) : (uint32_t)b4);
                ukec += mine ? u : 0u;
            }
        }
    }
    // ---- draws inside the discard's suit, seven discards per pass (nine lanes each)
    {
        const int slot = (lane * 57) >> 9, r = lane - 9 * slot;   // lane / 9, lane % 9
        for (int ps = 0; 7 * ps < ncand; ps++) {
            const int c2 = 7 * ps + slot;
            bool ok = slot < 7 && c2 < ncand;
            int d = __builtin_amdgcn_ds_bpermute((34 + (ok ? c2 : 0)) << 2, rd);
            d = (d < 0 || d > 33) ? 0 : d;
            const int qd = suit_of(d);
            int tt;
            if (sm && qd == 3) tt = r < 7 ? 27 + r : (r == 7 ? 0 : 8);
            else { tt = 9 * qd + r; ok = ok && (qd < 3 || r < 7); }
            tt = ok ? tt : d;
            PH x = h;
            ph_sub(x, d);
            const int c_t = ph_cnt(x, tt);
            ok = ok && type_ok(tt) && c_t < 4;
            ph_add(x, tt);
            S6 a{0u, 0u};
            if (ok) a = s6_vec(word(x, qd), qd, T);
            const S6 b = s6_pull(OrM, d);
            const uint32_t di = (uint32_t)__builtin_amdgcn_ds_bpermute(d << 2, (int)dinfo);
            const int nsh = __builtin_amdgcn_ds_bpermute(d << 2, nsh_l);
            const bool f = ok && fin(s6_entry(a, b), len4, di, c_t, is_term(tt)) < nsh;
            const uint64_t fb = __ballot(f);
            const bool mine = is_c && (ci >= 7 ? 1 : 0) == ps;
            const uint32_t sft = 9u * (uint32_t)(ci >= 7 ? ci - 7 : ci);
            if (eff_loop) effc += mine ? (uint32_t)__popc((uint32_t)(fb >> sft) & 0x1FFu) : 0u;
            if (want_uke) {
                const uint32_t ti = (uint32_t)__builtin_amdgcn_ds_bpermute(tt << 2, (int)tinfo);
                const uint32_t w = f ? (ti >> 4) & 7u : 0u;
                const uint64_t b1 = __ballot(w & 1u), b2 = __ballot(w & 2u), b4 = __ballot(w & 4u);
                const uint32_t u = (uint32_t)__popc((uint32_t)(b1 >> sft) & 0x1FFu) + 2u * (uint32_t)__popc((uint32_t)(b2 >> sft) & 0x1FFu) +
                                   4u * (uint32_t)__popc((uint32_t)(b4 >> sft) & 0x1FFu);
                ukec += mine ? u : 0u;
            }
        }
    }
    if (eff_loop) eff = wave_max_u32(is_c ? effc : 0u);
    if (want_uke) uke = wave_max_u32(is_c ? ukec : 0u);
}

}  // namespace rmj
