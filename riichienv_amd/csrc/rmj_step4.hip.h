// Four games per wavefront: the fast path of the step kernel re-expressed with ONE 16-LANE DPP ROW PER GAME.
//
// Why (profiles/r02_pmc_k_step.json): with one game per wave the step is bound by instruction ISSUE on the
// vector and the scalar port together (~620 VALU + ~640 SALU wave-instructions per game-step, both ports ~85 % busy at
// eight waves per SIMD), and the vector instructions use at most 16 of their 64 lanes for real work - everything else is
// wave-uniform bookkeeping.  Here the four rows of a wave hold four different games: lane r of a row is hand slot r
// (0..13), seat r (0..3), meld slot r ... exactly the 16-lane sub-layouts the one-game code already used; the per-game
// "scalar" bookkeeping runs once per wave on the vector unit for four games at a time, and the scalar unit is left with
// loop control and addresses.  Cross-lane traffic stays inside a row: DPP row shifts / sums, ballots sliced per row,
// ds_bpermute for "read lane k of my row".
//
// Coverage: every common transition - discard (incl. the riichi discard), claim generation (Pon / Daiminkan / Chi lists,
// ordered like the reference), Pass / Pon / Chi responses, the next draw, the drawer's legal list, wait-cache refills by
// the isolated-tile bound and the table shanten, 3P Kita.  Anything that needs a yaku evaluation, a wait probe, a kan, a
// riichi declaration, a round end or a restart makes ITS ROW bail; the wave finishes the other rows, stores them, and then
// runs the bailed games one by one through the complete state machine (ol_step_full) from their untouched HBM records.
// Results are identical to k_step by construction of the parity suite (every GPU test runs on this kernel).
//
// Included once per variant inside namespace RMJ_NS, after rmj_kernels.hip.h.
namespace RMJ_NS {

#define R4_LIST 16 /* staged list entries per (game, seat); a longer list makes the row bail */
/* The drawer's list (WaitAct: the only list of its game) may run on into the next seat's slots where that seat is never the last of a
   row - 3P, whose lists are the long ones (fourteen discards + a Kita per North + Riichi: 0.2 % of the game-steps beyond 16 entries) */
#define R4_ACT_CAP (KSANMA ? 2 * R4_LIST : R4_LIST)
// what makes a row stop in pass 1 of a step (Quad4Shared::rmode; the caller runs r4_round_end / r4_yaku_answers, then pass 2)
#define R4_RE_DRAW 1u      /* exhaustive draw */
#define R4_RE_RESTART 2u   /* a finished game restarts (auto-reset) */
#define R4_RE_PUB 3u       /* pass 2 of step4_body: the round has been dealt, the row's observation is still to be published */
#define R4_RE_YAKU_CLAIMS 4u /* paused in r4_resolve_discard: seats without a riichi wait on the discard - their yaku decide (pass 2 resumes there) */
#define R4_RE_YAKU_TSUMO 5u  /* paused in r4_gen_act_legal: an open hand is complete - its yaku decide whether Tsumo is offered */
#define R4_RE_WIN_TSUMO 6u   /* the drawer's Tsumo: settled between the passes (r4_round_end) */
#define R4_RE_WIN_RON 7u     /* Ron answers on the discard / the kan: settled between the passes; the winners' seats in Quad4Shared::yk */
#define R4_RET_ROUND 0x100u /* step4_body's result: some row of the wave ends a round - the caller runs r4_round_end and pass 2 */
// Games per wave of the non-ticket kernels (flags bits 20..21: 0 = four, 1 = one, 2 = two): a batch that gives the chip fewer than two
// waves per SIMD at four games per wave is latency bound - the same rows spread over more waves hide each other's LDS / HBM round trips
// (the instruction stream of a wave does not shrink with its rows, so this only pays while the vector units idle).
#define STEP_F_ROWS_SHIFT 20
__device__ __forceinline__ uint32_t r4_rows(uint32_t flags) {
    const uint32_t c = (flags >> STEP_F_ROWS_SHIFT) & 3u;
    return c == 0u ? 4u : c;
}
#ifndef RMJ_INLINE_STEP
#define RMJ_INLINE_STEP 1     /* the step of the fused rollouts inlined into the rollout loop (see step4_call_inl); 0: out of line, as in rounds 2-4 */
#endif
#ifndef RMJ_ROW_SETTLE
#define RMJ_ROW_SETTLE 1      /* Tsumo / Ron settlements of the rich tier stay in tier 0 too (r4_round_end); 0: they bail to the full path */
#endif
#ifndef RMJ_ROW_ROUND_END
#define RMJ_ROW_ROUND_END 1   /* exhaustive draws, next rounds and restarts stay in tier 0 (r4_round_end); 0: they enter the full path at the exit */
#endif
#ifndef RMJ_FULL_PRIO
#define RMJ_FULL_PRIO 0
#endif
#ifndef RMJ_RON_SKIP
#define RMJ_RON_SKIP 1   /* Ron eligibility: skipped when no seat of any game of the wave waits on its game's discard (0: A/B) */
#endif
#ifndef RMJ_CHI_SKIP
#define RMJ_CHI_SKIP 1   /* chi lists: a pattern no row of the wave can form is skipped as a whole (0: A/B) */
#endif
#ifndef RMJ_GROUP_FILTER_ALL
#define RMJ_GROUP_FILTER_ALL 0   /* 1: the group-residue tests also in the kernels of one step per launch (A/B; see step4_body) */
#endif
#ifndef RMJ_GROUP_FILTER13
#define RMJ_GROUP_FILTER13 1   /* wait-cache refill: the group-residue test in front of the table shanten (r4_group_residues; 0: A/B) */
#endif
#ifndef RMJ_GROUP_FILTER14
#define RMJ_GROUP_FILTER14 1   /* riichi bound of the drawer's list: the same on 14 tiles (0: A/B) */
#endif
#ifndef RMJ_HEAVY_TENPAI
#define RMJ_HEAVY_TENPAI 1   /* heavy-first order of the per-step kernel: games with a seat that waits without a riichi count as heavy (0: A/B) */
#endif
#ifdef RMJ_CUTS   /* instruction accounting build (scripts/valu_sections4.py, scripts/bail_census.py) */
/* (an asm s_endpgm, not __builtin_amdgcn_endpgm: the builtin is noreturn, and a noreturn call inside divergent control flow lets the
   compiler drop the EXEC restore behind the region - rows that were masked off there stayed off for the rest of the step in the greedy
   instantiation: ADVICE r3 "mark 43", journal r04 section 16) */
#define R4M(id) do { if (rmj::g_cut == (id)) asm volatile("s_endpgm" ::: "memory"); } while (0)   /* the wave ends at mark g_cut */
#define R4BAIL(q, id) do { (q).bail = true; if ((q).r == 0) atomicAdd(&rmj::g_bail_reason[id], 1u); } while (0)   /* bail census */
#else
#define R4M(id) do {} while (0)
#ifdef RMJ_CENSUS   /* bail census on the shipped instruction stream (no accounting marks): scripts/bail_census.py; the reason travels in a register and is counted where the full path is entered */
#define R4BAIL(q, id) do { (q).bail = true; (q).why = (id); } while (0)
#elif defined(RMJ_TL4)
#define R4BAIL(q, id) do { (q).bail = true; if ((q).r == 0) rmj::g_tl4[(size_t)blockIdx.x * RMJ_TL4_ROW + 10 + (q).row] = (unsigned long long)(id) + 1ull; } while (0)
#else
#define R4BAIL(q, id) do { (q).bail = true; } while (0)
#endif
#endif
#ifdef RMJ_TL4   /* timeline build (scripts/timeline4.py): core cycles between the outer marks of step4_body, summed over the waves */
#define R4T(k) do { const uint64_t t__ = __builtin_readcyclecounter(); \
        if ((threadIdx.x & 63) == 0) rmj::g_tl4[(size_t)blockIdx.x * RMJ_TL4_ROW + (k)] = (unsigned long long)(t__ - tl_prev); tl_prev = t__; } while (0)
#else
#define R4T(k) do {} while (0)
#endif
struct Quad4Tier0 {
    alignas(16) uint32_t ev[4][RMJ_EV_STAGE][8];  // staged MJAI records per game
    uint32_t evidx[4][RMJ_EV_STAGE];
    uint64_t lst[4][4][R4_LIST];                   // staged legal lists per (game, seat)
    uint32_t mk[4][4][4];                          // 82-bit action-id masks per (game, seat), built with LDS atomics
};
struct Quad4Enc {          // byte staging of Observation.encode() inside the fused step + encode rollout (after the step: the union is free)
    alignas(16) uint8_t raw[(ENC_CH * (KSANMA ? ENC_W3 : ENC_W4) + 4 + 15) / 16 * 16];
    uint32_t hist[ENC_HIST_WORDS];
};
#ifndef R4_RS_WORDS
#define R4_RS_WORDS 100   /* round-end scratch (r4_round_end): 32 words of packed bucket counters + 136 16-bit sort words.  Measured (profiles/r04_lds_sweep.txt): up to 6 336 B of LDS per wave the fused rollout runs at 1.81-1.82 G env.step/s, at 6 512 B at 1.70 G */
#endif
struct Quad4Shared {
    GState st[4];
    union {
        WaveScratch x;   // scratch of the full path (bailed games, after the tier-0 rows have been stored)
        Quad4Tier0 t;
        Quad4Enc e;
    } u;
    uint32_t rs[R4_RS_WORDS];   // outside the union: a round ends while the other rows' lists and events are still staged
    uint32_t rfl[4];            // rows whose round ends in this call: their publication flags (pass 2 of step4_body) ...
    uint32_t rmode[4];          // ... and what ends it (R4_RE_*; 0: nothing)
    uint32_t yk[4];             // answers of r4_yaku_answers: R4_RE_YAKU_CLAIMS: seats that may Ron | seats with a shape but no yaku << 4; _TSUMO: bit 0 = offer
};
// The wave's working set (every block of the four-games-per-wave kernels is one wave).  One namespace-scope variable instead of a
// static in each entry point: the out-of-line pieces of a step (r4_round_end) address it directly, as LDS.
static_assert(sizeof(Quad4Shared) <= 6336, "LDS per wave: one allocation step more costs the fused rollouts 7 % (R4_RS_WORDS)");
__shared__ Quad4Shared g_q4;
#ifdef RMJ_QTL
__shared__ uint32_t g_qtl_cnt[2];   // ticket timeline build: calls of the step function in the current ticket, and the live rows summed over them
#endif

// ballot of the lane's own row (round 6: one byte permute of the two halves, row_ballot16 in rmj_hand.hip.h - the select + bit-field
// extract of rounds 2-5 was compiled to a 64-bit shift by a vector register, which gfx950 gets wrong now and then when that register is
// the wave's last allocated one)
__device__ __forceinline__ uint32_t rballot(bool p, int rb) { return row_ballot16(p, rb); }
__device__ __forceinline__ int rbc(int v, int src_lane) { return __builtin_amdgcn_ds_bpermute(src_lane << 2, v); }
__device__ __forceinline__ uint64_t rbc64(uint64_t v, int src_lane) {
    return (uint64_t)(uint32_t)rbc((int)(uint32_t)v, src_lane) | ((uint64_t)(uint32_t)rbc((int)(uint32_t)(v >> 32), src_lane) << 32);
}
__device__ __forceinline__ uint32_t row_or16(uint32_t v) {  // OR over a 16-lane row, result in lane 15 of the row
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    return v;
}

// Per-row context: everything is a per-lane value that is uniform within the row.
struct R4 {
    GState* G;
    Quad4Tier0* T;
    CEnv* E;
    int lane, r, rb, row;
    uint32_t g;
    bool live;        // the row still runs in tier 0
    bool bail;        // the row goes to the full path
    int cont;         // with bail: the full path is entered AT this point of the step, on the record tier 0 leaves in LDS (STEP_F_CONT_*), instead of starting over
    uint32_t rend;    // R4_RE_*: the row's game ends a round in this call (r4_round_end), or pauses for a yaku check (R4_RE_YAKU_*)
    bool pause_ok;    // pass 1: a row that needs the evaluator pauses (its caller runs r4_yaku_answers, pass 2 resumes it); else it bails
    uint32_t yk;      // pass 2: the answers of r4_yaku_answers for this row ...
    uint32_t yk_mode; // ... and what they answer (R4_RE_YAKU_*; 0: none)
#ifdef RMJ_CENSUS
    int why;          // the R4BAIL site that made the row bail
#endif
    int evn;          // staged events
    uint32_t dirty;
    bool gf;          // the group-residue tests in front of the table shanten (r4_group_residues): a compile-time constant of the instantiation, see step4_body
};

__device__ __forceinline__ void r4_emit(R4& q, uint32_t w0, uint32_t w1, uint32_t w6) {
    if (q.E->skip_log) return;
    const uint32_t evc = q.G->ev_count;
    if (q.evn >= RMJ_EV_STAGE) { R4BAIL(q, 1); return; }
    if (q.r == 0) {
        uint4* b = reinterpret_cast<uint4*>(q.T->ev[q.row][q.evn]);
        b[0] = make_uint4(w0, w1, 0u, 0u);
        b[1] = make_uint4(0u, 0u, w6, (uint32_t)KNP << 24);
        q.T->evidx[q.row][q.evn] = evc & q.E->ring_mask;
        q.G->ev_count = evc + 1;
    }
    q.evn += 1;
    wave_sync();
}
__device__ __forceinline__ void r4_emit_simple(R4& q, uint32_t type, uint32_t actor, uint32_t tile, uint32_t fl = 0) {
    r4_emit(q, type | (actor << 8) | (tile << 24), 0u, fl);
}

// concealed histogram of a seat's hand (optionally without slot `skip`): lane r contributes hand[r]; uniform in the row
__device__ __forceinline__ PH r4_hist(const R4& q, const PState* P, int skip) {
    const int n = P->hand_len;
    uint32_t a = 0, b = 0, c = 0, d = 0;
    if (q.r < n && q.r != skip) {
        const int t = P->hand[q.r] >> 2, s = t_suit(t);
        const uint32_t one = 1u << (3 * (t - 9 * s));
        a = s == 0 ? one : 0u; b = s == 1 ? one : 0u; c = s == 2 ? one : 0u; d = s == 3 ? one : 0u;
    }
    a = row_sum16(a); b = row_sum16(b); c = row_sum16(c); d = row_sum16(d);
    const int l15 = q.rb + 15;
    PH h;
    h.a = (uint32_t)rbc((int)a, l15); h.b = (uint32_t)rbc((int)b, l15); h.c = (uint32_t)rbc((int)c, l15); h.d = (uint32_t)rbc((int)d, l15);
    return h;
}
// isolated tiles of a histogram, bitwise (same number as isolated_tiles(): held exactly once, nothing within two ranks)
__device__ __forceinline__ int r4_isolated(const PH& h) {
    int iso = 0;
#pragma unroll
    for (int s = 0; s < 3; s++) {
        const uint32_t x = ph_get(h, s);
        const uint32_t nz = (x | (x >> 1) | (x >> 2)) & O9_1;
        const uint32_t one = x & ~(x >> 1) & ~(x >> 2) & O9_1;                                  // fields equal to 1
        const uint32_t nb = ((nz << 3) | (nz << 6) | (nz >> 3) | (nz >> 6)) & O9_1;             // a held rank within two
        iso += __popc(one & ~nb);
    }
    const uint32_t x = h.d;
    iso += __popc(x & ~(x >> 1) & ~(x >> 2) & O7_1);
    return iso;
}
// table shanten (4P tables) of one histogram per row: lane i (< 9) computes its term of the perfect hash suit by suit
__device__ __forceinline__ int r4_shanten(const R4& q, const PH& h, int len_div3) {
    const ShantenTables T = sh_tables_of(*q.E);
    const int i = q.r, l15 = q.rb + 15;
    uint64_t vec[4];
    uint32_t val[4];
#pragma unroll
    for (int s = 0; s < 4; s++) {   // the four rank-table loads are in flight together, then the four vector loads
        val[s] = 0;
        if (i < (s < 3 ? 9 : 7)) {
            const uint32_t x = ph_get(h, s);
            uint32_t c = (x >> (3 * i)) & 7u;
            uint32_t sm = (uint32_t)field_sum(x & ((1u << (3 * i)) - 1u));
            if (c > 4u) c = 4u;
            if (sm > 14u) sm = 14u;
            if (sm + c > 14u) c = 14u - sm;
            val[s] = (s < 3 ? T.rank9 : T.rank7)[(i * 15 + sm) * 5 + c];
        }
    }
#pragma unroll
    for (int s = 0; s < 4; s++) val[s] = (uint32_t)rbc((int)row_sum16(val[s]), l15);
#pragma unroll
    for (int s = 0; s < 4; s++) vec[s] = s < 3 ? T.suit[val[s]] : T.honor[val[s]];
    const int m = len_div3 > 4 ? 4 : len_div3;
    // lanes 0..9: entry idx of merge(a, b) and the entry of merge(c, d) that pairs with it in the final entry (pair = 1, m)
    const int p = i >= 5 ? 1 : 0, k = i - 5 * p;
    uint32_t t = 99u;
    if (i < 10 && k <= m) {
        const uint32_t e1 = sh_merge_entry((uint32_t)vec[0] & 0xFFFFFu, (uint32_t)(vec[0] >> 20) & 0xFFFFFu, (uint32_t)vec[1] & 0xFFFFFu,
                                           (uint32_t)(vec[1] >> 20) & 0xFFFFFu, p, k);
        const uint32_t e2 = sh_merge_entry((uint32_t)vec[2] & 0xFFFFFu, (uint32_t)(vec[2] >> 20) & 0xFFFFFu, (uint32_t)vec[3] & 0xFFFFFu,
                                           (uint32_t)(vec[3] >> 20) & 0xFFFFFu, 1 - p, m - k);
        t = e1 + e2;
    }
    t = row_min16(t);
    uint32_t rep = (uint32_t)rbc((int)t, l15);
    if (rep > 15u) rep = 15u;
    int sres = (int)rep - 1;
    if (sres <= 0 || len_div3 < 4) return sres;
    const int ch = sh_chiitoi(h, false);
    sres = ch < sres ? ch : sres;
    if (sres > 0) {
        const int kk = sh_kokushi(h);
        sres = kk < sres ? kk : sres;
    }
    return sres;
}
// "Spread" key of a tile type: ranks of one suit stay adjacent, suits are 16 apart, honors 3 apart - two tiles can belong to
// one set, pair or taatsu iff their keys differ by at most 2 (honors: iff equal).
__device__ __forceinline__ int r4_key(int t34) {
    return t34 < 27 ? t34 + 7 * ((t34 >= 9) + (t34 >= 18)) : 48 + 3 * (t34 - 27);
}
// Shape numbers of a SORTED run of `n` tiles of P's hand (13-tile hands are kept sorted; a 14th, drawn tile sits behind
// them) without building a histogram: lane r looks at its two sorted neighbours (DPP row shifts).
struct R4Shape {
    int iso;      // tiles held exactly once with nothing within two ranks (isolated_tiles())
    int yaochu;   // kinds of terminals and honors
    int kinds;    // distinct tile types
    int pairs;    // types held at least twice
    bool start;   // per lane: this tile opens a connected group (nothing within two ranks below it)
    uint32_t S;   // the row's ballot of `start`
};
__device__ __forceinline__ R4Shape r4_shape_sorted(const R4& q, const PState* P, int n) {
    const int r = q.r, rb = q.rb;
    const bool in = r < n;
    const int t = in ? (int)(P->hand[r] >> 2) : 99;
    const int k = in ? r4_key(t) : 1000;
    const int kp = __builtin_amdgcn_update_dpp(-1000, k, 0x111 /* row_shr:1 */, 0xf, 0xf, false);   // lane r - 1 (row start: -1000)
    const int kn = __builtin_amdgcn_update_dpp(1000, k, 0x101 /* row_shl:1 */, 0xf, 0xf, false);     // lane r + 1 (row end: 1000)
    const bool first = in && k != kp, last_of_kind = in && k != kn;
    R4Shape o;   // (round 5: the "one extra tile" form of rounds 3-4 had a single caller without an extra tile)
    o.start = in && k - kp > 2;
    o.S = rballot(o.start, rb);
    o.iso = __popc(rballot(o.start && kn - k > 2, rb));
    const bool term = t >= 27 || t == 0 || t == 8 || t == 9 || t == 17 || t == 18 || t == 26;
    o.yaochu = __popc(rballot(first && term, rb));   // (needed whenever the kokushi bound 12 - kinds can undercut the isolated-tile bound: no cheap gate)
    o.kinds = __popc(rballot(first, rb));
    o.pairs = __popc(rballot(first && !last_of_kind, rb));
    return o;
}
// Sizes mod 3 of the connected groups of a sorted run of n tiles (a group = maximal chain of tiles whose neighbours are within two
// ranks: every set, pair and taatsu lies inside one group).  Returns (#groups of size 1 mod 3) | (#groups of size 2 mod 3) << 4.
// A standard-form tenpai of 3m + 1 tiles is m sets and a single ({1}), or m - 1 sets, a pair and a pair / taatsu: the two two-tile
// blocks in one group ({1}) or in two ({2, 2}); every other residue pattern is at least one tile away from tenpai.  Adding one tile
// anywhere (3m + 2 tiles one discard away from tenpai): {2}, {1, 1} or {2, 2, 1}.  (round 6: random 13-tile hands that still need the
// table shanten after the isolated-tile bound: 5.8 % -> 0.6 %)
__device__ __forceinline__ uint32_t r4_group_residues(bool start, uint32_t S, int n, int r, int rb) {
    const uint32_t rest = S >> (r + 1);
    const int size = (rest ? __ffs((int)rest) : n - r);   // distance to the next group's first tile, or to the end of the run
    const int res = size - 3 * ((size * 11) >> 5);        // size % 3 for size <= 14
    return (uint32_t)__popc(rballot(start && res == 1, rb)) | ((uint32_t)__popc(rballot(start && res == 2, rb)) << 4);
}
// One suit word of a hand judged on its own (agari.rs:183-245, boolean): `tot` = its tile count mod 3; 0: sets only, 2: a pair and
// sets (three pair candidates by pair_residue), 1: never.  s = 3: honors.
__device__ __forceinline__ bool r4_word_ok(uint32_t w, int s, int tot) {
    bool ok = false;
    if (tot != 1) {
        if (s == 3) {
            ok = tot == 0 ? honors_ok0(w) : honors_ok2(w);
        } else {
            int j = pair_residue(w);
            uint32_t y = w;
            bool go = true;
            if (tot == 2) {
                go = ((w >> (3 * j)) & 7u) >= 2u;
                y = w - (2u << (3 * j));
            }
            ok = go && mentsu_ok(y);
            if (tot == 2 && !ok) {
#pragma unroll 1
                for (int k = 0; k < 2 && !ok; k++) {
                    j += 3;
                    if (((w >> (3 * j)) & 7u) >= 2u) ok = mentsu_ok(w - (2u << (3 * j)));
                }
            }
        }
    }
    return ok;
}
// HandEvaluator::get_waits (hand_evaluator.rs:196-213) of the row's 3n+1-tile histogram h (row-uniform), in row form - the wait
// probe that made a row leave tier 0 whenever a hand came within reach of tenpai (half of all exits once the policy plays to
// win).  Adding tile t changes ONE suit word, so the standard form factorises like in wave_waits: lanes 0..3 judge the four
// unmodified words; a suit can take the winning tile only if the three others are consistent with exactly one pair among the
// four (at most two suits qualify: tile counts mod 3 are {1,0,0,0} or {2,2,0,0}); per qualifying suit one pass, lane = rank.
// Chiitoi (six pairs and a single: closed form) and kokushi (lane = terminal kind) only behind their kind / pair gates.
// Quirk Q7: a type already held four times (concealed) is never a wait.
__device__ __noinline__ uint64_t r4_waits_probe(uint32_t ha, uint32_t hb, uint32_t hc, uint32_t hd) {
    const int lane = threadIdx.x & 63, r = lane & 15, rb = lane & 48;
    const PH h = {ha, hb, hc, hd};
    const int t0 = field_sum(h.a) % 3, t1 = field_sum(h.b) % 3, t2 = field_sum(h.c) % 3, t3 = field_sum(h.d) % 3;
    const uint32_t res2 = (uint32_t)(t0 == 2) | ((uint32_t)(t1 == 2) << 1) | ((uint32_t)(t2 == 2) << 2) | ((uint32_t)(t3 == 2) << 3);
    const uint32_t res0 = (uint32_t)(t0 == 0) | ((uint32_t)(t1 == 0) << 1) | ((uint32_t)(t2 == 0) << 2) | ((uint32_t)(t3 == 0) << 3);
    const int my_s = r & 3;
    const int my_t = my_s == 0 ? t0 : (my_s == 1 ? t1 : (my_s == 2 ? t2 : t3));
    const uint32_t okb = rballot(r < 4 && r4_word_ok(ph_get(h, my_s), my_s, my_t), rb) & 0xFu;   // unmodified words that are consistent
    // suit s can take the tile: its count becomes 0 or 2 mod 3, the others are consistent, one pair among the four
    uint32_t V = 0u;
#pragma unroll
    for (int s4 = 0; s4 < 4; s4++) {
        const uint32_t others = 0xFu & ~(1u << s4);
        const bool was2 = (res2 >> s4) & 1u, was0 = (res0 >> s4) & 1u;           // new residue: was 2 -> 0, was 1 -> 2, was 0 -> 1 (never)
        const int pairs_after = __popc(res2 & others) + ((!was2 && !was0) ? 1 : 0);
        if (!was0 && (okb & others) == others && pairs_after == 1) V |= 1u << s4;
    }
    uint64_t W = 0ull;
    while (__ballot(V != 0u)) {
        if (V) {
            const int s = __ffs((int)V) - 1;
            V &= V - 1u;
            const uint32_t w = ph_get(h, s);
            const bool in = r < (s == 3 ? 7 : 9);
            const uint32_t sh3 = in ? 3u * (uint32_t)r : 0u;
            const bool live = in && ((w >> sh3) & 7u) < 4u;
            const int tot = ((res2 >> s) & 1u) ? 0 : 2;
            const bool ok = live && r4_word_ok(w + (1u << sh3), s, tot);
            W |= (uint64_t)(rballot(ok, rb) & 0x1FFu) << (9 * s);
        }
    }
    // chiitoi / kokushi (13 concealed tiles only: six pairs resp. twelve terminal kinds)
    const int pairs = __popc(h.a & O9_2 & ~(h.a << 1)) + __popc(h.b & O9_2 & ~(h.b << 1)) + __popc(h.c & O9_2 & ~(h.c << 1)) +
                      __popc(h.d & O7_2 & ~(h.d << 1));  // fields equal to 2 or 6
    const uint32_t pres_d = (h.d | (h.d >> 1) | (h.d >> 2)) & O7_1;
    const int kinds = __popc(pres_d) + ((h.a & 7u) != 0u) + ((h.a >> 24) != 0u) + ((h.b & 7u) != 0u) + ((h.b >> 24) != 0u) +
                      ((h.c & 7u) != 0u) + ((h.c >> 24) != 0u);
    if (__ballot(pairs >= 6 || kinds >= 12)) {
        if (pairs >= 6) {
            // is_chiitoi(h + t): every field of h is 0 or 2 except field t, which is 1, and there are six pairs
            const uint32_t odd_a = h.a & O9_1, odd_b = h.b & O9_1, odd_c = h.c & O9_1, odd_d = h.d & O7_1;
            const uint32_t four = (h.a & O9_4) | (h.b & O9_4) | (h.c & O9_4) | (h.d & O7_4);
            const int n_odd = __popc(odd_a) + __popc(odd_b) + __popc(odd_c) + __popc(odd_d);
            const int n_two = __popc(h.a & O9_2) + __popc(h.b & O9_2) + __popc(h.c & O9_2) + __popc(h.d & O7_2);   // fields with bit 1: 2 or 3
            if (four == 0u && n_odd == 1 && n_two == 6) {
                const int t = odd_a ? (__ffs((int)odd_a) - 1) / 3 : (odd_b ? 9 + (__ffs((int)odd_b) - 1) / 3 : (odd_c ? 18 + (__ffs((int)odd_c) - 1) / 3 : 27 + (__ffs((int)odd_d) - 1) / 3));
                if (ph_cnt(h, t) == 1) W |= 1ull << t;
            }
        }
        if (kinds >= 12) {
            // lane = terminal kind (13 of them): 1m 9m 1p 9p 1s 9s E S W N P F C
            const int t = r < 6 ? (r >> 1) * 9 + (r & 1) * 8 : 27 + (r - 6);
            bool kw = false;
            if (r < 13 && ph_cnt(h, t) < 4) {
                PH x = h;
                ph_add(x, t);
                kw = is_kokushi(x);
            }
            const uint32_t kb = rballot(kw, rb);
            // scatter the 13 lane bits to their types
            uint64_t kmask = 0ull;
#pragma unroll
            for (int i = 0; i < 13; i++) {
                const int ti = i < 6 ? (i >> 1) * 9 + (i & 1) * 8 : 27 + (i - 6);
                kmask |= (uint64_t)((kb >> i) & 1u) << ti;
            }
            W |= kmask;
        }
    }
    return W;
}

// fill_waits13 of tier 0 for the sorted 13-tile hand P->hand[0 .. n): the isolated-tile bound, else the table shanten (the
// only user of a histogram); a hand with a possible wait (shanten <= 0) needs the probe -> the row bails.  Writes the cache
// like fill_waits13 (waits13 = 0 for every shanten >= 1).
// RICH: the tier-0 build with the wait probe and the riichi transitions (see step4_body); without it a possible wait makes the row bail
template <bool RICH>
__device__ __forceinline__ void r4_fill_waits13(R4& q, PState* P, int n) {
    uint64_t W = 0ull;
    const R4Shape sp = r4_shape_sorted(q, P, n);
    const int iso = sp.iso, yaochu = sp.yaochu;
    int lb = iso >= 6 ? 4 : (iso == 5 ? 3 : (iso == 4 ? 2 : 0));
    const int len3 = n / 3;
    if (len3 == 4) {
        const int koku = 12 - yaochu;
        lb = koku < lb ? koku : lb;
        if (lb > 2) {
            const int chi = 6 - sp.pairs + (sp.kinds < 7 ? 7 - sp.kinds : 0);   // sh_chiitoi
            lb = chi < lb ? chi : lb;
        }
    }
    int sh = lb;
    if (lb < 2) {
        // A tenpai 13-tile hand has at most ONE isolated tile (a tanki wait; chiitoi tenpai: its single), and a kokushi
        // tenpai holds twelve kinds of terminals and honors: two isolated tiles and fewer than twelve such kinds mean
        // shanten >= 1 - no waits, which is all the cache must know; sh13 = 1 is then a lower bound (its users only ever
        // skip work on ">= 2").  Only the remaining ~5 % of the hands take the table shanten.
        // Of the rest, a standard-form tenpai needs its groups' sizes mod 3 to be {1} or {2, 2} (r4_group_residues); seven pairs
        // and kokushi keep their own gates.
        bool far = iso >= 2 && yaochu < 12;
        if (RMJ_GROUP_FILTER13 && q.gf && !far && !(len3 == 4 && (sp.pairs >= 6 || yaochu >= 12))) {
            const uint32_t gr = r4_group_residues(sp.start, sp.S, n, q.r, q.rb);
            far = gr != 0x01u && gr != 0x20u;
        }
        if (far) {
            sh = 1;
        } else {
            PH h13 = {0, 0, 0, 0};
            {   // histogram of the n sorted tiles
                uint32_t a = 0, b = 0, c = 0, d = 0;
                if (q.r < n) {
                    const int t = P->hand[q.r] >> 2, su = t_suit(t);
                    const uint32_t one = 1u << (3 * (t - 9 * su));
                    a = su == 0 ? one : 0u; b = su == 1 ? one : 0u; c = su == 2 ? one : 0u; d = su == 3 ? one : 0u;
                }
                a = row_sum16(a); b = row_sum16(b); c = row_sum16(c); d = row_sum16(d);
                const int l15 = q.rb + 15;
                h13.a = (uint32_t)rbc((int)a, l15); h13.b = (uint32_t)rbc((int)b, l15); h13.c = (uint32_t)rbc((int)c, l15); h13.d = (uint32_t)rbc((int)d, l15);
            }
            sh = r4_shanten(q, h13, len3);
            if (sh <= 0) {
                if (!RICH) { R4BAIL(q, 2); return; }
                W = r4_waits_probe(h13.a, h13.b, h13.c, h13.d);
                sh = 0;
            }
        }
    }
    if (q.r == 0) {
        P->waits13 = W;
        P->sh13 = (uint8_t)sh;
        P->flags |= PF_WAITS_VALID;
    }
    wave_sync();
}

// The same from a histogram, for a hand whose 13 tiles are not one sorted run (right after a Kita the previous drawn tile
// sits behind the twelve sorted ones).
template <bool RICH>
__device__ __forceinline__ void r4_fill_waits13_h(R4& q, PState* P, const PH& h13) {
    uint64_t W = 0ull;
    const uint32_t T9 = 1u | (1u << 24);
    const int yaochu = __popc((h13.a | (h13.a >> 1) | (h13.a >> 2)) & T9) + __popc((h13.b | (h13.b >> 1) | (h13.b >> 2)) & T9) +
                       __popc((h13.c | (h13.c >> 1) | (h13.c >> 2)) & T9) + __popc((h13.d | (h13.d >> 1) | (h13.d >> 2)) & O7_1);
    const int iso = r4_isolated(h13);
    int lb = iso >= 6 ? 4 : (iso == 5 ? 3 : (iso == 4 ? 2 : 0));
    const int len3 = ph_total(h13) / 3;
    if (len3 == 4) {
        const int koku = 12 - yaochu;
        lb = koku < lb ? koku : lb;
        if (lb > 2) {
            const int chi = sh_chiitoi(h13, false);
            lb = chi < lb ? chi : lb;
        }
    }
    int sh = lb;
    if (lb < 2) {
        if (iso >= 2 && yaochu < 12) {
            sh = 1;
        } else {
            sh = r4_shanten(q, h13, len3);
            if (sh <= 0) {
                if (!RICH) { R4BAIL(q, 2); return; }
                W = r4_waits_probe(h13.a, h13.b, h13.c, h13.d);
                sh = 0;
            }
        }
    }
    if (q.r == 0) {
        P->waits13 = W;
        P->sh13 = (uint8_t)sh;
        P->flags |= PF_WAITS_VALID;
    }
    wave_sync();
}

// accept_riichi (state/mod.rs:1549-1567)
__device__ __forceinline__ void r4_accept_riichi(R4& q) {
    GState* G = q.G;
    const int p = G->riichi_pending;
    if (p != 0xFF) {
        q.dirty |= 1u << p;
        if (q.r == 0) {
            G->p[p].score -= 1000;
            G->p[p].score_delta -= 1000;
            G->riichi_sticks += 1;
            G->p[p].flags |= PF_RIICHI_DECLARED | PF_IPPATSU;
            G->riichi_pending = 0xFF;
        }
        wave_sync();
        r4_emit_simple(q, RMJ_EV_REACH_ACCEPTED, (uint32_t)p, 0);
    }
}
// check_abortive_draw (state/mod.rs:1970-2019): any abortive draw makes the row bail; lane = seat * 4 + meld slot
__device__ __forceinline__ void r4_check_abortive(R4& q) {
    GState* G = q.G;
    const int p = (q.r >> 2) & 3, m = q.r & 3;
    const PState& P = G->p[p];
    const bool seat_lane = m == 0;
    const int nm = P.n_melds;
    const uint32_t turns_ok = rballot(seat_lane && P.n_discards == 1, q.rb);
    const uint32_t has_melds = rballot(seat_lane && nm != 0, q.rb);
    const uint32_t riichi = rballot(seat_lane && (P.flags & PF_RIICHI_DECLARED), q.rb);
    const uint32_t kan = rballot(m < nm && P.meld_type[m] >= RMJ_MELD_DAIMINKAN, q.rb);
    const uint32_t all_seats = 0x1111u;
    if (!KSANMA && turns_ok == all_seats && has_melds == 0u) {
        const int first = G->p[0].discards[0] >> 2;
        if (first >= 27 && first <= 30) {
            const uint32_t same = rballot(seat_lane && (P.discards[0] >> 2) == first, q.rb);
            if (same == all_seats) R4BAIL(q, 4);
        }
    }
    if (__popc(kan) == 4) {
        const int owner = (__ffs((int)kan) - 1) >> 2;
        if (kan & ~(0xFu << (4 * owner))) R4BAIL(q, 5);
    }
    if (!KSANMA && riichi == all_seats) R4BAIL(q, 6);
}
// deal_next (state/mod.rs:1569-1593); pf = the prefetched live-wall tile W[live_end - 1]
__device__ __forceinline__ void r4_deal_next(R4& q, int pf) {
    GState* G = q.G;
    const int drawable = G->drawable_count;
    if (drawable == 0) {   // exhaustive draw
        if (RMJ_ROW_ROUND_END && q.pause_ok) q.rend = 1u /* R4_RE_DRAW */;   // the round ends in row form, behind the transitions of this call (r4_round_end)
        else { R4BAIL(q, 7); q.cont = 1; }                                    // the full path takes over right here (trigger_ryukyoku)
        return;
    }
    const int live_end = G->live_end;
    const int pid = G->current_player;
    PState* P = &G->p[pid];
    const int hl = P->hand_len;
    if (live_end > G->rinshan_count) {
        q.dirty |= 1u << pid;
        if (q.r == 0) {
            G->is_rinshan = 0;
            G->live_end = (uint8_t)(live_end - 1);
            G->drawable_count = (uint8_t)(drawable - 1);
            if (hl < 14) { P->hand[hl] = (uint8_t)pf; P->hand_len = (uint8_t)(hl + 1); }
            G->drawn_tile = (uint8_t)pf;
            G->needs_tsumo = 0;
            G->phase = RMJ_WAIT_ACT;
            G->active_mask = (uint8_t)(1u << pid);
            P->n_forbidden = 0;
        }
        wave_sync();
        r4_emit_simple(q, RMJ_EV_TSUMO, (uint32_t)pid, (uint32_t)pf);
    } else if (q.r == 0) {
        G->is_rinshan = 0;
    }
}

// stage one list entry together with its action id (Action::encode / encode_3p, action.rs:158-346) in the free top byte:
// where an entry is generated its kind is known, so the id is a constant or one shift - the publication does not decode
__device__ __forceinline__ int r4_tile_id(int t34) {   // discard id of a tile type: the type, 3P: the compact index
    return KSANMA ? (t34 == 0 ? 0 : (t34 == 8 ? 1 : t34 - 7)) : t34;
}
__device__ __forceinline__ void r4_put(R4& q, int seat, int pos, uint64_t a, int id) {
    if (pos < R4_ACT_CAP) (&q.T->lst[q.row][seat][0])[pos] = a | ((uint64_t)(uint32_t)id << 56);   // (claim lists stay below R4_LIST: r4_resolve_discard checks)
}

// wall tile `idx` of the row's game (fused rollouts read their own earlier stores past the vector L1, see step4_body)
template <bool LOOP>
__device__ __forceinline__ int r4_wall_tile(const R4& q, int idx) {
    const uint8_t* Wg = q.E->wall + (size_t)q.g * RMJ_WALL_STRIDE;
    if (LOOP) {
        const uint32_t w = __hip_atomic_load(reinterpret_cast<const uint32_t*>(Wg) + (idx >> 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return (int)((w >> (8 * (idx & 3))) & 0xFFu);
    }
    return Wg[idx];
}
// _reveal_kan_dora (state/mod.rs:2021-2046) `extra` more times after the pending ones (flush_pending_kan_dora)
template <bool LOOP>
__device__ __forceinline__ void r4_flush_kan_dora(R4& q, int extra) {
    GState* G = q.G;
    int pk = (int)G->pending_kan_dora + extra;
    while (__ballot(pk > 0)) {
        if (pk > 0) {
            pk -= 1;
            const int count = G->n_dora;
            const int widx = KSANMA ? 8 + 2 * count : 4 + 2 * count;
            const bool flip = count < 5 && (KSANMA || widx < (int)G->live_end);
            const int t = flip ? r4_wall_tile<LOOP>(q, widx) : 0;
            if (q.r == 0) {
                G->pending_kan_dora = (uint8_t)(pk >= extra ? pk - extra : 0);
                if (flip) { G->dora[count] = (uint8_t)t; G->n_dora = (uint8_t)(count + 1); }
            }
            wave_sync();
            if (flip) r4_emit_simple(q, RMJ_EV_DORA, 0u, (uint32_t)t);
        }
    }
}
// _resolve_kan from the replacement draw on (state/mod.rs:1475-1547): every seat loses ippatsu, the first turn is over, the
// replacement tile comes from the front of the dead wall, then the meld event (w0 / cons / n; 0: a Kakan announced itself
// earlier), the indicators (an Ankan flips its own at once, an open kan leaves it pending), the tsumo event
template <bool LOOP>
__device__ __forceinline__ void r4_kan_draw(R4& q, int pid, PState* P, bool ankan, uint32_t ev_w0, uint32_t ev_cons, uint32_t ev_n) {
    GState* G = q.G;
    const int r = q.r;
    if (G->drawable_count == 0) { R4BAIL(q, 29); return; }
    if (r < 4) G->p[r].flags &= ~PF_IPPATSU;
    const int rc = G->rinshan_count;
    const int t = r4_wall_tile<LOOP>(q, rc);
    if (r == 0) {
        G->is_first_turn = 0;
        G->rinshan_count = (uint8_t)(rc + 1);
        G->drawable_count -= 1;
        const int h2 = P->hand_len;
        if (h2 < 14) { P->hand[h2] = (uint8_t)t; P->hand_len = (uint8_t)(h2 + 1); }
        G->drawn_tile = (uint8_t)t;
        G->is_rinshan = 1;
    }
    wave_sync();
    if (ev_w0) r4_emit(q, ev_w0, ev_cons, (ev_n << 4) & 0xFFu);
    r4_flush_kan_dora<LOOP>(q, ankan ? 1 : 0);
    if (!ankan && r == 0) G->pending_kan_dora += 1;
    wave_sync();
    r4_emit_simple(q, RMJ_EV_TSUMO, (uint32_t)pid, (uint32_t)t);
    if (r == 0) {
        G->phase = RMJ_WAIT_ACT;
        G->active_mask = (uint8_t)(1u << pid);
    }
    wave_sync();
}
// Yaku check of a complete hand for tier 0 (round 4): seat `seat` of the row's game wins on `tile` (136-id; a Ron tile is added to the 13
// held ones, a drawn tile is held already) under the condition flags cf - the legality form of seat_calc (no ura, no kita count:
// legal_actions.rs:44-61, 254-310) through the row-form evaluator, the rows' seats side by side.  Every lane of the wave calls, rows
// with on = 0 idle.  Result (row-uniform): bit 0 = WinResult.is_win, bit 1 = the hand has a win shape, bit 2 = yakuman or han >= 1.
// SETTLE: the settlement form (state/mod.rs:700-745, 990-1030): the caller's honba, ura indicators off the wall when `use_ura`
// (_get_ura_indicators, state/mod.rs:2048-2057; 3P: pre-extracted W[9 + 2k]), the seat's Norths as Conditions.kita_count.
template <bool SETTLE>
__device__ __forceinline__ E4Out r4_seat_eval(uint32_t on, uint32_t seat, uint32_t tile, uint32_t cf, uint32_t honba, bool use_ura, const uint8_t* Wg) {
    const int lane = threadIdx.x & 63, row = lane >> 4, r = lane & 15, rb = lane & 48;
    const GState* G = &g_q4.st[row];
    const PState* P = &G->p[seat & 3u];
    const bool act = on != 0u;
    const int hl = act ? (int)P->hand_len : 0, nm = act ? (int)P->n_melds : 0;
    // lane = hand slot
    const int tid = r < hl ? (int)P->hand[r] : 0;
    uint32_t ca = 0, cb = 0, cc = 0, cd = 0;
    if (r < hl) {
        const int t = tid >> 2, su = t_suit(t);
        const uint32_t one = 1u << (3 * (t - 9 * su));
        ca = su == 0 ? one : 0u; cb = su == 1 ? one : 0u; cc = su == 2 ? one : 0u; cd = su == 3 ? one : 0u;
    }
    PH h;
    h.a = e4_rsum(ca, rb); h.b = e4_rsum(cb, rb); h.c = e4_rsum(cc, rb); h.d = e4_rsum(cd, rb);
    int aka = __popc(e4_ballot(r < hl && is_aka(tid), rb));
    // lane = meld
    uint32_t ma_ = 0, mb_ = 0, mc_ = 0, md_ = 0;
    E4Meld mp;
    {
        const bool mv = r < nm;
        const int m = r & 3;
        const uint32_t type = P->meld_type[m];
        const int nt = type >= RMJ_MELD_DAIMINKAN ? 4 : 3;
        const uint32_t t0 = P->meld_tiles[m][0], t1 = P->meld_tiles[m][1], t2 = P->meld_tiles[m][2], t3 = P->meld_tiles[m][3];
        if (mv) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int t = (int)(k == 0 ? t0 : (k == 1 ? t1 : (k == 2 ? t2 : t3))) >> 2;
                if (k < nt) {
                    const int su = t_suit(t);
                    const uint32_t one = 1u << (3 * (t - 9 * su));
                    ma_ += su == 0 ? one : 0u; mb_ += su == 1 ? one : 0u; mc_ += su == 2 ? one : 0u; md_ += su == 3 ? one : 0u;
                }
            }
        }
        mp = e4_meld_lane(mv, type, nt, t0, t1, t2, t3, (int)(t0 >> 2), type != RMJ_MELD_CHI, type != RMJ_MELD_ANKAN);
    }
    const E4Meld ma = e4_meld_reduce(mp, rb);
    aka += e4m_aka(ma);
    PH full = h;
    full.a += e4_rsum(ma_, rb); full.b += e4_rsum(mb_, rb); full.c += e4_rsum(mc_, rb); full.d += e4_rsum(md_, rb);
    const int win34 = (int)(tile >> 2) < 34 ? (int)(tile >> 2) : 33;
    if (hl + 3 * nm == 13) {
        ph_add(h, win34);
        ph_add(full, win34);
        aka += is_aka((int)tile);
    }
    // lane = indicator
    int dora, ura = 0;
    const int kita = (SETTLE && KSANMA && act) ? (int)P->n_kita : 0;
    {
        const bool is_d = act && r < 5 && r < (int)G->n_dora;
        int cnt = 0;
        if (is_d) {
            const int nt = next_dora34(((int)G->dora[r & 7] >> 2) < 34 ? ((int)G->dora[r & 7] >> 2) : 33, KSANMA);
            cnt = ph_cnt(full, nt);
            if (SETTLE && KSANMA && nt == 30) cnt += kita;   // hand_evaluator_3p.rs:110-116
        }
        dora = (int)e4_rsum((uint32_t)cnt, rb);
        if (SETTLE) {
            const int idx = KSANMA ? 9 + 2 * r : 5 + 2 * r;
            int ucnt = 0;
            if (is_d && use_ura && (KSANMA || idx < (int)G->live_end)) {
                const uint32_t ww = __hip_atomic_load(reinterpret_cast<const uint32_t*>(Wg) + (idx >> 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int wt = (int)((ww >> (8 * (idx & 3))) & 0xFFu) >> 2;
                const int nt = next_dora34(wt < 34 ? wt : 33, KSANMA);
                ucnt = ph_cnt(full, nt);
                if (KSANMA && nt == 30) ucnt += kita;
            }
            ura = (int)e4_rsum((uint32_t)ucnt, rb);
        }
    }
    E4In in;
    in.on = act;
    in.hand14 = h;
    in.ma = ma;
    in.win34 = win34;
    in.cf = cf;
    in.dora = dora & 0xFF; in.aka = aka; in.ura = ura & 0xFF; in.nuki = kita;
    in.round_wind34 = 27 + ((int)G->round_wind & 3);
    {
        int sw = (int)(seat & 3u) + KNP - (int)G->oya;
        sw = sw >= KNP ? sw - KNP : sw;
        in.seat_wind34 = 27 + (sw & 3);
    }
    in.sanma = KSANMA;
    in.honba = SETTLE ? honba : (uint32_t)G->honba;
    return e4_calc(in, r, rb);
}
__device__ __forceinline__ uint32_t r4_yaku_check(uint32_t on, uint32_t seat, uint32_t tile, uint32_t cf) {
    const E4Out o = r4_seat_eval<false>(on, seat, tile, cf, 0u, false, nullptr);
    return (o.is_win ? 1u : 0u) | (o.shape ? 2u : 0u) | ((o.yakuman || o.han >= 1) ? 4u : 0u);
}

// _resolve_discard (state/mod.rs:1317-1413) incl. claim generation (legal_actions.rs:254-508) for the row's game.
// nl[] = list length of seat r (lanes r < 4) after the call; returns through G->phase / active_mask like the reference.
// known_sh: the exact shanten of the 13 tiles the discard leaves when the policy has just computed it (99: unknown)
// RESUME (pass 2): the discard itself - bookkeeping, indicators, the dahai event, the wait-cache refills - was done in pass 1, which
// paused at the Ron check (R4_RE_YAKU_CLAIMS); the function picks up there with the evaluator's answers (q.yk)
template <bool RICH, bool LOOP, bool RESUME = false>
__device__ __forceinline__ void r4_resolve_discard(R4& q, int pid, int tile, bool tsumogiri, int pf, int& nl_mine, uint64_t& w_mine, int known_sh = 99) {
    GState* G = q.G;
    PState* P = &G->p[pid];
    const int r = q.r, rb = q.rb;
    if (!RESUME) {
    // (kan dora indicators waiting for this discard: flipping them here was measured - 0.05 % fewer exits, 2.7 % slower, the
    //  load and the loop cost registers in the hottest function)
    // RICH: the indicators of earlier open kans are flipped here, in front of the dahai event, like the full path does
    // (state/mod.rs:1357-1361); the lean tier leaves a discard after a kan to the full path
    if (!RICH && G->pending_kan_dora > 0) { R4BAIL(q, 8); return; }
    {
        const int tt = tile >> 2;
        uint32_t fl = P->flags;
        const bool stage = fl & PF_RIICHI_STAGE;
        int nd = P->n_discards;
        if (r == 0) {
            if (KSANMA) { G->pending_kan_pid = 0xFF; G->pending_kan_action = 0; }
            G->is_rinshan = 0;
            fl &= ~(uint32_t)PF_IPPATSU;
            if (nd < RMJ_MAX_DISCARDS) {
                P->discards[nd] = (uint8_t)tile;
                if (!tsumogiri) P->discard_from_hand_bits |= 1u << nd;
                if (stage) P->discard_is_riichi_bits |= 1u << nd;
                nd += 1;
                P->n_discards = (uint8_t)nd;
            }
            P->discard_type_mask |= 1ull << tt;
            G->last_discard_pid = (uint8_t)pid;
            G->last_discard_tile = (uint8_t)tile;
            G->drawn_tile = 0xFF;
            if (!tsumogiri) P->last_tedashi = (uint8_t)tile;
            if (known_sh != 99 && known_sh >= 1) {
                // a hand with a wait has shanten 0: the cache of the remaining 13 tiles is "no waits, shanten known_sh" - no refill,
                // no table lookup when the next discard arrives, and an exact number for the riichi bound after the next draw
                P->waits13 = 0ull;
                P->sh13 = (uint8_t)known_sh;
                fl |= PF_WAITS_VALID;
            } else if (!tsumogiri) {
                const int lb = P->sh13;
                if ((fl & PF_WAITS_VALID) && lb >= 3 && known_sh == 99) P->sh13 = (uint8_t)(lb - 1);
                else fl &= ~(uint32_t)PF_WAITS_VALID;
            }
            G->needs_tsumo = 1;
            if (stage) {
                fl |= PF_RIICHI_DECLARED;
                if (G->is_first_turn) fl |= PF_DOUBLE_RIICHI;
                P->riichi_decl_idx = (uint8_t)(nd - 1);
                fl &= ~(uint32_t)PF_RIICHI_STAGE;
                G->riichi_pending = (uint8_t)pid;
            }
            fl &= ~(uint32_t)PF_MISSED_DOUJUN;
            if (!is_terminal_tile136(tile)) fl &= ~(uint32_t)PF_NAGASHI;
            P->flags = (uint8_t)fl;
            G->active_mask = 0;
            G->ron_offer_mask = 0;
        }
    }
    wave_sync();
    if (RICH) {   // flush_pending_kan_dora / _reveal_kan_dora (state/mod.rs:2021-2046)
        if (__ballot(G->pending_kan_dora > 0)) r4_flush_kan_dora<LOOP>(q, 0);
        if (q.bail) return;   // (event staging full)
    }
    r4_emit_simple(q, RMJ_EV_DAHAI, (uint32_t)pid, (uint32_t)tile, tsumogiri ? 1u : 0u);
    R4M(50);
    if (q.bail) return;
    // ---- A: refill stale wait caches of the other seats that hold 13 tiles: every row walks ITS OWN stale seats (usually
    //      one - the previous discarder), so the loop runs once or twice per wave, not once per seat index
    {
        const PState& S0 = G->p[r & 3];
        uint32_t need_m = rballot(r < KNP && r != pid && (S0.hand_len + 3 * S0.n_melds == 13) && !(S0.flags & PF_WAITS_VALID), rb) & 0xFu;
        q.dirty |= need_m;
        while (__ballot(need_m != 0u)) {
            if (need_m) {
                const int i = __ffs((int)need_m) - 1;
                need_m &= need_m - 1u;
                PState* Q = &G->p[i];
                r4_fill_waits13<RICH>(q, Q, Q->hand_len);
                if (q.bail) need_m = 0u;
            }
        }
    }
    R4M(51);
    if (q.bail) return;
    }   // (!RESUME)
    const int tt = tile >> 2;
    // ---- B (lane = seat): Ron eligibility: a seat that waits on the tile and is not furiten needs the yaku check -> bail
    const PState& S4 = G->p[r & 3];
    const bool other = r < KNP && r != pid;
    const uint32_t qfl = S4.flags;
    const bool holds13 = other && (S4.hand_len + 3 * S4.n_melds == 13);
    const uint64_t W = holds13 ? S4.waits13 : 0ull;
    const uint32_t riichi_m = rballot(r < 4 && (qfl & PF_RIICHI_DECLARED), rb) & 0xFu;
    // A seat in riichi that waits on the tile and is not furiten may win without a look at its yaku (riichi is one, the shape is
    // the cached wait: calc.is_win of legal_actions.rs:254-310 is true): RICH offers that Ron here; any other seat that could
    // win needs the evaluator - full path.
    uint32_t ron_m = 0u;
    if (!RMJ_RON_SKIP || __ballot(((W >> tt) & 1ull) != 0ull)) {   // (wave-uniform) nobody in the wave's games waits on its game's tile (nearly every discard): no furiten arithmetic
        const uint64_t dtm = S4.discard_type_mask;
        const bool in_discards = (dtm >> tt) & 1ull;
        const bool in_missed = (qfl & PF_MISSED_DOUJUN) || ((qfl & PF_RIICHI_DECLARED) && (qfl & PF_MISSED_RIICHI));
        const bool furiten = (W & dtm) != 0ull || (qfl & (PF_MISSED_RIICHI | PF_MISSED_DOUJUN));
        ron_m = rballot(other && !in_discards && !in_missed && !furiten && ((W >> tt) & 1ull), rb) & 0xFu;
    }
    if (ron_m && !RICH) { R4BAIL(q, 9); return; }
    if (RICH && (ron_m & ~riichi_m)) {
        // seats that wait on the tile without a riichi: their yaku decide.  Pass 1 pauses here - the evaluator runs between the passes
        // (r4_yaku_answers), pass 2 resumes with its answers: a shape without a yaku costs the seat its turn's Ron chances
        // (state/mod.rs:1386-1389)
        if (!RESUME) {
            if (q.pause_ok) q.rend = R4_RE_YAKU_CLAIMS; else R4BAIL(q, 9);
            return;
        }
        const uint32_t ok_m = (ron_m & riichi_m) | (q.yk & ron_m & ~riichi_m), miss_m = (q.yk >> 4) & ron_m & ~riichi_m;
        if (r < 4 && ((miss_m >> r) & 1u)) G->p[r].flags |= PF_MISSED_DOUJUN;
        q.dirty |= miss_m;
        ron_m = ok_m;
        wave_sync();
    }
    w_mine = r < 4 ? W : 0ull;
    const bool can_call = G->drawable_count > 0;
    const bool kuikae = (q.E->rule_bits & RMJ_RULE_KUIKAE_FORBIDDEN) != 0;
    R4M(52);
    // list length of each seat so far (row-uniform; no indexed array: no scratch); a Ron is the first entry of its seat's list
    int nl0 = (RICH && (ron_m & 1u)) ? 1 : 0, nl1 = (RICH && (ron_m & 2u)) ? 1 : 0, nl2 = (RICH && (ron_m & 4u)) ? 1 : 0, nl3 = (RICH && (ron_m & 8u)) ? 1 : 0;
    auto nl_get = [&](int i) { return i == 0 ? nl0 : (i == 1 ? nl1 : (i == 2 ? nl2 : nl3)); };
    auto nl_set = [&](int i, int v) {
        nl0 = i == 0 ? v : nl0; nl1 = i == 1 ? v : nl1; nl2 = i == 2 ? v : nl2; nl3 = i == 3 ? v : nl3;
    };
    // ---- C: Pon / Daiminkan material, seat by seat (lane = hand slot)
#pragma unroll
    for (int i = 0; i < KNP; i++) {
        if (i == pid) continue;
        const PState* Q = &G->p[i];
        const int hl = Q->hand_len;
        const int ht = r < hl ? (int)Q->hand[r] : 0xFF;
        const uint32_t sm = can_call ? rballot(r < hl && (ht >> 2) == tt, rb) : 0u;
        const int count = __popc(sm);
        if (count < 2 || ((riichi_m >> i) & 1u)) continue;
        const int i0 = __ffs((int)sm) - 1;
        const uint32_t m1 = sm & (sm - 1u);
        const int i1 = __ffs((int)m1) - 1;
        const int i2 = count >= 3 ? __ffs((int)(m1 & (m1 - 1u))) - 1 : 0;
        const uint32_t h0 = (uint32_t)rbc(ht, rb + i0), h1 = (uint32_t)rbc(ht, rb + i1), h2 = (uint32_t)rbc(ht, rb + i2);
        int n = nl_get(i);
        if (r == 0) {
            if (hl >= 3 && (kuikae ? (hl - count) > 0 : (hl - 2) > 0)) {
                r4_put(q, i, n, mk_action(RMJ_PON, tile, 2, h0, h1), KSANMA ? 28 : 41);
                if (count >= 3) {
                    r4_put(q, i, n + 1, mk_action(RMJ_PON, tile, 2, h0, h2), KSANMA ? 28 : 41);
                    r4_put(q, i, n + 2, mk_action(RMJ_PON, tile, 2, h1, h2), KSANMA ? 28 : 41);
                }
            }
        }
        if (hl >= 3 && (kuikae ? (hl - count) > 0 : (hl - 2) > 0)) n += count >= 3 ? 3 : 1;
        if (count >= 3) {
            if (r == 0) r4_put(q, i, n, mk_action(RMJ_DAIMINKAN, tile, 3, h0, h1, h2), (KSANMA ? 29 : 42) + r4_tile_id(tt));
            n += 1;
        }
        nl_set(i, n);
    }
    R4M(53);
    // ---- D: Chi for the next seat (lane = a * 4 + b, pattern by pattern); no Chi in 3P
    if (!KSANMA && can_call && tt < 27) {
        const int i = (pid + 1) & 3;
        const PState* Q = &G->p[i];
        const int hl = Q->hand_len;
        if (!((riichi_m >> i) & 1u) && hl >= 3) {
            const int ht = r < hl ? (int)Q->hand[r] : 0xFF;
            const int hty = ht >> 2;
            const int r9 = tt % 9;
            const uint32_t m_m2 = rballot(r < hl && hty == tt - 2, rb), m_m1 = rballot(r < hl && hty == tt - 1, rb);
            const uint32_t m_p1 = rballot(r < hl && hty == tt + 1, rb), m_p2 = rballot(r < hl && hty == tt + 2, rb);
            if ((m_m2 && m_m1) || (m_m1 && m_p1) || (m_p1 && m_p2)) {
                const uint32_t m_0 = rballot(r < hl && hty == tt, rb);
                const uint32_t m_p3 = rballot(r < hl && hty == tt + 3, rb), m_m3 = rballot(r < hl && hty == tt - 3, rb);
                const int a = (r >> 2) & 3, b = r & 3;
                int n = nl_get(i);
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const bool pat_ok = k == 0 ? (r9 >= 2) : (k == 1 ? (r9 >= 1 && r9 <= 7) : (r9 <= 6));
                    const uint32_t ma = k == 0 ? m_m2 : (k == 1 ? m_m1 : m_p1);
                    const uint32_t mbb = k == 0 ? m_m1 : (k == 1 ? m_p1 : m_p2);
#if RMJ_CHI_SKIP
                    if (!__ballot(pat_ok && ma != 0u && mbb != 0u)) continue;   // (wave-uniform) no row of the wave holds this pattern's two tiles: nothing to list
#endif
                    int forb = __popc(m_0);
                    if (k == 2 && r9 <= 5) forb += __popc(m_p3);
                    if (k == 0 && r9 >= 3) forb += __popc(m_m3);
                    const bool kk_ok = kuikae ? (hl - 2 - forb) > 0 : (hl - 2) > 0;
                    const bool valid = pat_ok && a < __popc(ma) && b < __popc(mbb) && kk_ok;
                    uint32_t x = ma, y = mbb;
                    for (int s = 0; s < a; s++) x &= x - 1u;
                    for (int s = 0; s < b; s++) y &= y - 1u;
                    const int ia = valid ? __ffs((int)x) - 1 : 0, ib = valid ? __ffs((int)y) - 1 : 0;
                    const uint32_t ta = (uint32_t)rbc(ht, rb + ia), tb = (uint32_t)rbc(ht, rb + ib);
                    const uint32_t vb = rballot(valid, rb);
                    if (valid) r4_put(q, i, n + __popc(vb & ((1u << r) - 1u)), mk_action(RMJ_CHI, tile, 2, ta, tb), 40 - k);
                    n += __popc(vb);
                }
                nl_set(i, n);
            }
        }
    }
    R4M(54);
    // ---- E: Pass, lengths, stale counts (lane = seat)
    int n_me = r < 4 ? nl_get(r) : 0;
    if (RICH && ron_m) {   // (slot 0 of the seat's list was kept free above; a seat in riichi has no other claim: its list is Ron, Pass)
        if (r < 4 && ((ron_m >> r) & 1u)) q.T->lst[q.row][r][0] = mk_action(RMJ_RON, tile, 0) | ((uint64_t)(KSANMA ? 56 : 79) << 56);
        if (r == 0) G->ron_offer_mask = (uint8_t)ron_m;
    }
    if (rballot(r < 4 && n_me + 1 > R4_LIST, rb)) { R4BAIL(q, 10); return; }
    if (r < 4) {
        G->stale_n[r] = (uint8_t)(n_me > 62 ? 62 : n_me);
        if (n_me > 0) {
            q.T->lst[q.row][r][n_me] = mk_action(RMJ_PASS, RMJ_TILE_NONE, 0) | ((uint64_t)(KSANMA ? 58 : 81) << 56);
            n_me += 1;
        }
    }
    nl_mine = r < 4 ? n_me : 0;
    const uint32_t claim_active = rballot(r < 4 && n_me > 0, rb) & 0xFu;
    wave_sync();
    R4M(55);
    if (claim_active) {
        if (r == 0) {
            G->phase = RMJ_WAIT_RESPONSE;
            G->active_mask = (uint8_t)claim_active;
        }
        wave_sync();
    } else {
        r4_accept_riichi(q);
        r4_check_abortive(q);
        if (q.bail) return;
        const uint32_t tc = G->turn_count + 1u;
        if (r == 0) {
            G->turn_count = tc;
            G->current_player = (uint8_t)(pid + 1 == KNP ? 0 : pid + 1);
        }
        wave_sync();
        r4_deal_next(q, pf);
        if (q.bail) return;
        if (r == 0 && tc >= (uint32_t)KNP) G->is_first_turn = 0;
        wave_sync();
    }
}

__device__ __forceinline__ uint32_t r4_tenpai_keep(const R4& q, const PState* P, const PH& full, int hl);
// _get_legal_actions_internal, WaitAct branch (legal_actions.rs:11-252), for the row's current player.  Everything that
// needs a yaku evaluation or a wait probe (a complete hand, a possible Riichi, a kan in riichi) makes the row bail.
template <bool RICH>
__device__ __forceinline__ void r4_gen_act_legal(R4& q, int& nl_mine) {
    GState* G = q.G;
    const int r = q.r, rb = q.rb;
    const int pid = G->current_player;
    PState* P = &G->p[pid];
    const int hl = P->hand_len, nmelds = P->n_melds;
    const uint32_t pflags = P->flags;
    const bool r_decl = pflags & PF_RIICHI_DECLARED, r_stage = pflags & PF_RIICHI_STAGE;
    const int drawn_tile = G->drawn_tile;
    const bool drawn = drawn_tile != 0xFF;
    const int drawable = G->drawable_count;
    if (hl + 3 * nmelds == 13 || (!RICH && r_stage)) { R4BAIL(q, 11); return; }
    const int ht = r < hl ? (int)P->hand[r] : 0xFF;
    const int hty = ht >> 2;
    int n = 0;
    int d_idx = -1;   // the drawn tile's slot
    // 1. Tsumo: is the drawn type a wait of the 13 other tiles?  (cache, else the cheap refill; a complete hand bails)
    if (drawn && !r_stage) {
        const uint32_t b = rballot(r < hl && ht == drawn_tile, rb);
        const int idx = b ? 31 - __clz((int)b) : -1;
        d_idx = idx;
        const int same_type = __popc(rballot(r < hl && hty == (drawn_tile >> 2), rb));
        if (!(idx >= 0 && same_type <= 4 && (hl - 1) + 3 * nmelds == 13)) { R4BAIL(q, 12); return; }
        if (!(pflags & PF_WAITS_VALID)) {
            // the drawn tile is the last one and the 13 others are normally one sorted run; after a Kita they are not
            const int nx = __builtin_amdgcn_update_dpp(0xFFFF, ht, 0x101 /* row_shl:1 */, 0xf, 0xf, false);
            if (idx != hl - 1 || rballot(r < hl - 2 && ht > nx, rb)) r4_fill_waits13_h<RICH>(q, P, r4_hist(q, P, idx));
            else r4_fill_waits13<RICH>(q, P, hl - 1);
            if (q.bail) return;
        }
        if ((P->waits13 >> (drawn_tile >> 2)) & 1ull) {
            // a complete hand.  Concealed (no meld but Ankan): menzen tsumo is a yaku, so the win is legal without the evaluator
            // (calc.is_win of legal_actions.rs:44-61 is true) - RICH lists Tsumo here; an open hand needs its yaku: full path.
            const bool concealed = rballot(r < nmelds && P->meld_type[r & 3] != RMJ_MELD_ANKAN, rb) == 0u;
            if (!RICH) { R4BAIL(q, 13); return; }
            bool offer = concealed;
            if (!concealed) {   // an open hand needs a yaku (legal_actions.rs:44-61): pass 1 pauses, pass 2 has the evaluator's answer
                if (q.yk_mode == R4_RE_YAKU_TSUMO) offer = (q.yk & 1u) != 0u;
                else if (q.pause_ok) { q.rend = R4_RE_YAKU_TSUMO; return; }
                else { R4BAIL(q, 13); return; }
            }
            if (offer) {
                if (r == 0) r4_put(q, pid, n, mk_action(RMJ_TSUMO, drawn_tile, 0), KSANMA ? 56 : 79);
                n += 1;
            }
        }
    }
    R4M(60);
    // 2. Discards (+ Riichi -> bail)
    const int nforb = P->n_forbidden;
    const bool forb = (nforb > 0 && (P->forbidden[0] >> 2) == hty) || (nforb > 1 && (P->forbidden[1] >> 2) == hty);
    const PH full = r4_hist(q, P, -1);
    if (r_decl) {
        if (drawn) {
            if (r == 0) r4_put(q, pid, n, mk_action(RMJ_DISCARD, drawn_tile, 0), r4_tile_id(drawn_tile >> 2));
            n += 1;
        }
    } else {
        const bool all_closed = rballot(r < nmelds && P->meld_type[r & 3] != RMJ_MELD_ANKAN, rb) == 0u;
        const bool riichi_pre = !r_stage && P->score >= 1000 && (KSANMA ? drawable > 0 : drawable >= 4) && all_closed;
        bool need_tp = r_stage;   // riichi stage: only the discards that keep the hand tenpai (legal_actions.rs:77-100)
        if (riichi_pre) {
            const int sh13 = (drawn && (P->flags & PF_WAITS_VALID)) ? (int)P->sh13 : -1;
            if (sh13 < 2) {
                // tenpai_after_discard: only a 14-tile shanten <= 0 can keep a tenpai 13 (those hands take the probes).  Such a
                // hand has at most TWO isolated tiles (the discard and a tanki) and, as a kokushi shape, thirteen kinds with
                // at most one missing: three isolated tiles and < 12 kinds of terminals / honors rule Riichi out without tables.
                auto yaochu_kinds = [&]() {
                    const uint32_t T9 = 1u | (1u << 24);
                    return __popc((full.a | (full.a >> 1) | (full.a >> 2)) & T9) + __popc((full.b | (full.b >> 1) | (full.b >> 2)) & T9) +
                           __popc((full.c | (full.c >> 1) | (full.c >> 2)) & T9) + __popc((full.d | (full.d >> 1) | (full.d >> 2)) & O7_1);
                };
                // (round 6) and, in standard form, groups of sizes {2}, {1, 1} or {2, 2, 1} mod 3 (r4_group_residues: a tenpai 13 plus one
                // tile) - which three isolated tiles never are; seven pairs need six pairs, thirteen orphans twelve groups.  The drawn
                // tile is merged into the sorted run by its rank; a hand that is not "sorted run + drawn tile" keeps the histogram bound.
                bool may;
                const int nx14 = __builtin_amdgcn_update_dpp(0xFFFF, ht, 0x101 /* row_shl:1 */, 0xf, 0xf, false);
                if (RMJ_GROUP_FILTER14 && q.gf && d_idx == hl - 1 && rballot(r < hl - 2 && ht > nx14, rb) == 0u) {
                    const bool in = r < hl;
                    const int k = in ? r4_key(hty) : 1000;
                    const int kd = rbc(k, rb + hl - 1);
                    const int m = __popc(rballot(r < hl - 1 && k <= kd, rb));   // the drawn tile's place in the run
                    const int ks = __builtin_amdgcn_update_dpp(-1000, k, 0x111 /* row_shr:1 */, 0xf, 0xf, false);
                    const int k14 = !in ? 1000 : (r < m ? k : (r == m ? kd : ks));
                    const int kp = __builtin_amdgcn_update_dpp(-1000, k14, 0x111 /* row_shr:1 */, 0xf, 0xf, false);
                    const bool start = in && k14 - kp > 2;
                    const uint32_t S = rballot(start, rb);
                    const uint32_t gr = r4_group_residues(start, S, hl, r, rb);
                    may = gr == 0x10u || gr == 0x02u || gr == 0x21u;
                    if (!may && hl == 14) {
                        const int kn = __builtin_amdgcn_update_dpp(1000, k14, 0x101 /* row_shl:1 */, 0xf, 0xf, false);
                        may = __popc(rballot(in && k14 != kp && k14 == kn, rb)) >= 6 || (__popc(S) >= 12 && yaochu_kinds() >= 12);
                    }
                } else {
                    may = !(r4_isolated(full) >= 3 && yaochu_kinds() < 12);
                }
                if (may) {
                    if (r4_shanten(q, full, hl / 3) <= 0) {
                        if (!RICH) { R4BAIL(q, 14); return; }
                        need_tp = true;
                    }
                }
            }
        }
        uint32_t tp = 0u;   // bit j: the hand without hand[j] has a wait
        if (r_stage && G->tp_seat == pid && G->tp_step == G->step_count) {   // computed for the Riichi entry one step ago (GState::tp_mask)
            tp = G->tp_mask;
            need_tp = false;
        }
        if (RICH && __ballot(need_tp)) {
            if (need_tp) {
                tp = r4_tenpai_keep(q, P, full, hl);
                if (r == 0 && !r_stage) { G->tp_seat = (uint8_t)pid; G->tp_mask = (uint16_t)tp; G->tp_step = G->step_count + 1u; }
            }
        }
        const bool ok = r < hl && !forb && (!r_stage || ((tp >> r) & 1u));
        const uint32_t vb = rballot(ok, rb);
        if (ok) r4_put(q, pid, n + __popc(vb & ((1u << r) - 1u)), mk_action(RMJ_DISCARD, ht, 0), r4_tile_id(hty));
        n += __popc(vb);
        if (riichi_pre && tp != 0u) {   // legal_actions.rs:113-140
            if (r == 0) r4_put(q, pid, n, mk_action(RMJ_RIICHI, RMJ_TILE_NONE, 0), KSANMA ? 27 : 37);
            n += 1;
        }
    }
    R4M(61);
    // 3. Kan
    if (drawable > 0 && drawn) {
        if (!r_decl && !r_stage) {
            const bool any4 = (((full.a | full.b | full.c) & O9_4) | (full.d & O7_4)) != 0u;
            if (any4) {
                // Ankan of every type held four times, in type order (legal_actions.rs:150-166).  With the hand a sorted run plus
                // the drawn tile, the first lane of a type is the one whose left neighbour differs, and the drawn tile is never
                // the first of a complete set.
                const int next_t = __builtin_amdgcn_update_dpp(0xFFFF, ht, 0x101 /* row_shl:1 */, 0xf, 0xf, false);
                if (rballot(r < hl - 2 && ht > next_t, rb)) { R4BAIL(q, 15); return; }   // (3P: loose tile after a Kita)
                const int prev_ty = __builtin_amdgcn_update_dpp(-1, hty, 0x111 /* row_shr:1 */, 0xf, 0xf, false);
                const bool is4 = r < hl - 1 && hty != prev_ty && ph_cnt(full, hty) == 4;
                const uint32_t ab = rballot(is4, rb);
                if (is4) {
                    const uint32_t lo = (uint32_t)hty * 4u;
                    r4_put(q, pid, n + __popc(ab & ((1u << r) - 1u)), mk_action(RMJ_ANKAN, lo, 4, lo, lo + 1, lo + 2, lo + 3),
                           (KSANMA ? 29 : 42) + r4_tile_id(hty));
                }
                n += __popc(ab);
            }
            const uint32_t pon_lane = rballot(r < nmelds && P->meld_type[r & 3] == RMJ_MELD_PON, rb);
            if (pon_lane) {                               // Kakan: meld order, then hand order
                for (int m = 0; m < nmelds; m++) {
                    if (P->meld_type[m] != RMJ_MELD_PON) continue;
                    const int target = P->meld_tiles[m][0] >> 2;
                    const bool hit = r < hl && hty == target;
                    const uint32_t kb = rballot(hit, rb);
                    if (hit)
                        r4_put(q, pid, n + __popc(kb & ((1u << r) - 1u)),
                               mk_action(RMJ_KAKAN, ht, 3, P->meld_tiles[m][0], P->meld_tiles[m][1], P->meld_tiles[m][2]),
                               (KSANMA ? 29 : 42) + r4_tile_id(target));
                    n += __popc(kb);
                }
            }
        } else if (r_decl) {
            if (ph_cnt(full, drawn_tile >> 2) == 4) { R4BAIL(q, 16); return; }   // ankan after riichi: wait probes
        }
    }
    // 4. Kyushu kyuhai: first turn, no calls, nine kinds of terminals and honors (the type set is OR-ed over the row)
    if (G->is_first_turn && !r_stage && (G->p[0].n_melds | G->p[1].n_melds | G->p[2].n_melds | G->p[3].n_melds) == 0) {
        const bool term = r < hl && is_terminal_tile136(ht);
        const uint32_t lo = row_or16(term && hty < 32 ? 1u << hty : 0u), hi = row_or16(term && hty >= 32 ? 1u << (hty - 32) : 0u);
        const int kinds = __popc((uint32_t)rbc((int)lo, rb + 15)) + __popc((uint32_t)rbc((int)hi, rb + 15));
        if (kinds >= 9) {
            if (r == 0) r4_put(q, pid, n, mk_action(RMJ_KYUSHU, RMJ_TILE_NONE, 0), KSANMA ? 57 : 80);
            n += 1;
        }
    }
    // 5. Kita (state_3p/sanma.rs:146-169)
    if (KSANMA && drawn && drawable > 0) {
        const bool hit = r < hl && hty == 30;
        const uint32_t kb = rballot(hit, rb);
        if (hit) r4_put(q, pid, n + __popc(kb & ((1u << r) - 1u)), mk_action(RMJ_KITA, ht, 0), 59);
        n += __popc(kb);
    }
    if (n > R4_ACT_CAP) { R4BAIL(q, 17); return; }
    nl_mine = r == pid ? n : 0;
    wave_sync();
}

// ---------------------------------------------------------------------------------------------------------------------------
// Round ends in ROW FORM (round 4).  Until here an exhaustive draw - 97 % of the exits of a RandomAgent rollout - sent its row to the
// full path: the whole step again from the HBM record, then ~42 k cycles of wave-uniform code for ONE game while the three other rows
// of the wave waited.  Here the rows that end a round stay in tier 0:
//   * _trigger_ryukyoku (state/mod.rs:1846-1968, exhaustive draw): the four tenpai flags from the wait caches (the discarder's is
//     refilled by the row-form probe), nagashi mangan / tenpai payments as lane = seat arithmetic, the ryukyoku event;
//   * _initialize_next_round (state/mod.rs:1595-1688): the end-of-game decision per row;
//   * RiichiEnv.reset defaults for a finished game under auto-reset (env.rs:799-851);
//   * _initialize_round (state/mod.rs:1695-1844): record reset per row, then - one game at a time with all 64 lanes, it is 136
//     elements wide - the wall (keys, a 128-bucket counting sort with byte counters, every tile scatters itself to the wall slab, to
//     its hand slot, to the indicator), the four hands sorted by counting (lane = 16 * seat + slot), start_kyoku / tehai / tsumo.
// The dealer's first list then comes from r4_gen_act_legal like any other WaitAct state, and the row is published with the others.
// Events of a round end go straight to the ring (up to nine per step: the staging area holds four).
// Out of line: its registers and code are not the hot path's.
// one MJAI record of the row's game straight to the ring: lane r < 8 holds dword r
__device__ __forceinline__ void r4_emit_now(R4& q, bool on, uint32_t w) {
    if (q.E->skip_log) return;
    const uint32_t evc = q.G->ev_count;
    wave_sync();
    if (on) {
        uint32_t* dst = reinterpret_cast<uint32_t*>(q.E->events + (size_t)q.g * (q.E->ring_mask + 1u) + (evc & q.E->ring_mask));
        if (q.r < 8) dst[q.r] = q.r == 7 ? ((w & 0x00FFFFFFu) | ((uint32_t)KNP << 24)) : w;
        if (q.r == 0) q.G->ev_count = evc + 1u;
    }
    wave_sync();
}
// RICH: with the settlements of the rich tier (the lean tier's Tsumo / Ron actions bail to the full path: its copy has no evaluator)
#ifdef RMJ_RE_PROF
#define RE_MARK(k) do { const unsigned long long t__ = __builtin_amdgcn_s_memrealtime(); if (lane == 0) atomicAdd(&rmj::g_re_prof[k], t__ - re_t); re_t = t__; } while (0)
#else
#define RE_MARK(k) do {} while (0)
#endif
template <bool RICH>
__device__ __noinline__ void r4_round_end(const Env* Ep, uint32_t g0) {
    const int lane = threadIdx.x & 63, row = lane >> 4, r = lane & 15, rb = lane & 48;
#ifdef RMJ_RE_PROF
    unsigned long long re_t = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) atomicAdd(&rmj::g_re_prof[16], re_t);   // (sum of entry times: minus the callers' sum of call times = the prologue)
#endif
    Quad4Shared& sh = g_q4;
    const uint32_t mode = sh.rmode[row];
    CEnv& E = *(CEnv*)uni_ptr(Ep);
    g0 = uni(g0);
    GState* G = &sh.st[row];
    R4 q;
    q.G = G; q.T = &sh.u.t; q.E = &E; q.lane = lane; q.r = r; q.rb = rb; q.row = row; q.g = g0 + (uint32_t)row;
    q.live = mode != 0u; q.bail = false; q.cont = 0; q.rend = mode; q.evn = 0; q.dirty = 0xFu; q.pause_ok = false; q.yk = 0u; q.yk_mode = 0u; q.gf = false;
    const bool draw = mode == R4_RE_DRAW, restart = mode == R4_RE_RESTART;
    const bool win_t = mode == R4_RE_WIN_TSUMO, win_r = mode == R4_RE_WIN_RON, win = win_t || win_r;
    bool newround = restart;            // the row's game starts a round below, with these parameters (row-uniform)
    int n_oya = 0, n_rw = 0, n_honba = 0;
    uint32_t n_sticks = 0u;
    const bool seat = r < KNP;
    const int oya = G->oya;
    bool oya_won = false;               // _initialize_next_round's first argument: the dealer won / is tenpai at the draw
    if (RICH && RMJ_ROW_SETTLE && __ballot(win)) {
        // ---- settlement of a Tsumo (state/mod.rs:690-880) or of the Ron answers (state/mod.rs:945-1142), rows side by side: one winner
        // per row and pass (the Tsumo, then the Ron winners by distance from the discarder), the evaluation in row form with the settlement's
        // conditions (r4_seat_eval<true>), the payments by lane = seat.
        const uint8_t* Wg = E.wall + (size_t)q.g * RMJ_WALL_STRIDE;
        const uint32_t rules = E.rule_bits;
        const bool from_discard = G->last_discard_pid != 0xFF;
        const int target = (win_r && from_discard) ? (int)G->last_discard_pid : (int)G->current_player;
        const int win_tile = win_t ? (G->drawn_tile != 0xFF ? (int)G->drawn_tile : 0) : (from_discard ? (int)G->last_discard_tile : 0);
        const uint32_t ron_m = win_r ? (sh.yk[row] & 0xFu) : 0u;
        const uint32_t honba0 = G->honba;
        int32_t total_d = 0;            // lane = seat: what the round's settlements pay the seat
        bool honba_taken = false, deposit_taken = false;   // the first winner's hand counts the honba, the first win takes the deposit
        for (int dist = 0; dist < KNP; dist++) {
            int w = target + dist;
            w = w >= KNP ? w - KNP : w;
            const bool on = dist == 0 ? win_t : (win_r && ((ron_m >> w) & 1u) != 0u);
            if (!__ballot(on)) continue;
            const PState& Wp = G->p[w & 3];
            const uint32_t wfl = on ? (uint32_t)Wp.flags : 0u;
            const bool riichi = (wfl & PF_RIICHI_DECLARED) != 0u;
            uint32_t cf = (riichi ? CF_RIICHI : 0u) | ((wfl & PF_DOUBLE_RIICHI) ? CF_DOUBLE_RIICHI : 0u) | ((wfl & PF_IPPATSU) ? CF_IPPATSU : 0u);
            const bool last_tile = G->drawable_count == 0 && !G->is_rinshan;
            if (win_t) {
                cf |= CF_TSUMO;
                if (last_tile) cf |= CF_HAITEI;
                if (G->is_rinshan) cf |= CF_RINSHAN;
                const bool no_melds = (G->p[0].n_melds | G->p[1].n_melds | G->p[2].n_melds | G->p[3].n_melds) == 0;
                if (G->is_first_turn && no_melds) cf |= CF_FIRST_TURN;   // quirk Q5 (settlement form)
            } else {
                if (last_tile) cf |= CF_HOUTEI;
                // a pending kita is a chankan-style claim but awards no chankan yaku (state_3p/mod.rs:896-902)
                if (G->pending_kan_pid != 0xFF && a_type(G->pending_kan_action) != RMJ_KITA) cf |= CF_CHANKAN;
            }
            const uint32_t hb = honba_taken ? 0u : honba0;
            if (on) honba_taken = true;
            E4Out o = r4_seat_eval<true>(on ? 1u : 0u, (uint32_t)w, (uint32_t)win_tile, cf, hb, riichi, Wg);
            const bool w_oya = w == oya;
            if (o.yakuman && o.han > 13) {   // double-yakuman cap, state/mod.rs:720-745 / 1005-1030
                int cap = 0;
                if (((o.ym >> 47) & 1ull) && !(rules & RMJ_RULE_JUNSEI_CHUUREN_DOUBLE)) cap += 13;
                if (((o.ym >> 48) & 1ull) && !(rules & RMJ_RULE_SUUANKOU_TANKI_DOUBLE)) cap += 13;
                if (((o.ym >> 49) & 1ull) && !(rules & RMJ_RULE_KOKUSHI13_DOUBLE)) cap += 13;
                if (((o.ym >> 50) & 1ull) && !(rules & RMJ_RULE_DAISUUSHII_DOUBLE)) cap += 13;
                if (cap > 0) {
                    const int hh = o.han > cap ? o.han - cap : 0;
                    o.han = hh < 13 ? 13 : hh;
                    const ScoreOut s2 = calc_score((uint32_t)o.han, 0u, w_oya, win_t, hb, (uint32_t)KNP);
                    o.ron = s2.ron; o.tsumo_oya = s2.tsumo_oya; o.tsumo_ko = s2.tsumo_ko;
                }
            }
            const bool won = on && o.is_win;   // (a listed Tsumo / Ron wins: the list entry was evaluated on this hand; ura and Norths only add han)
            // yakuman values and pao liability (37 daisangen, 50 daisuushii) of the winner
            int total_val = 0, pao_val = 0, pao_seat = -1;
            if (won && o.yakuman) {
                const uint64_t ids = (0x7FFull << 35) | (0xFull << 47);
                total_val = __popcll(o.ym & ids);
                if (((o.ym >> 47) & 1ull) && (rules & RMJ_RULE_JUNSEI_CHUUREN_DOUBLE)) total_val += 1;
                if (((o.ym >> 48) & 1ull) && (rules & RMJ_RULE_SUUANKOU_TANKI_DOUBLE)) total_val += 1;
                if (((o.ym >> 49) & 1ull) && (rules & RMJ_RULE_KOKUSHI13_DOUBLE)) total_val += 1;
                const int v50 = (rules & RMJ_RULE_DAISUUSHII_DOUBLE) ? 2 : 1;
                if (((o.ym >> 50) & 1ull) && v50 == 2) total_val += 1;
                if (((o.ym >> 37) & 1ull) && Wp.pao37 != 0xFF) { pao_val += 1; pao_seat = Wp.pao37; }
                if (((o.ym >> 50) & 1ull) && Wp.pao50 != 0xFF) { pao_val += v50; pao_seat = Wp.pao50; }
            }
            const int32_t sticks = (won && !deposit_taken) ? (int32_t)(G->riichi_sticks * 1000u) : 0;
            int32_t d = 0;              // lane = seat: this winner's payments
            {
                if (won && win_t && r < KNP && r != w) {
                    if (pao_val > 0) {
                        // state_3p/mod.rs:713-721: (np-1)*16000 for the dealer, 16000+(np-2)*8000 otherwise
                        const int32_t unit = w_oya ? (KNP - 1) * 16000 : 16000 + (KNP - 2) * 8000;
                        const int32_t honba_total = (int32_t)hb * (KNP - 1) * 100;
                        if (rules & RMJ_RULE_PAO_LIABILITY_ONLY) {
                            const int32_t non = total_val - pao_val;
                            if (non > 0) d -= (w_oya || r == oya) ? non * 16000 : non * 8000;
                            if (r == pao_seat) d -= pao_val * unit + honba_total;
                        } else if (r == pao_seat) {
                            d -= total_val * unit + honba_total;
                        }
                    } else {
                        d = -(int32_t)(w_oya ? o.tsumo_ko : (r == oya ? o.tsumo_oya : o.tsumo_ko));
                    }
                }
                const int32_t tw = -(int32_t)e4_rsum((uint32_t)d, rb);   // what the others pay
                if (won && win_t && r == w) d = tw + sticks;
            }
            if (won && win_r) {
                const int32_t score = (int32_t)o.ron;
                int payer = target;
                int32_t pao_amt = 0;
                if (pao_seat >= 0) {
                    payer = pao_seat;
                    const int32_t unit = w_oya ? 48000 : 32000;
                    const int32_t honba_ron = (int32_t)hb * (KNP - 1) * 100;
                    const int32_t split_base = (rules & RMJ_RULE_PAO_LIABILITY_ONLY) ? pao_val * unit : total_val * unit;
                    pao_amt = split_base / 2 + honba_ron;
                }
                if (r == w) d += score + sticks;
                if (r == payer) d -= pao_amt;
                if (r == target) d -= score - pao_amt;
            }
            if (won) {
                total_d += d;
                if (w_oya) oya_won = true;
                if (!deposit_taken && r == 0) G->riichi_sticks = 0u;
                deposit_taken = true;
            }
            wave_sync();
            {   // win_results.insert(seat, val) (state/mod.rs:855-863, 1100-1107): the capped result with its ordered yaku list, the pao payer
                uint8_t* yl = reinterpret_cast<uint8_t*>(sh.rs + 6 * row);
                if (r < 5) sh.rs[6 * row + r] = 0u;
                wave_sync();
                const int ny = e4_yaku_list(o.kind, o.ym, yl, r, rb);
                wave_sync();
                if (won && r < 12) {
                    uint32_t v;
                    if (r == 0) v = 1u | (o.yakuman ? 0x100u : 0u) | (o.shape ? 0x10000u : 0u) | ((uint32_t)ny << 24);
                    else if (r < 6) v = sh.rs[6 * row + r - 1];
                    else if (r == 6) v = (uint32_t)o.han;
                    else if (r == 7) v = (uint32_t)o.fu;
                    else if (r == 8) v = o.ron;
                    else if (r == 9) v = o.tsumo_oya;
                    else if (r == 10) v = o.tsumo_ko;
                    else v = (uint32_t)(pao_seat & 0xFF);
                    reinterpret_cast<uint32_t*>(E.win + (size_t)q.g * 4 + w)[r] = v;
                }
                if (won && r == 0) G->win_mask |= (uint8_t)(1u << w);
                wave_sync();
            }
            {   // the hora event: this winner's deltas, the ura indicators of a riichi hand
                const int idx = KSANMA ? 9 + 2 * r : 5 + 2 * r;
                const bool uv = won && riichi && r < 5 && r < (int)G->n_dora && (KSANMA || idx < (int)G->live_end);
                uint32_t ub = 0u;
                if (uv) {
                    const uint32_t ww = __hip_atomic_load(reinterpret_cast<const uint32_t*>(Wg) + (idx >> 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ub = (ww >> (8 * (idx & 3))) & 0xFFu;
                }
                const uint32_t nu = (uint32_t)__popc(rballot(uv, rb));
                const uint32_t u0 = (uint32_t)rbc((int)ub, rb), u1 = (uint32_t)rbc((int)ub, rb + 1), u2 = (uint32_t)rbc((int)ub, rb + 2);
                const uint32_t u3 = (uint32_t)rbc((int)ub, rb + 3), u4 = (uint32_t)rbc((int)ub, rb + 4);
                const int32_t dv = rbc(d, rb + ((r - 2) & 3));
                uint32_t wv = 0u;
                if (r == 0) wv = (uint32_t)RMJ_EV_HORA | ((uint32_t)w << 8) | ((uint32_t)target << 16);
                else if (r >= 2 && r < 6) wv = (uint32_t)dv;
                else if (r == 6) wv = (win_t ? 1u : 0u) | (nu << 8) | (u0 << 16) | (u1 << 24);
                else if (r == 7) wv = u2 | (u3 << 8) | (u4 << 16);
                r4_emit_now(q, won, wv);
            }
        }
        if (win && r < 4) {
            PState& S4 = G->p[r];
            S4.score += total_d;
            S4.score_delta = total_d;
        }
        wave_sync();
    }
    if (__ballot(draw)) {
        // ---- tenpai of the seats (HandEvaluator::is_tenpai through the wait cache, like seat_tenpai): stale caches first
        {
            const PState& S0 = G->p[r & 3];
            uint32_t need_m = draw ? (rballot(seat && (S0.hand_len + 3 * S0.n_melds == 13) && !(S0.flags & PF_WAITS_VALID), rb) & 0xFu) : 0u;
            while (__ballot(need_m != 0u)) {
                if (need_m) {
                    const int i = __ffs((int)need_m) - 1;
                    need_m &= need_m - 1u;
                    PState* Q = &G->p[i];
                    r4_fill_waits13<true>(q, Q, Q->hand_len);
                }
            }
        }
        PState& S4 = G->p[r & 3];
        RE_MARK(12);   // entry + the refills of stale wait caches
        const uint32_t tenpai_m = rballot(draw && seat && (S4.hand_len + 3 * S4.n_melds == 13) && S4.waits13 != 0ull, rb) & 0xFu;
        const uint32_t nag_m = rballot(draw && seat && (S4.flags & PF_NAGASHI), rb) & 0xFu;
        int reason = RMJ_RK_EXHAUSTIVE;
        if (draw && seat) {
            int32_t sc = S4.score, sd = S4.score_delta;
            if (nag_m) {   // nagashi mangan: every eligible seat takes a mangan tsumo from the others (calculate_score(5, 30, ., true, 0))
#pragma unroll
                for (int w = 0; w < KNP; w++) {
                    if ((nag_m >> w) & 1u) {
                        const int32_t d = r == w ? (w == oya ? 4000 * (KNP - 1) : 4000 + 2000 * (KNP - 2)) : -((w == oya || r == oya) ? 4000 : 2000);
                        sc += d; sd += d;
                    }
                }
            } else {
                const int ntp = __popc(tenpai_m);
                if (ntp > 0 && ntp < KNP) {
                    const int pool = KSANMA ? 2000 : 3000;   // state_3p/game_mode.rs:39-41
                    const int pk = ntp == 1 ? pool : (ntp == 2 ? pool / 2 : pool / 3), nn = KNP - ntp;
                    const int pn = nn == 1 ? pool : (nn == 2 ? pool / 2 : pool / 3);
                    const int32_t d = ((tenpai_m >> r) & 1u) ? pk : -pn;
                    sc += d; sd = d;
                }
            }
            S4.score = sc; S4.score_delta = sd;
        }
        if (nag_m) reason = RMJ_RK_NAGASHI;
        wave_sync();
        if (draw) oya_won = nag_m ? ((nag_m >> oya) & 1u) != 0u : ((tenpai_m >> oya) & 1u) != 0u;
        {   // the ryukyoku event: deltas = the seats' score_delta
            const int32_t dv = (r >= 2 && r < 2 + KNP) ? G->p[(r - 2) & 3].score_delta : 0;
            const uint32_t w = r == 0 ? (uint32_t)RMJ_EV_RYUKYOKU : (r == 6 ? (uint32_t)reason : (uint32_t)dv);
            r4_emit_now(q, draw, (r == 1 || r == 7) ? 0u : w);
        }
    }
    RE_MARK(13);   // payments + the ryukyoku record
    const bool decide = draw || win;
    if (__ballot(decide)) {
        // ---- _initialize_next_round(oya_won, is_draw) (state/mod.rs:1595-1700)
        {
            const int32_t goal = KSANMA ? 40000 : 30000;
            const int32_t s0 = G->p[0].score, s1 = G->p[1].score, s2 = G->p[2].score, s3 = KNP > 3 ? G->p[3].score : 0;
            const bool neg = s0 < 0 || s1 < 0 || s2 < 0 || (KNP > 3 && s3 < 0);
            int32_t mx = s0 > s1 ? s0 : s1;
            mx = mx > s2 ? mx : s2;
            if (KNP > 3) mx = mx > s3 ? mx : s3;
            const int32_t ds = oya == 0 ? s0 : (oya == 1 ? s1 : (oya == 2 ? s2 : s3));
            bool top = true;
            top = top && (oya == 0 || ds > s0 || (ds == s0 && oya <= 0));
            top = top && (oya == 1 || ds > s1 || (ds == s1 && oya <= 1));
            top = top && (oya == 2 || ds > s2 || (ds == s2 && oya <= 2));
            if (KNP > 3) top = top && (oya == 3 || ds > s3 || (ds == s3 && oya <= 3));
            const uint32_t gm = E.game_mode;
            const int rw = G->round_wind;
            bool last_regular = false;
            if (gm == 1u || gm == 4u) last_regular = rw == 0 && oya == KNP - 1;
            if (gm == 2u || gm == 5u) last_regular = rw == 1 && oya == KNP - 1;
            bool end = neg || (oya_won && last_regular && top && ds >= goal);
            int next_honba = G->honba, next_oya = oya, next_rw = rw;
            next_honba = (oya_won || draw) ? (next_honba + 1 > 255 ? 255 : next_honba + 1) : 0;   // (a draw keeps counting whoever deals next)
            if (!oya_won) {
                next_oya = next_oya + 1 == KNP ? 0 : next_oya + 1;
                if (next_oya == 0) next_rw += 1;
            }
            if (!end) {
                if (gm == 1u || gm == 4u) end = next_rw >= 1 && (mx >= goal || next_rw > 1);
                else if (gm == 2u || gm == 5u) end = next_rw >= 2 && (mx >= goal || next_rw > 2);
                else if (gm == 0u || gm == 3u) end = true;
                else end = next_rw >= 1;
            }
            if (decide && end && r == 0) G->is_done = 1;   // process_end_game
            r4_emit_now(q, decide, r == 0 ? (uint32_t)RMJ_EV_END_KYOKU : 0u);
            if (__ballot(decide && end)) r4_emit_now(q, decide && end, r == 0 ? (uint32_t)RMJ_EV_END_GAME : 0u);
            if (decide && !end) {
                newround = true;
                n_oya = next_oya; n_rw = next_rw; n_honba = next_honba; n_sticks = G->riichi_sticks;
            }
        }
    }
    RE_MARK(14);   // next-round decision + end_kyoku / end_game records
    if (__ballot(restart)) {   // GameState::reset clears the logs (state/mod.rs:171-187), then start_game
        if (restart) {
            const uint32_t evc0 = G->ev_count;
            if (r == 0) G->ev_base = evc0;
            if (r < 4) { G->obs_from[r] = evc0; G->obs_upto[r] = evc0; }
        }
        wave_sync();
        r4_emit_now(q, restart, r == 0 ? (uint32_t)RMJ_EV_START_GAME : 0u);
    }
#ifdef RMJ_RE_PROF
    {
        const unsigned long long b0 = __ballot(draw && r == 0), b1 = __ballot(restart && r == 0), b2 = __ballot(newround && r == 0), b3 = __ballot(q.live && G->is_done && r == 0);
        if (lane == 0) {
            atomicAdd(&rmj::g_re_prof[20], (unsigned long long)__popcll(b0)); atomicAdd(&rmj::g_re_prof[21], (unsigned long long)__popcll(b1));
            atomicAdd(&rmj::g_re_prof[22], (unsigned long long)__popcll(b2)); atomicAdd(&rmj::g_re_prof[23], (unsigned long long)__popcll(b3));
        }
    }
#endif
    if (!__ballot(newround)) return;
    // ---- _initialize_round, the per-row part: PlayerState::reset_round (state/player.rs:66-86) by lane = seat, the globals by lane 0
    if (newround) {
        if (r < 4) {
            PState& P = G->p[r];
            P.hand_len = 0; P.n_melds = 0; P.n_discards = 0;
            P.flags = PF_NAGASHI;
            P.pao37 = 0xFF; P.pao50 = 0xFF;
            P.n_forbidden = 0;
            P.riichi_decl_idx = 0xFF; P.riichi_sutehai = 0xFF; P.last_tedashi = 0xFF;
            P.score_delta = 0;
            P.discard_from_hand_bits = 0; P.discard_is_riichi_bits = 0;
            P.discard_type_mask = 0;
            P.n_kita = 0;
            if (restart && r < KNP) P.score = KSANMA ? 35000 : 25000;   // state_3p/game_mode.rs:31-33
            G->stale_n[r] = 0;
        }
        if (r == 0) {
            G->oya = (uint8_t)n_oya; G->kyoku_idx = (uint8_t)n_oya; G->current_player = (uint8_t)n_oya;
            G->honba = (uint8_t)n_honba; G->riichi_sticks = n_sticks; G->round_wind = (uint8_t)n_rw;
            G->is_done = 0;
            G->pending_kan_pid = 0xFF; G->pending_kan_action = 0;
            G->is_rinshan = 0; G->rinshan_count = 0; G->pending_kan_dora = 0;
            G->is_first_turn = 1; G->riichi_pending = 0xFF; G->turn_count = 0;
            G->last_discard_pid = 0xFF; G->last_discard_tile = 0;
            G->ron_offer_mask = 0; G->win_mask = 0;
            G->tp_seat = 0xFF;
            G->wall_total = KSANMA ? 108 : 136;
            G->n_dora = 1;
        }
    }
    wave_sync();
    RE_MARK(15);   // restart record + the per-row round reset
    // ---- the wall and the deal, one game at a time with all 64 lanes
#ifdef RMJ_RE_PROF
    const unsigned long long pw0 = __builtin_amdgcn_s_memrealtime();
    { const unsigned long long bn = __ballot(newround && r == 0); if (lane == 0) atomicAdd(&rmj::g_re_prof[6], (unsigned long long)__popcll(bn)); }
#endif
    constexpr int N = KSANMA ? 108 : 136;
    uint32_t* const cnt = sh.rs;                                    // [32] = 128 byte counters, then the bucket offsets in place
    uint16_t* const gk = reinterpret_cast<uint16_t*>(sh.rs + 32);   // [136] sort words grouped by bucket: key bits 56..49 | element index
    uint64_t todo = __ballot(newround && r == 0);
    while (todo) {
        const int br = (__ffsll((long long)todo) - 1) >> 4;
        todo &= todo - 1ull;
        GState* Gb = &sh.st[br];
        const uint32_t gb = g0 + (uint32_t)br;
        uint8_t* Wg = E.wall + (size_t)gb * RMJ_WALL_STRIDE;
        const int oya = Gb->oya;
        const uint32_t hidx = Gb->hand_index;
        const uint64_t hs = sm64(Gb->wall_seed + (uint64_t)hidx);   // the build's seed -> wall (shuffle_wall): ids sorted by (key, id)
        // tile `id` has rank `rk` in w (w[rk] = id): the wall row in load_wall orientation, the indicator, and the deal:
        // pop #n = W[N - 1 - n] = the element of rank n; three rounds of four tiles per seat from the dealer, then one each
        auto place = [&](int rk, int id) {
            const int wpos = N - 1 - rk;
            Wg[wpos] = (uint8_t)id;
            if (wpos == (KSANMA ? 8 : 4)) Gb->dora[0] = (uint8_t)id;   // state_3p/wall.rs:104-112
            const int n = rk;
            if (n < 12 * KNP) {
                const int rr = n >= 8 * KNP ? 2 : (n >= 4 * KNP ? 1 : 0), rem = n - rr * 4 * KNP;
                int p = (rem >> 2) + oya;
                p = p >= KNP ? p - KNP : p;
                Gb->p[p].hand[rr * 4 + (rem & 3)] = (uint8_t)id;
            } else if (n < 13 * KNP) {
                int p = (n - 12 * KNP) + oya;
                p = p >= KNP ? p - KNP : p;
                Gb->p[p].hand[12] = (uint8_t)id;
            } else if (n == 13 * KNP) {   // the dealer's first draw
                Gb->p[oya].hand[13] = (uint8_t)id;
                Gb->drawn_tile = (uint8_t)id;
            }
        };
        if (E.rule_bits & RMJ_RULE_REFERENCE_RNG) {   // the reference's own seed -> wall (rmj_refrng.hip.h)
            uint8_t* const wl = reinterpret_cast<uint8_t*>(sh.rs) + 256;   // 64 dwords of scratch (key stream, then idx[N]), then w[N]: 392 of the 400 bytes
            const uint64_t salt = refrng_wall<N, KSANMA>(hs, lane, sh.rs, wl);
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const int i = lane + 64 * k;
                if (i < N) place(i, wl[i]);
            }
            if (lane < (RMJ_WALL_STRIDE - N)) Wg[N + lane] = (N + lane >= 136) ? (uint8_t)(salt >> (8 * (N + lane - 136))) : (uint8_t)0;
            if (lane == 0) Gb->wall_meta = 1;
        } else {
        if (lane < 32) cnt[lane] = 0u;
        uint64_t key[3];
        uint32_t pos[3];
#pragma unroll
        for (int k = 0; k < 3; k++) key[k] = sm64(hs + (uint64_t)(lane + 64 * k) * 0x9E3779B97F4A7C15ull);
        wave_sync();
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const uint32_t b = (uint32_t)(key[k] >> 57);
            pos[k] = 0u;
            if (lane + 64 * k < N) pos[k] = (atomicAdd(&cnt[b >> 2], 1u << (8u * (b & 3u))) >> (8u * (b & 3u))) & 0xFFu;
        }
        wave_sync();
        {   // exclusive prefix sum over the 128 byte counters, two per lane, written back in place (offsets <= 136 fit a byte)
            const uint32_t wv = cnt[lane >> 1];
            const uint32_t v0 = (wv >> (16u * (lane & 1))) & 0xFFu, v1 = (wv >> (16u * (lane & 1) + 8u)) & 0xFFu;
            uint32_t incl = v0 + v1;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t up = (uint32_t)__shfl_up((int)incl, off, 64);
                if (lane >= off) incl += up;
            }
            const uint32_t excl = incl - (v0 + v1);
            wave_sync();
            reinterpret_cast<uint16_t*>(cnt)[lane] = (uint16_t)(excl | ((excl + v0) << 8));
        }
        wave_sync();
        const uint8_t* off8 = reinterpret_cast<const uint8_t*>(cnt);
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int i = lane + 64 * k;
            if (i < N) gk[off8[key[k] >> 57] + pos[k]] = (uint16_t)((((uint32_t)(key[k] >> 49) & 0xFFu) << 8) | (uint32_t)i);
        }
        wave_sync();
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int i = lane + 64 * k;
            if (i < N) {
                const uint32_t b = (uint32_t)(key[k] >> 57);
                const uint32_t kw = (((uint32_t)(key[k] >> 49) & 0xFFu) << 8) | (uint32_t)i;
                const int lo = off8[b], hi = b == 127u ? N : (int)off8[b + 1u];
                int rk = lo;
                for (int t = lo; t < hi; t++) {
                    const uint32_t kt = gk[t];
                    bool less = kt < kw;
                    if ((kt >> 8) == (kw >> 8) && kt != kw) {   // 15 key bits equal (one bucket mate in 256): the full keys decide
                        const uint32_t it = kt & 0xFFu;
                        const uint64_t kf = sm64(hs + (uint64_t)it * 0x9E3779B97F4A7C15ull);
                        less = kf < key[k] || (kf == key[k] && it < (uint32_t)i);
                    }
                    rk += less ? 1 : 0;
                }
                place(rk, (KSANMA && i >= 4) ? i + 28 : i);   // i-th id of the tile universe (3P: no 2m-8m, types.rs:378-382)
            }
        }
        if (lane < (RMJ_WALL_STRIDE - N)) Wg[N + lane] = 0;
        }
        wave_sync();
        {   // the four hands sorted by counting: lane = 16 * seat + slot
            const int sp = lane >> 4, sl = lane & 15;
            const bool in = sp < KNP && sl < 13;
            const int t = in ? (int)Gb->p[sp & 3].hand[sl] : 0xFFFF;
            int rk = 0;
#pragma unroll
            for (int k = 0; k < 13; k++) rk += rbc(t, rb + k) < t ? 1 : 0;
            wave_sync();
            if (in) Gb->p[sp & 3].hand[rk] = (uint8_t)t;
            if (sp < KNP && sl == 0) Gb->p[sp & 3].hand_len = sp == oya ? 14 : 13;
        }
        if (lane == 0) {
            Gb->hand_index = hidx + 1u;
            Gb->live_end = (uint8_t)(N - 13 * KNP - 1);
            Gb->drawable_count = (uint8_t)(N - 13 * KNP - 14 - 1);
            Gb->needs_tsumo = 0;
            Gb->phase = RMJ_WAIT_ACT;
            Gb->active_mask = (uint8_t)(1u << oya);
        }
        wave_sync();
        if (!E.skip_log) {   // start_kyoku + its two tehai records + the dealer's tsumo
            const uint32_t evc = Gb->ev_count, ring = E.ring_mask + 1u;
            RmjEvent* ring0 = E.events + (size_t)gb * ring;
            if (lane < 8) {
                const uint32_t ks = Gb->riichi_sticks;
                uint32_t w = 0u;
                if (lane == 0) w = (uint32_t)RMJ_EV_START_KYOKU | ((uint32_t)oya << 8) | ((uint32_t)(oya + 1) << 16) | ((uint32_t)Gb->dora[0] << 24);
                if (lane == 1) w = ((uint32_t)Gb->round_wind & 3u) | ((uint32_t)Gb->honba << 8) | ((ks & 0xFFu) << 16) | (((ks >> 8) & 0xFFu) << 24);
                if (lane >= 2 && lane < 2 + KNP) w = (uint32_t)Gb->p[(lane - 2) & 3].score;
                if (lane == 7) w = (uint32_t)KNP << 24;
                reinterpret_cast<uint32_t*>(ring0 + (evc & E.ring_mask))[lane] = w;
            }
            {   // lane = byte of the two tehai records
                const int rec = lane >> 5, b = lane & 31;
                uint32_t v = 0u;
                if (b == 0) v = RMJ_EV_TEHAI;
                if (b == 1) v = (uint32_t)rec;
                if (b >= 4 && b < 30) {
                    const int j = b - 4, sp = 2 * rec + (j >= 13 ? 1 : 0), k = j >= 13 ? j - 13 : j;
                    v = sp < KNP ? Gb->p[sp & 3].hand[k] : 0u;
                }
                if (b == 31) v = (uint32_t)KNP;
                reinterpret_cast<uint8_t*>(ring0 + ((evc + 1u + (uint32_t)rec) & E.ring_mask))[b] = (uint8_t)v;
            }
            if (lane < 8) {
                uint32_t w = 0u;
                if (lane == 0) w = (uint32_t)RMJ_EV_TSUMO | ((uint32_t)oya << 8) | ((uint32_t)Gb->drawn_tile << 24);
                if (lane == 7) w = (uint32_t)KNP << 24;
                reinterpret_cast<uint32_t*>(ring0 + ((evc + 3u) & E.ring_mask))[lane] = w;
            }
            wave_sync();
            if (lane == 0) Gb->ev_count = evc + 4u;
        }
        wave_sync();
    }
#ifdef RMJ_RE_PROF
    if (lane == 0) { const unsigned long long te = __builtin_amdgcn_s_memrealtime(); atomicAdd(&rmj::g_re_prof[5], te - pw0); atomicAdd(&rmj::g_re_prof[18], te); }   // ([18]: sum of exit times of the calls that dealt)
#endif
}

// The evaluator between the passes of a step (out of line; no call sits in the step function itself - a call there costs every step
// its callee-saved registers): rows that paused for a yaku check (R4_RE_YAKU_CLAIMS: the seats without a riichi that wait on the
// discard; R4_RE_YAKU_TSUMO: the drawer's complete open hand) get their answers into Quad4Shared::yk.  One loop serves both kinds: each
// pass evaluates one (seat, tile) per row.
__device__ __noinline__ void r4_yaku_answers() {
    const int lane = threadIdx.x & 63, row = lane >> 4, r = lane & 15, rb = lane & 48;
    Quad4Shared& sh = g_q4;
    const GState* G = &sh.st[row];
    const uint32_t mode = sh.rmode[row];
    const bool claims = mode == R4_RE_YAKU_CLAIMS, tsumo = mode == R4_RE_YAKU_TSUMO;
    uint32_t todo = 0u;
    const int tile = claims ? (int)G->last_discard_tile : (int)G->drawn_tile;
    if (__ballot(claims)) {   // the candidates of section B of r4_resolve_discard (lane = seat)
        const int pid = G->last_discard_pid, tt = tile >> 2;
        const PState& S4 = G->p[r & 3];
        const bool other = claims && r < KNP && r != pid;
        const uint32_t qfl = S4.flags;
        const bool holds13 = other && (S4.hand_len + 3 * S4.n_melds == 13);
        const uint64_t W = holds13 ? S4.waits13 : 0ull;
        const uint64_t dtm = S4.discard_type_mask;
        const bool in_discards = (dtm >> tt) & 1ull;
        const bool in_missed = (qfl & PF_MISSED_DOUJUN) || ((qfl & PF_RIICHI_DECLARED) && (qfl & PF_MISSED_RIICHI));
        const bool furiten = (W & dtm) != 0ull || (qfl & (PF_MISSED_RIICHI | PF_MISSED_DOUJUN));
        const uint32_t riichi_m = rballot(r < 4 && (qfl & PF_RIICHI_DECLARED), rb) & 0xFu;
        todo = rballot(other && !in_discards && !in_missed && !furiten && ((W >> tt) & 1ull), rb) & 0xFu & ~riichi_m;
    }
    if (tsumo) todo = 1u << G->current_player;
    uint32_t ok_m = 0u, miss_m = 0u, han_m = 0u;
    while (__ballot(todo != 0u)) {
        const uint32_t on = todo != 0u ? 1u : 0u;
        const int i = on ? __ffs((int)todo) - 1 : 0;
        todo &= todo - 1u;
        const uint32_t fl_i = G->p[i].flags;
        uint32_t cf = ((fl_i & PF_RIICHI_DECLARED) ? CF_RIICHI : 0u) | ((fl_i & PF_DOUBLE_RIICHI) ? CF_DOUBLE_RIICHI : 0u) | ((fl_i & PF_IPPATSU) ? CF_IPPATSU : 0u);
        if (tsumo) {
            cf |= CF_TSUMO;
            if (G->drawable_count == 0 && !G->is_rinshan) cf |= CF_HAITEI;
            if (G->is_rinshan) cf |= CF_RINSHAN;
            if (G->is_first_turn && G->p[i].n_discards == 0) cf |= CF_FIRST_TURN;   // quirk Q5
        } else if (G->drawable_count == 0 && !G->is_rinshan) {
            cf |= CF_HOUTEI;
        }
        const uint32_t res = r4_yaku_check(on, (uint32_t)i, (uint32_t)tile, cf);
        if (on) {
            if (res & 1u) ok_m |= 1u << i;
            else if (res & 2u) miss_m |= 1u << i;
            if (res & 4u) han_m |= 1u << i;
        }
    }
    if (r == 0) sh.yk[row] = tsumo ? (((ok_m & han_m) != 0u) ? 1u : 0u) : (ok_m | (miss_m << 4));
    wave_sync();
}

// ---------------------------------------------------------------------------------------------------------------------------
// The device policy that PLAYS (rmj_step_greedy; the oracle's twin is orc_game_greedy_actions): for every seat that is to act,
// over its ordered legal list, the first entry of the best class -
//   Tsumo / Ron  >  Kita  >  Riichi  >  Ankan  >  Kakan  >  Daiminkan  >  [Pon > Chi when (key >> 24) < call_rate]  >
//   Discard  >  Pass  >  Kyushu kyuhai
// (Kita before Riichi: the 3P reference offers Kita in the riichi stage and can leave a seat that takes it without any legal
// action - quirk Q15 -; a seat that has set its Norths aside before it declares cannot get there)
// - and among the Discard entries the one that leaves the hand with the lowest shanten (shanten.rs:228-241 / :454-468 of the hand
// without the tile), ties broken by policy_tie(key, #ties) in list order; key = policy_key32(splitmix64(seed + global game), step, seat)
// like the RandomAgent's (rmj_common.hip.h).  What scripts/soak_greedy.py plays on the CPU, now resident: games reach tenpai, declare riichi and end
// with wins, so the rare transitions of the step (yaku checks, settlements, wait probes) are no longer rare.
// Lane = list entry while a seat's list is scanned (16 entries per pass), lane = discard candidate for the shanten of "hand minus
// my tile": the three suits a discard leaves alone are looked up once per row (lanes 0..3), their pair merges are computed entry
// per lane (lanes 0..9), and every candidate looks up its own suit and finishes with one triple (min,+) for the entry (pair, m).
__device__ __forceinline__ uint32_t r4_prio(uint32_t ty, bool call) {
    const uint32_t lo = call ? 0x90205678u : 0x90205CC8u, hi = 0xFFFF1A43u;   // nibble per action type, see the order above
    return __builtin_amdgcn_ubfe(ty < 8u ? lo : hi, (ty & 7u) * 4u, 4u);
}
// entry (pair = 1, mentsu = m) of merge(merge(nv, other), side) as ONE triple (min,+) over packed cost vectors (idx = p * 5 + k,
// 4 bits, 15 = infeasible): the same number as sh_merge + sh_merge_entry (every partial sum that reaches 15 stays >= 15)
__device__ __forceinline__ uint32_t sh_triple_1m(uint64_t nv, uint64_t other, uint64_t side, int m) {
    const uint32_t n0 = (uint32_t)nv & 0xFFFFFu, n1 = (uint32_t)(nv >> 20) & 0xFFFFFu;
    const uint32_t o0 = (uint32_t)other & 0xFFFFFu, o1 = (uint32_t)(other >> 20) & 0xFFFFFu;
    const uint32_t s0 = ((uint32_t)side & 0xFFFFFu) | 0xFFF00000u, s1 = ((uint32_t)(side >> 20) & 0xFFFFFu) | 0xFFF00000u;
    uint32_t S0[5], S1[5], N0[5], N1[5], O0[5], O1[5];
#pragma unroll
    for (int j = 0; j < 5; j++) {   // side entry (p, m - j); 15 when j > m
        const uint32_t off = m >= j ? 4u * (uint32_t)(m - j) : 20u;
        S0[j] = (s0 >> off) & 15u; S1[j] = (s1 >> off) & 15u;
        N0[j] = __builtin_amdgcn_ubfe(n0, 4u * j, 4u); N1[j] = __builtin_amdgcn_ubfe(n1, 4u * j, 4u);
        O0[j] = __builtin_amdgcn_ubfe(o0, 4u * j, 4u); O1[j] = __builtin_amdgcn_ubfe(o1, 4u * j, 4u);
    }
    uint32_t best = 15u;
#pragma unroll
    for (int k1 = 0; k1 < 5; k1++)
#pragma unroll
        for (int k2 = 0; k1 + k2 < 5; k2++) {
            const int j = k1 + k2;
            best = min(best, N0[k1] + O0[k2] + S1[j]);
            best = min(best, N1[k1] + O0[k2] + S0[j]);
            best = min(best, N0[k1] + O1[k2] + S0[j]);
        }
    return best;
}
__device__ __forceinline__ uint32_t row_min16u(uint32_t v) {  // minimum over a 16-lane row of values up to 2^31 - 1, in lane 15 of the row
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0x7FFFFFFF, (int)v, 0x111, 0xf, 0xf, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0x7FFFFFFF, (int)v, 0x112, 0xf, 0xf, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0x7FFFFFFF, (int)v, 0x114, 0xf, 0xf, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0x7FFFFFFF, (int)v, 0x118, 0xf, 0xf, false));
    return v;
}
// sh_vec for the words of real hands (at most 14 tiles, at most four of a kind): the digit clamps of sh_rank are not needed, five
// instructions per rank instead of nine; the final min keeps the table load in bounds whatever a poked state holds.
__device__ __forceinline__ uint64_t sh_vec_fast(uint32_t word, int q, const ShantenTables& T) {
    const uint32_t* R = q < 3 ? T.rank9 : T.rank7;
    uint32_t h = 0u, sum = 0u;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        if (i < 7 || q < 3) {
            const uint32_t c = __builtin_amdgcn_ubfe(word, 3u * i, 3u);
            h += R[(i * 15 + (sum > 14u ? 14u : sum)) * 5 + (c > 4u ? 4u : c)];
            sum += c;
        }
    }
    return q < 3 ? T.suit[h < SH_SUIT_ENTRIES ? h : SH_SUIT_ENTRIES - 1] : T.honor[h < SH_HONOR_ENTRIES ? h : SH_HONOR_ENTRIES - 1];
}
// entry (p, k) of the (min,+) merge of two packed cost vectors for a lane-local (p, k), like sh_merge_entry, with the second
// vector pre-rotated so that every partner entry sits at a compile-time nibble (entries that do not exist read 15): 4 instead of
// ~10 instructions per combination
__device__ __forceinline__ uint32_t sh_merge_entry_rot(uint64_t x, uint64_t y, int p, int k) {
    const uint32_t x0 = (uint32_t)x & 0xFFFFFu, x1 = (uint32_t)(x >> 20) & 0xFFFFFu;
    const uint32_t y0 = (uint32_t)y & 0xFFFFFu, y1 = (uint32_t)(y >> 20) & 0xFFFFFu;
    const uint32_t sft = 4u * (uint32_t)(4 - k), fill = (1u << sft) - 1u;
    const uint32_t ya = ((p ? y1 : y0) << sft) | fill;          // partner of x[0][k1]: y[p][k - k1] at nibble 4 - k1
    const uint32_t yb = p ? ((y0 << sft) | fill) : 0xFFFFFFFFu;  // partner of x[1][k1]: y[p - 1][k - k1]
    uint32_t best = 15u;
#pragma unroll
    for (int k1 = 0; k1 < 5; k1++) {
        best = min(best, __builtin_amdgcn_ubfe(x0, 4u * k1, 4u) + __builtin_amdgcn_ubfe(ya, 4u * (4 - k1), 4u));
        best = min(best, __builtin_amdgcn_ubfe(x1, 4u * k1, 4u) + __builtin_amdgcn_ubfe(yb, 4u * (4 - k1), 4u));
    }
    return best;
}
// Shanten (shanten.rs:228-241 / :454-468) of the row's hand h (hl tiles, row-uniform) WITHOUT one tile of type t34, per lane
// (`cand` lanes; 99 elsewhere): the three suits a discard leaves alone are looked up once per row (lanes 0..3), their pair merges
// are computed entry per lane (lanes 0..9), and every candidate looks up its own suit and finishes with one triple (min,+) for
// the entry (pair, m).  3P: the hands are compared after the relocation of 1m / 9m into empty honor slots; a candidate whose
// relocated hand differs from the row's in more than one word (only when all nine slots are taken) takes the plain per-lane lookup.
__device__ __forceinline__ int r4_shanten_minus(const R4& q, const ShantenTables& T, const PH& h, int hl, bool cand, int t34) {
    const int r = q.r, rb = q.rb;
    const int len3 = (hl - 1) / 3, m = len3 > 4 ? 4 : len3;
    const PH B = KSANMA ? sh_relocate_3p(h) : h;
    // the row's base: suit vectors by lanes 0..3, pair merges ab / cd entry per lane (0..9)
    const uint64_t vq = r < 4 ? sh_vec_fast(ph_get(B, r & 3), r & 3, T) : 0ull;
    const uint64_t v0 = rbc64(vq, rb), v1 = rbc64(vq, rb + 1), v2 = rbc64(vq, rb + 2), v3 = rbc64(vq, rb + 3);
    uint64_t ab, cd;
    {
        const int pp = r >= 5 ? 1 : 0, kk = r - 5 * pp;
        uint32_t ea = 0u, ec = 0u;
        if (r < 10) {
            ea = sh_merge_entry_rot(v0, v1, pp, kk);
            ec = sh_merge_entry_rot(v2, v3, pp, kk);
        }
        const uint32_t sh_lo = (uint32_t)(r & 7) * 4u;
        const uint32_t a_lo = row_or16(r < 8 ? ea << sh_lo : 0u), a_hi = row_or16((r >= 8 && r < 10) ? ea << sh_lo : 0u);
        const uint32_t c_lo = row_or16(r < 8 ? ec << sh_lo : 0u), c_hi = row_or16((r >= 8 && r < 10) ? ec << sh_lo : 0u);
        ab = (uint64_t)(uint32_t)rbc((int)a_lo, rb + 15) | ((uint64_t)(uint32_t)rbc((int)a_hi, rb + 15) << 32);
        cd = (uint64_t)(uint32_t)rbc((int)c_lo, rb + 15) | ((uint64_t)(uint32_t)rbc((int)c_hi, rb + 15) << 32);
    }
    int sres = 99;
    bool slow = false;
    PH x = h;
    if (cand) {
        ph_sub(x, t34);
        const PH X = KSANMA ? sh_relocate_3p(x) : x;
        int qd = t_suit(t34);
        if (KSANMA) {   // the relocation may move the change into the honor word (1m / 9m) or touch two words (all slots taken)
            const int nd4 = (X.a != B.a) + (X.b != B.b) + (X.c != B.c) + (X.d != B.d);
            qd = X.b != B.b ? 1 : (X.c != B.c ? 2 : (X.d != B.d ? 3 : 0));
            slow = nd4 != 1;
        }
        if (!slow) {
            const uint64_t nv = sh_vec_fast(ph_get(X, qd), qd, T);
            const uint64_t other = qd == 0 ? v1 : (qd == 1 ? v0 : (qd == 2 ? v3 : v2));
            sres = (int)sh_triple_1m(nv, other, qd < 2 ? cd : ab, m) - 1;
        }
    }
    if (KSANMA && __ballot(slow)) {
        if (slow) sres = sh_normal(sh_relocate_3p(x), len3, T);
    }
    if (cand && sres > 0 && len3 >= 4) {
        const int c7 = sh_chiitoi(x, KSANMA);
        sres = c7 < sres ? c7 : sres;
        if (sres > 0) {
            const int k13 = sh_kokushi(x);
            sres = k13 < sres ? k13 : sres;
        }
    }
    return sres;
}
// tenpai_after_discard (legal_actions.rs:113-140 / :77-100) for every slot of the row's acting seat (hl = 14 - 3 melds tiles,
// histogram `full`): bit j = "the hand without hand[j] has a wait".  The table shanten of the 13 remaining tiles sieves the
// slots (a hand with a wait has shanten 0); the survivors - one to three types - get the exact wait probe, once per type.
__device__ __forceinline__ uint32_t r4_tenpai_keep(const R4& q, const PState* P, const PH& full, int hl) {
    const int r = q.r, rb = q.rb;
    const ShantenTables T = sh_tables_of(*q.E);
    const int ty = r < hl ? (int)(P->hand[r] >> 2) : 0;
    const int sm = r4_shanten_minus(q, T, full, hl, r < hl, ty);
    uint32_t todo = rballot(r < hl && sm <= 0, rb);
    uint32_t out = 0u;
    while (__ballot(todo != 0u)) {
        if (todo) {
            const int j = __ffs((int)todo) - 1;
            const int tj = rbc(ty, rb + j);
            const uint32_t same = rballot(r < hl && ty == tj, rb);
            todo &= ~same;
            PH h13 = full;
            ph_sub(h13, tj);
            if (r4_waits_probe(h13.a, h13.b, h13.c, h13.d) != 0ull) out |= same;
        }
    }
    return out;
}
template <bool LOOP>
__device__ __forceinline__ uint64_t r4_list_entry(const uint64_t* src) {
    return LOOP ? __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *src;
}
// returns (lane = seat) the seat's action; rows that are not `on` get nothing
template <bool LOOP>
__device__ __forceinline__ uint64_t r4_policy_greedy(R4& q, bool on, const uint64_t* Lg, uint64_t gs, uint32_t call_rate, int& pol_seat, int& pol_sh) {
    GState* G = q.G;
    const int r = q.r, rb = q.rb;
    const ShantenTables T = sh_tables_of(*q.E);
    uint64_t mine = RMJ_NO_ACTION;
    pol_seat = -1;
    pol_sh = 99;
    uint32_t todo = on ? ((uint32_t)G->active_mask & 0xFu) : 0u;
    while (__ballot(todo != 0u)) {
        if (todo) {
            const int p = __ffs((int)todo) - 1;
            todo &= todo - 1u;
            const int n = G->nlegal[p];
            const uint32_t key = policy_key32(gs, G->step_count, (uint32_t)p);
            const bool call = policy_calls(key, call_rate);
            // ---- scan: best (class, index) over the list, 16 entries per pass
            uint32_t best = 0xFFFFu;
            uint64_t e0 = 0ull;      // lane's entry of the first pass (the Discard entries of a list sit in its first 15 slots)
            uint32_t disc = 0u;      // Discard entries of the first pass
            for (int base = 0; base < n; base += 16) {
                const bool in = base + r < n;
                const uint64_t e = in ? r4_list_entry<LOOP>(Lg + p * RMJ_MAX_LEGAL + base + r) : 0ull;
                const uint32_t ty = (uint32_t)e & 0xFFu;
                uint32_t sc = in ? r4_prio(ty > 15u ? 15u : ty, call) * 64u + (uint32_t)(base + r) : 0xFFFFu;
                sc = row_min16u(sc);
                sc = (uint32_t)rbc((int)sc, rb + 15);
                best = sc < best ? sc : best;
                if (base == 0) { e0 = e; disc = rballot(in && ty == RMJ_DISCARD, rb); }
            }
            int ci = (int)(best & 63u);
            const int nd = __popc(disc);
            if (n > 0 && (best >> 6) == 8u && nd >= 2) {
                // ---- shanten of the hand without each candidate
                const PState* P = &G->p[p];
                const int hl = P->hand_len;
                const PH h = r4_hist(q, P, -1);
                const bool cand = (disc >> r) & 1u;
                const int sres = r4_shanten_minus(q, T, h, hl, cand, (int)((e0 >> 8) & 0xFFull) >> 2);
                const uint32_t key_s = (uint32_t)(sres + 2);                      // (-1 .. 14) -> 1 .. 16, non-candidates 101
                const uint32_t smin = (uint32_t)rbc((int)row_min16u(cand ? key_s : 200u), rb + 15);
                const uint32_t tie = rballot(cand && key_s == smin, rb);
                const uint32_t kth = policy_tie(key, (uint32_t)__popc(tie));
                const uint32_t pick = rballot(((tie >> r) & 1u) && (uint32_t)__popc(tie & ((1u << r) - 1u)) == kth, rb);
                ci = __ffs((int)pick) - 1;
                pol_seat = p;                 // the shanten of the 13 tiles this discard leaves: the wait cache takes it (r4_resolve_discard)
                pol_sh = (int)smin - 2;
            }
            if (n > 0 && r == p) mine = r4_list_entry<LOOP>(Lg + p * RMJ_MAX_LEGAL + ci);
        }
    }
    return mine;
}

// One step of four consecutive games per wave (device policy only: rmj_step_random / rmj_bench_rollout); `load`: fetch the records from HBM first (the rollout loop keeps them in LDS)
// INLR (fused RandomAgent rollouts): a row whose discard drew claims answers them in the SAME call - the policy picks the seats'
// responses from the staged lists and the WaitResponse branch runs right behind the WaitAct branch - so that every call starts with
// (nearly) all four rows in WaitAct instead of 44 % of the waves paying for both branches with part of their rows idle in each.
// Rows then advance one or two game-steps per call: `left` = the steps the row's game still has to take in this rollout / ticket
// (row-uniform), the result = how many it took (0: none left).  `final_chunk`: the rollout ends when `left` runs out (the last
// step publishes masks and status, STEP_F_ALLROWS; every other step is quiet) - with INLR the caller passes `flags` without those bits.
template <bool LOOP, int POL, bool INLR = false, bool PASS2 = false>
__device__ __forceinline__ uint32_t step4_body(const Env* Ep, Quad4Shared& sh, uint64_t policy_seed, uint32_t flags, uint32_t g_base, uint32_t g_end,
                                               bool load, uint64_t gs_row, const uint64_t* __restrict__ actions = nullptr, uint32_t quad = 0xFFFFFFFFu,
                                               uint32_t left = 1u, bool final_chunk = true) {
    constexpr bool pass2 = PASS2;   // see q.live below
    static_assert(!INLR || LOOP, "inline responses: fused device-policy rollouts only");
    CEnv& E = *(CEnv*)Ep;
    // RICH tier 0 (the wait probe, the Riichi offer / declaration / riichi-stage list in row form): for policies that play - the
    // greedy instantiation and the per-step kernels an external policy drives.  The fused RandomAgent rollout keeps the lean
    // tier 0: those exits are 0.3 % of its game-steps, and the extra code costs it 6 % (1.61 -> 1.51 G env.step/s, measured).
    constexpr bool RICH = (POL == 1) || !LOOP;
#ifdef RMJ_TL4
    uint64_t tl_prev = __builtin_readcyclecounter();
#endif
    int lane_ = threadIdx.x & 63;
#if RMJ_INLINE_STEP
    if constexpr (INLR) asm volatile("" : "+v"(lane_));   // nothing derived from the lane id is hoisted out of the rollout loop and kept live across the step (step4_call_inl)
#endif
#if RMJ_INLINE_ENC
    if constexpr (LOOP && !INLR && !PASS2) asm volatile("" : "+v"(lane_));
#endif
    const int lane = lane_;
    const int row = lane >> 4, r = lane & 15, rb = lane & 48;
    const uint32_t rows_pw = r4_rows(flags);
    const uint32_t g0 = g_base + (quad == 0xFFFFFFFFu ? blockIdx.x : quad) * rows_pw;   // (k_step4_queue names the quad, the others own quad = block)
    const uint32_t g = g0 + (uint32_t)row;
    const uint32_t n_here = g0 >= g_end ? 0u : (g_end - g0 < rows_pw ? g_end - g0 : rows_pw);   // games of this wave
    if (load) {   // ---- records: every row fetches its own 640 B (40 chunks of 16 B, three per lane)
        if ((uint32_t)row < n_here) {
            if (LOOP) {
                // A fused rollout may take its quad over from a wave of another CU (tickets: k_step4_queue).  Everything tier 0 reads of a
                // game's global state - the record here, the wall tile and the list entry further down - goes past the vector L1
                // (agent-scope loads, served by the XCD's L2 that the previous holder's write-through stores reached), so a pick-up needs
                // no L1 invalidate: that invalidate wipes the cache for all 24 waves of the CU, and at one ticket per few wave-steps
                // it cost the 20-step window 8 % (profiles/r05_ticket_acquire_ab.txt).  The rare paths that read more with plain
                // loads (ol_step_full, the rich tier's settlement) invalidate when they are entered.
                uint64_t v[5];
#pragma unroll
                for (int k = 0; k < 5; k++) v[k] = __hip_atomic_load(reinterpret_cast<const uint64_t*>(E.core + g) + r + 16 * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int k = 0; k < 5; k++) reinterpret_cast<uint64_t*>(&sh.st[row])[r + 16 * k] = v[k];
            } else {
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const int off = r + 16 * k;
                    if (off < (int)(sizeof(GState) / 16)) reinterpret_cast<uint4*>(&sh.st[row])[off] = reinterpret_cast<const uint4*>(E.core + g)[off];
                }
            }
        }
        wave_sync();
    }
    R4M(40);
    R4T(0);
    {
    R4 q;
    q.G = &sh.st[row]; q.T = &sh.u.t; q.E = &E; q.lane = lane; q.r = r; q.rb = rb; q.row = row; q.g = g;
    // (round 6, one box, 65 536 games: with the tests the fused rollouts gain 1.5 % (2.11 -> 2.14 G); the kernels of one step per launch lose - the step at
    //  80 registers has no room for them: 0.97 -> 0.88 G - and keep the isolated-tile bounds alone)
    q.gf = LOOP || RMJ_GROUP_FILTER_ALL;
    // pass 2 (RMJ_ROW_ROUND_END): the rows whose round ended in pass 1 and has been dealt since (r4_round_end, run by the caller between the
    // passes - a call in here would cost every step ten more callee-saved registers): they skip policy and transitions, get their
    // first list and are published like any other row
    q.live = pass2 ? ((uint32_t)row < n_here && sh.rmode[row] != 0u) : ((uint32_t)row < n_here && (!INLR || left != 0u));
    GState* G = q.G;
    const uint64_t* Lg = E.legal + (size_t)g * 4 * RMJ_MAX_LEGAL;
    q.bail = false; q.cont = 0; q.rend = 0u; q.evn = 0; q.dirty = 0xFu;
#ifdef RMJ_CENSUS
    q.why = 0;
#endif
    // the next live-wall draw (valid while nothing moves live_end: kans bail)
    int pf = 0;
    if (q.live) {
        const int le = G->live_end;
        const int wi = le > 0 ? le - 1 : 0;
        const uint8_t* Wg = E.wall + (size_t)g * RMJ_WALL_STRIDE;
        if (LOOP) {
            const uint32_t w = __hip_atomic_load(reinterpret_cast<const uint32_t*>(Wg) + (wi >> 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            pf = (int)((w >> (8 * (wi & 3))) & 0xFFu);
        } else {
            pf = Wg[wi];
        }
    }
    // ---- policy (lane = seat): RandomAgent keyed per (game, step, seat), see k_step; POL = 1: the greedy policy (r4_policy_greedy)
    uint64_t mine = RMJ_NO_ACTION;
    q.pause_ok = RMJ_ROW_ROUND_END && !pass2;
    q.yk = 0u; q.yk_mode = 0u;
    if (pass2) {
        if (q.live) {
            q.rend = R4_RE_PUB;
            const uint32_t md = sh.rmode[row];
            if (md == R4_RE_YAKU_CLAIMS || md == R4_RE_YAKU_TSUMO) { q.yk_mode = md; q.yk = sh.yk[row]; }
        }
    } else if (q.live && G->is_done) {   // finished game: restart (auto-reset: in row form) or nothing to do (full path)
        if (RMJ_ROW_ROUND_END && (flags & STEP_F_AUTORESET)) q.rend = R4_RE_RESTART; else R4BAIL(q, 18);
    }
    int pol_seat = -1, pol_sh = 99;   // greedy policy: shanten of the hand the chosen discard of seat pol_seat leaves
    if (POL == 1) {
        const uint64_t gs = LOOP ? gs_row : sm64(policy_seed + E.game_offset + (uint64_t)g);
        mine = r4_policy_greedy<LOOP>(q, q.live && !q.bail && q.rend == 0u, Lg, gs, (flags >> 8) & 0xFFu, pol_seat, pol_sh);
    } else if (q.live && !q.bail && q.rend == 0u) {
        if (flags & STEP_F_RANDOM) {
            if (r < 4) {
                const uint32_t n = G->nlegal[r];
                if (((G->active_mask >> r) & 1u) && n != 0u) {
                    const uint64_t gs = LOOP ? gs_row : sm64(policy_seed + E.game_offset + (uint64_t)g);   // (the loop hashes the game once)
                    const uint32_t ch = policy_pick(policy_key32(gs, G->step_count, (uint32_t)r), n > 64u ? 64u : n);
                    const uint64_t* src = Lg + r * RMJ_MAX_LEGAL + ch;
                    mine = LOOP ? __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *src;
                }
            }
        } else if (!LOOP) {
            // ---- the caller's actions (lane = seat): packed actions, validated against the stored lists like GameState::step
            // does (state/mod.rs:339-402), or action ids mapped to the first legal action with that id (Observation.find_action,
            // observation/python.rs:119-122).  Anything the lists do not confirm goes to the full path (illegal action penalty).
            const bool by_id = (flags & STEP_F_IDS) != 0u;
            int want = -1;
            uint64_t given = RMJ_NO_ACTION;
            bool has = false;
            const uint32_t am = G->active_mask;
            if (r < 4) {
                if (by_id) {   // ids of seats that are not to act are ignored (k_step does the same)
                    want = reinterpret_cast<const int32_t*>(actions)[(size_t)g * 4 + r];
                    has = want >= 0 && ((am >> r) & 1u) && G->nlegal[r] != 0;
                } else {
                    const uint64_t a = actions[(size_t)g * 4 + r];
                    has = (a & 0xFFull) != 0xFFull;
                    given = has ? a_canon(a) : RMJ_NO_ACTION;
                    mine = given;   // (a bailed row hands the caller's actions to the full path as they are)
                }
            }
            if (!by_id && rballot(r < 4 && has && (!((am >> r) & 1u) || G->nlegal[r & 3] == 0), rb)) {
                R4BAIL(q, 27);       // a seat that is not to act sent something: the reference's validation decides
            } else {
                uint32_t todo = rballot(r < 4 && has, rb) & 0xFu;
                while (__ballot(todo != 0u)) {
                    if (todo) {
                        const int p = __ffs((int)todo) - 1;
                        todo &= todo - 1u;
                        const int n = G->nlegal[p];
                        const int want_p = rbc(want, rb + p);
                        const uint64_t given_p = rbc64(given, rb + p);
                        bool found = false;
                        uint64_t chosen = 0ull;
                        for (int base = 0; base < n; base += 16) {   // 16 stored entries per pass
                            const bool in = base + r < n;
                            const uint64_t e = in ? Lg[p * RMJ_MAX_LEGAL + base + r] : 0ull;
                            const bool hit = in && (by_id ? (KSANMA ? a_encode_3p(e) : a_encode(e)) == want_p : a_match(e, given_p));
                            const uint32_t hb = rballot(hit, rb);
                            if (hb && !found) {
                                found = true;
                                chosen = rbc64(e, rb + __ffs((int)hb) - 1);
                            }
                        }
                        if (by_id && r == p) mine = found ? chosen : mk_action(0x7F, RMJ_TILE_NONE, 0);   // no legal action with that id
                        if (!found) { R4BAIL(q, 28); todo = 0u; }   // illegal action: penalty in the full path
                    }
                }
            }
        }
    }
    R4M(41);
    R4T(1);
    int nl_mine = 0;          // lane = seat: length of the seat's list produced by this step
    uint64_t w_mine = 0ull;   // lane = seat: waits published for the seat
    const bool t0 = q.live && !q.bail;
    bool second = false;      // INLR: this row answers the claims on its own discard in this call (its second game-step)
    bool noop = false;        // WaitAct and nothing from the seat that is to act: the step changes nothing (state/mod.rs:404-408)
    if (t0 && q.rend == 0u) {
        if (r == 0) G->step_count += 1;
        const int phase = G->phase;
        if (phase == RMJ_WAIT_ACT) {
            const int pid = G->current_player;
            const uint32_t act = (uint32_t)rbc((int)(uint32_t)mine, rb + pid);   // type and tile live in the low dword
            const uint32_t ty = act & 0xFFu;
            PState* P = &G->p[pid];
            if (RICH && ty == RMJ_RIICHI && ((act >> 8) & 0xFFu) == RMJ_TILE_NONE && P->score >= 1000 &&
                (KSANMA ? G->drawable_count > 0 : G->drawable_count >= 4) && !(P->flags & (PF_RIICHI_DECLARED | PF_RIICHI_STAGE))) {
                // ---- Riichi declared (state/mod.rs:440-457): the seat enters the riichi stage and stays to act; its next list
                // (below) holds the discards that keep the hand tenpai
                q.dirty = 1u << pid;
                if (r == 0) P->flags |= PF_RIICHI_STAGE;
                wave_sync();
                r4_emit_simple(q, RMJ_EV_REACH, (uint32_t)pid, 0);
            } else if (act == 0xFFFFFFFFu) {
                // no action for the current player (a policy that skipped it; 3P: a seat the reference leaves without any legal action
                // after a Kita in its riichi stage, quirk Q15 - such a game stays like this for good, one bail per step before round 4)
                noop = true;
            } else if (RICH && RMJ_ROW_SETTLE && ty == RMJ_TSUMO && q.pause_ok) {
                // (with or without the tile: a caller's Tsumo matched the list's entry by type)
                q.rend = R4_RE_WIN_TSUMO;   // the settlement, the next round or the end of the game between the passes (r4_round_end)
            } else if (((act >> 8) & 0xFFu) == RMJ_TILE_NONE) {
                R4BAIL(q, 19);
            } else if (ty == RMJ_DISCARD) {
                q.dirty = 1u << pid;
                const int tile = (int)((act >> 8) & 0xFFu);
                const int hl = P->hand_len;
                const int t = r < hl ? (int)P->hand[r] : 0xFFFF;
                const int drawn = G->drawn_tile;
                const bool tsumogiri = drawn != 0xFF && drawn == tile;
                const uint32_t fm = rballot(r < hl && t == tile, rb);
                const int idx = fm ? __ffs((int)fm) - 1 : -1;
                // hand.remove(idx); hand.sort(): the hand is 13 sorted tiles + the drawn one (else: poked state -> bail)
                const int nxt = __builtin_amdgcn_update_dpp(0xFFFF, t, 0x101 /* row_shl:1 */, 0xf, 0xf, false);
                const uint32_t uns = rballot(r < hl - 2 && t > nxt, rb);
                // 3P: right after a Kita the tile drawn before it sits behind the sorted run, in front of the replacement draw
                const bool two_loose = (KSANMA || RICH) && hl >= 3 && uns == (1u << (hl - 3));   // (RICH: the tile drawn before a kan stays in front of the replacement draw)
                if (idx < 0 || (uns && !two_loose && !RICH)) {
                    R4BAIL(q, 20);
                } else {
                    int np;
                    if (uns && !two_loose) {
                        // any other order (rich tier: several replacement draws in a row - Kita, kan - leave more than two loose tiles;
                        // a poked hand): the new slot of a tile = the number of remaining tiles before it in (id, slot) order
                        int cnt = 0;
#pragma unroll 1
                        for (int k = 0; k < hl; k++) {
                            const int tk = rbc(t, rb + k);
                            cnt += (k != idx && (tk < t || (tk == t && k < r))) ? 1 : 0;
                        }
                        np = cnt;
                    } else if (!uns) {
                        const int d = rbc(t, rb + hl - 1);
                        const int before_d = __popc(rballot(r < hl - 1 && r != idx && t <= d, rb));
                        np = r - (idx < r ? 1 : 0) + ((idx != hl - 1 && d < t) ? 1 : 0);
                        if (r == hl - 1) np = before_d;
                    } else {
                        // sorted run of m = hl - 2 tiles + two loose ones (A, B): the new slot of a tile = the number of remaining
                        // tiles with a smaller id (ids are unique)
                        const int m = hl - 2;
                        const int A = rbc(t, rb + m), B = rbc(t, rb + m + 1);
                        const bool remA = idx != m, remB = idx != m + 1;
                        const bool in_run = r < m && r != idx;
                        const int lessA = __popc(rballot(in_run && t < A, rb)), lessB = __popc(rballot(in_run && t < B, rb));
                        np = r - (idx < r ? 1 : 0) + ((remA && A < t) ? 1 : 0) + ((remB && B < t) ? 1 : 0);
                        if (r == m) np = lessA + ((remB && B < A) ? 1 : 0);
                        if (r == m + 1) np = lessB + ((remA && A < B) ? 1 : 0);
                    }
                    wave_sync();
                    if (r < hl && r != idx) P->hand[np] = (uint8_t)t;
                    if (r == 0) P->hand_len = (uint8_t)(hl - 1);
                    wave_sync();
                    R4M(42);
                    r4_resolve_discard<RICH, LOOP>(q, pid, tile, tsumogiri, pf, nl_mine, w_mine, (POL == 1 && pol_seat == pid) ? pol_sh : 99);
                    R4M(56);
                }
            } else if (KSANMA && ty == RMJ_KITA) {
                // ---- handle_kita (state_3p/sanma.rs:9-144): a common action of 3P games; a seat that could rob the tile bails
                q.dirty = 0xFu;
                const int tile = (int)((act >> 8) & 0xFFu);
                const int hl = P->hand_len;
                const int t = r < hl ? (int)P->hand[r] : 0xFFFF;
                const uint32_t fm = rballot(r < hl && t == tile, rb);
                if (G->pending_kan_dora > 0 || !fm || (tile >> 2) != 30 || G->drawable_count == 0) {
                    R4BAIL(q, 21);
                } else {
                    const int idx = __ffs((int)fm) - 1;
                    wave_sync();
                    if (r > idx && r < hl) P->hand[r - 1] = (uint8_t)t;     // hand.remove(idx), order kept
                    if (r == 0) {
                        P->hand_len = (uint8_t)(hl - 1);
                        P->flags &= ~PF_WAITS_VALID;
                        if (P->n_kita < 4) P->kita[P->n_kita++] = (uint8_t)tile;
                        G->is_first_turn = 0;
                        G->ron_offer_mask = 0;
                    }
                    wave_sync();
                    r4_emit_simple(q, RMJ_EV_KITA, (uint32_t)pid, (uint32_t)tile);
                    // the other seats: refill stale wait caches, then "waits on North and not furiten" needs the yaku check
                    {
                        const PState& S0 = G->p[r & 3];
                        uint32_t need_m = rballot(r < KNP && r != pid && (S0.hand_len + 3 * S0.n_melds == 13) && !(S0.flags & PF_WAITS_VALID), rb) & 0xFu;
                        while (__ballot(need_m != 0u)) {
                            if (need_m) {
                                const int i = __ffs((int)need_m) - 1;
                                need_m &= need_m - 1u;
                                PState* Q = &G->p[i];
                                r4_fill_waits13<RICH>(q, Q, Q->hand_len);
                                if (q.bail) need_m = 0u;
                            }
                        }
                    }
                    if (!q.bail) {
                        const PState& S4 = G->p[r & 3];
                        const bool other = r < KNP && r != pid;
                        const bool holds13 = other && (S4.hand_len + 3 * S4.n_melds == 13);
                        const uint64_t W = holds13 ? S4.waits13 : 0ull;
                        const bool furiten = (W & S4.discard_type_mask) != 0ull || (S4.flags & (PF_MISSED_RIICHI | PF_MISSED_DOUJUN));
                        if (rballot(other && !furiten && ((W >> 30) & 1ull), rb)) R4BAIL(q, 22);
                    }
                    if (!q.bail) {
                        // resolve_kita_rinshan (state_3p/sanma.rs:171-204): replacement draw from the dead wall, no new dora
                        const int rc = G->rinshan_count;
                        const uint8_t* Wg = E.wall + (size_t)g * RMJ_WALL_STRIDE;
                        int rt;
                        if (LOOP) {
                            const uint32_t w = __hip_atomic_load(reinterpret_cast<const uint32_t*>(Wg) + (rc >> 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            rt = (int)((w >> (8 * (rc & 3))) & 0xFFu);
                        } else {
                            rt = Wg[rc];
                        }
                        if (r < 4) G->p[r].flags &= ~PF_IPPATSU;
                        wave_sync();
                        if (r == 0) {
                            G->rinshan_count = (uint8_t)(rc + 1);
                            G->drawable_count -= 1;
                            const int h2 = P->hand_len;
                            if (h2 < 14) { P->hand[h2] = (uint8_t)rt; P->hand_len = (uint8_t)(h2 + 1); }
                            G->drawn_tile = (uint8_t)rt;
                            G->is_rinshan = 1;
                            G->phase = RMJ_WAIT_ACT;
                            G->active_mask = (uint8_t)(1u << pid);
                        }
                        wave_sync();
                        r4_emit_simple(q, RMJ_EV_TSUMO, (uint32_t)pid, (uint32_t)rt);
                    }
                }
            } else if (RICH && (ty == RMJ_ANKAN || ty == RMJ_KAKAN)) {
                // ---- Ankan / Kakan (state/mod.rs:472-683, _resolve_kan :1415-1547) when nobody can rob the tile; everything is
                // done on the LDS copy, so any complication found on the way (a seat that could win on the tile, a full event
                // stage) simply bails: the full path starts over from the untouched record
                q.dirty = 0xFu;
                const uint64_t full_act = rbc64(mine, rb + pid);
                const int hl = P->hand_len;
                const int t = r < hl ? (int)P->hand[r] : 0xFFFF;
                const int tile = (int)((act >> 8) & 0xFFu);
                const int tt = tile >> 2;
                const bool ankan = ty == RMJ_ANKAN;
                // the tiles that leave the hand: the four of the Ankan, the one of the Kakan
                const uint32_t c0 = a_c(full_act, 0), c1 = a_c(full_act, 1), c2 = a_c(full_act, 2), c3 = a_c(full_act, 3);
                const bool gone = r < hl && (ankan ? ((uint32_t)t == c0 || (uint32_t)t == c1 || (uint32_t)t == c2 || (uint32_t)t == c3) : t == tile);
                const uint32_t rm = rballot(gone, rb);
                const uint32_t pon_hit = rballot(r < (int)P->n_melds && P->meld_type[r & 3] == RMJ_MELD_PON && (P->meld_tiles[r & 3][0] >> 2) == tt, rb);
                if (a_n(full_act) != (ankan ? 4u : 3u) || __popc(rm) != (ankan ? 4 : 1) || (!ankan && !pon_hit) || P->n_melds >= (ankan ? 4 : 5) ||
                    G->drawable_count == 0) {
                    R4BAIL(q, 23);
                } else {
                    // who could rob it: a Kakan by any seat that waits on the tile and is not furiten (chankan), an Ankan only under the
                    // kokushi rule - such a seat needs the evaluator.  Stale wait caches are refilled first.
                    const bool rob_rule = !ankan || (q.E->rule_bits & RMJ_RULE_RON_ON_ANKAN_KOKUSHI) != 0;
                    if (rob_rule) {
                        const PState& S0 = G->p[r & 3];
                        uint32_t need_m = rballot(r < KNP && r != pid && (S0.hand_len + 3 * S0.n_melds == 13) && !(S0.flags & PF_WAITS_VALID), rb) & 0xFu;
                        while (__ballot(need_m != 0u)) {
                            if (need_m) {
                                const int i = __ffs((int)need_m) - 1;
                                need_m &= need_m - 1u;
                                PState* Q = &G->p[i];
                                r4_fill_waits13<RICH>(q, Q, Q->hand_len);
                                if (q.bail) need_m = 0u;
                            }
                        }
                        if (!q.bail) {
                            const PState& S4 = G->p[r & 3];
                            const bool other = r < KNP && r != pid && (S4.hand_len + 3 * S4.n_melds == 13);
                            const uint64_t W = other ? S4.waits13 : 0ull;
                            const bool blocked = ankan ? ((S4.discard_type_mask >> tt) & 1ull) != 0ull
                                                       : ((W & S4.discard_type_mask) != 0ull || (S4.flags & (PF_MISSED_RIICHI | PF_MISSED_DOUJUN)) != 0u);
                            if (rballot(other && !blocked && ((W >> tt) & 1ull), rb)) R4BAIL(q, 30);
                        }
                    }
                    if (!q.bail) {
                        wave_sync();
                        if (r < hl && !gone) P->hand[r - __popc(rm & ((1u << r) - 1u))] = (uint8_t)t;
                        uint32_t kc0 = c0, kc1 = c1, kc2 = c2;   // consume tiles of the event
                        if (r == 0) {
                            P->hand_len = (uint8_t)(hl - __popc(rm));
                            P->flags &= ~PF_WAITS_VALID;
                            if (rob_rule) G->ron_offer_mask = 0;   // (the reference clears it where it looks for robbers)
                            if (ankan) {
                                push_meld(*P, RMJ_MELD_ANKAN, c0, c1, c2, c3, 4, 0xFF, 0xFF);
                            } else {
                                const int m = __ffs((int)pon_hit) - 1;
                                uint32_t v[4] = {P->meld_tiles[m][0], P->meld_tiles[m][1], P->meld_tiles[m][2], (uint32_t)tile};
#pragma unroll
                                for (int a = 0; a < 3; a++)
#pragma unroll
                                    for (int b = 0; b < 3; b++)
                                        if (v[b] > v[b + 1]) { const uint32_t x = v[b]; v[b] = v[b + 1]; v[b + 1] = x; }
                                for (int a = 0; a < 4; a++) P->meld_tiles[m][a] = (uint8_t)v[a];
                                P->meld_type[m] = RMJ_MELD_KAKAN;
                            }
                        }
                        wave_sync();
                        if (ankan) {
                            r4_kan_draw<LOOP>(q, pid, P, true, (uint32_t)RMJ_EV_ANKAN | ((uint32_t)pid << 8) | ((uint32_t)tile << 24),
                                              kc0 | (kc1 << 8) | (kc2 << 16) | (c3 << 24), 4u);
                        } else {
                            // the Kakan announces itself before anything else, then the indicators of earlier open kans (state/mod.rs:568-587)
                            r4_emit(q, (uint32_t)RMJ_EV_KAKAN | ((uint32_t)pid << 8) | ((uint32_t)tile << 24), kc0 | (kc1 << 8) | (kc2 << 16), (3u << 4) & 0xFFu);
                            if (__ballot(G->pending_kan_dora > 0)) r4_flush_kan_dora<LOOP>(q, 0);
                            if (!q.bail) r4_kan_draw<LOOP>(q, pid, P, false, 0u, 0u, 0u);
                        }
                    }
                }
            } else {
                R4BAIL(q, 23);   // Riichi (with a tile), Kyushu; lean tier: kans, Tsumo
            }
        }
        uint64_t mine_r = mine;   // the seats' responses (lane = seat)
        if (INLR) {
            // the discard drew claims and the game has a step left: what the NEXT call would do first - the claim-offered seats get
            // their observation (event cursors, state/mod.rs:211-218), the policy keys (game, step, seat) pick from the staged lists
            // (the lean tier never leaves a Ron offer or a pending kan / kita behind: those discards bail)
            second = phase == RMJ_WAIT_ACT && !q.bail && G->phase == RMJ_WAIT_RESPONSE && left >= 2u && (!RICH || G->pending_kan_pid == 0xFF);
            if (POL == 1 ? __ballot(second) != 0ull : second) {
                const uint32_t am2 = second ? (uint32_t)G->active_mask & 0xFu : 0u;
                const uint32_t sc = G->step_count;
                uint64_t pick = RMJ_NO_ACTION;
                if (POL == 1) {
                    // the greedy policy's answer (r4_policy_greedy without its discard part): the best class of the seat's list
                    const uint32_t call_rate = (flags >> 8) & 0xFFu;
                    uint32_t todo = am2;
                    while (__ballot(todo != 0u)) {
                        if (todo) {
                            const int p = __ffs((int)todo) - 1;
                            todo &= todo - 1u;
                            const int n = rbc(nl_mine, rb + p);
                            const bool call = policy_calls(policy_key32(gs_row, sc, (uint32_t)p), call_rate);
                            const bool in = r < n;
                            const uint64_t e = in ? q.T->lst[row][p][r] & 0x00FFFFFFFFFFFFFFull : 0ull;
                            const uint32_t ty = (uint32_t)e & 0xFFu;
                            uint32_t scv = in ? r4_prio(ty > 15u ? 15u : ty, call) * 64u + (uint32_t)r : 0xFFFFu;
                            scv = (uint32_t)rbc((int)row_min16u(scv), rb + 15);
                            const uint64_t chosen = rbc64(e, rb + (int)(scv & 15u));
                            if (n > 0 && r == p) pick = chosen;
                        }
                    }
                } else if (r < 4 && ((am2 >> r) & 1u) && nl_mine > 0) {
                    const uint32_t ch = policy_pick(policy_key32(gs_row, sc, (uint32_t)r), nl_mine > 64 ? 64u : (uint32_t)nl_mine);
                    pick = q.T->lst[row][r][ch] & 0x00FFFFFFFFFFFFFFull;
                }
                // a Ron among the answers is a settlement (full path): the row stops after its discard, the next call takes it from there
                if (RICH && rballot(r < 4 && pick != RMJ_NO_ACTION && a_type(pick) == RMJ_RON, rb)) second = false;
                if (second) {
                    mine_r = pick;
                    if (r < 4 && ((am2 >> r) & 1u)) {
                        G->obs_from[r] = G->obs_upto[r];
                        G->obs_upto[r] = G->ev_count;
                    }
                    wave_sync();
                    if (r == 0) G->step_count = sc + 1u;
                }
            }
        }
        if (phase != RMJ_WAIT_ACT || second) {
            // ---- WaitResponse (state/mod.rs:900-1314), lane = seat
            q.dirty = 0xFu;
            const bool has = r < 4 && mine_r != RMJ_NO_ACTION;
            const uint32_t my_ty = a_type(mine_r);
            const uint32_t act_m = G->active_mask;
            const bool is_act = has && ((act_m >> r) & 1u);
            const uint32_t roned = rballot(has && my_ty == RMJ_RON, rb) & 0xFu;
            const uint32_t offer = G->ron_offer_mask;
            if (roned & act_m) {                                  // Ron settlement
                if (RICH && RMJ_ROW_SETTLE && q.pause_ok && !(!KSANMA && __popc(roned & act_m) >= 3 && (E.rule_bits & RMJ_RULE_SANCHAHO_DRAW))) {
                    if (r < 4 && ((offer & ~roned) >> r) & 1u) {  // a Ron offer that was not taken (state/mod.rs:905-915 runs before the settlement)
                        uint32_t fl = G->p[r].flags | PF_MISSED_DOUJUN;
                        if (fl & PF_RIICHI_DECLARED) fl |= PF_MISSED_RIICHI;
                        G->p[r].flags = (uint8_t)fl;
                    }
                    if (r == 0) sh.yk[row] = roned & act_m;
                    q.rend = R4_RE_WIN_RON;                       // between the passes (r4_round_end)
                } else {
                    R4BAIL(q, 24);                                // (three Rons under the sanchaho rule: an abortive draw)
                }
            } else {
                if (r < 4 && ((offer & ~roned) >> r) & 1u) {      // a Ron offer that was not taken
                    uint32_t fl = G->p[r].flags | PF_MISSED_DOUJUN;
                    if (fl & PF_RIICHI_DECLARED) fl |= PF_MISSED_RIICHI;
                    G->p[r].flags = (uint8_t)fl;
                }
                const uint32_t pon_m = rballot(is_act && (my_ty == RMJ_PON || my_ty == RMJ_DAIMINKAN), rb) & 0xFu;
                const uint32_t chi_m = KSANMA ? 0u : (rballot(is_act && my_ty == RMJ_CHI, rb) & 0xFu);
                const int claimer = pon_m ? __ffs((int)pon_m) - 1 : (chi_m ? __ffs((int)chi_m) - 1 : -1);
                wave_sync();
                if (claimer >= 0) {
                    const uint64_t claim = rbc64(mine_r, rb + claimer);
                    const uint32_t ty = a_type(claim);
                    if (ty == RMJ_DAIMINKAN && !RICH) {
                        R4BAIL(q, 25);
                    } else if (ty == RMJ_DAIMINKAN) {
                        // ---- Daiminkan (state/mod.rs:1143-1160 + _resolve_kan): the call itself like Pon, then the replacement draw
                        PState* C = &G->p[claimer];
                        r4_accept_riichi(q);
                        const int ldp = G->last_discard_pid, tile = G->last_discard_tile;
                        if (r < 4) {
                            uint32_t fl = G->p[r].flags & ~(uint32_t)PF_IPPATSU;
                            if (r == claimer) fl &= ~(uint32_t)(PF_MISSED_DOUJUN | PF_WAITS_VALID);
                            if (r == ldp) fl &= ~(uint32_t)PF_NAGASHI;
                            G->p[r].flags = (uint8_t)fl;
                        }
                        const int hl = C->hand_len;
                        const int hc = r < hl ? (int)C->hand[r] : 0xFFFF;
                        const uint32_t c0 = a_c(claim, 0), c1 = a_c(claim, 1), c2 = a_c(claim, 2);
                        const bool gone = r < hl && ((uint32_t)hc == c0 || (uint32_t)hc == c1 || (uint32_t)hc == c2);
                        const uint32_t rm = rballot(gone, rb);
                        if (__popc(rm) != 3 || a_n(claim) != 3u || C->n_melds >= 4) {
                            R4BAIL(q, 25);
                        } else {
                            wave_sync();
                            if (r < hl && !gone) C->hand[r - __popc(rm & ((1u << r) - 1u))] = (uint8_t)hc;
                            if (r == 0) {
                                C->hand_len = (uint8_t)(hl - 3);
                                G->is_rinshan = 0;
                                G->current_player = (uint8_t)claimer;
                                G->active_mask = (uint8_t)(1u << claimer);
                                C->n_forbidden = 0;
                                push_meld(*C, RMJ_MELD_DAIMINKAN, c0, c1, c2, (uint32_t)tile, 4, ldp, tile);
                                int nd = 0, nw = 0;   // pao_check (state/mod.rs:1443-1472)
                                for (int m = 0; m < C->n_melds; m++) {
                                    const int tm = C->meld_tiles[m][0] >> 2;
                                    if (C->meld_type[m] != RMJ_MELD_CHI) { nd += (tm >= 31 && tm <= 33); nw += (tm >= 27 && tm <= 30); }
                                }
                                const int tv = tile >> 2;
                                if (tv >= 31 && tv <= 33) { if (nd == 3) C->pao37 = (uint8_t)ldp; }
                                else if (tv >= 27 && tv <= 30) { if (nw == 4) C->pao50 = (uint8_t)ldp; }
                            }
                            wave_sync();
                            r4_kan_draw<LOOP>(q, claimer, C, false, (uint32_t)RMJ_EV_DAIMINKAN | ((uint32_t)claimer << 8) | ((uint32_t)ldp << 16) | ((uint32_t)tile << 24),
                                              c0 | (c1 << 8) | (c2 << 16), 3u);
                        }
                    } else {
                        PState* C = &G->p[claimer];
                        r4_accept_riichi(q);
                        const int ldp = G->last_discard_pid, tile = G->last_discard_tile;
                        if (r < 4) {
                            uint32_t fl = G->p[r].flags & ~(uint32_t)PF_IPPATSU;
                            if (r == claimer) fl &= ~(uint32_t)PF_MISSED_DOUJUN;
                            if (r == ldp) fl &= ~(uint32_t)PF_NAGASHI;
                            if (r == claimer) fl &= ~(uint32_t)PF_WAITS_VALID;   // hand_remove_tiles
                            G->p[r].flags = (uint8_t)fl;
                        }
                        // hand_remove_tiles: both consumed ids leave the hand, order preserved
                        const int hl = C->hand_len;
                        const int hc = r < hl ? (int)C->hand[r] : 0xFFFF;
                        const uint32_t c0 = a_c(claim, 0), c1 = a_c(claim, 1);
                        const uint32_t f0 = rballot(r < hl && (uint32_t)hc == c0, rb), f1 = rballot(r < hl && (uint32_t)hc == c1, rb);
                        const uint32_t rm = (f0 ? (f0 & (0u - f0)) : 0u) | (f1 ? (f1 & (0u - f1)) : 0u);
                        wave_sync();
                        if (r < hl && !((rm >> r) & 1u)) C->hand[r - __popc(rm & ((1u << r) - 1u))] = (uint8_t)hc;
                        if (r == 0) {
                            C->hand_len = (uint8_t)(hl - __popc(rm));
                            G->is_rinshan = 0;
                            G->is_first_turn = 0;
                            push_meld(*C, ty == RMJ_PON ? RMJ_MELD_PON : RMJ_MELD_CHI, c0, c1, (uint32_t)tile, 0, 3, ldp, tile);
                        }
                        wave_sync();
                        {   // emit_meld
                            const uint32_t cons = c0 | (c1 << 8);
                            r4_emit(q, (uint32_t)(ty == RMJ_PON ? RMJ_EV_PON : RMJ_EV_CHI) | ((uint32_t)claimer << 8) | ((uint32_t)ldp << 16) |
                                           ((uint32_t)tile << 24), cons, (2u << 4) & 0xFFu);
                        }
                        if (r == 0) {
                            if (ty == RMJ_PON) {   // pao_check (state/mod.rs:1228-1259)
                                int nd = 0, nw = 0;
                                for (int m = 0; m < C->n_melds; m++) {
                                    const int tm = C->meld_tiles[m][0] >> 2;
                                    if (C->meld_type[m] != RMJ_MELD_CHI) { nd += (tm >= 31 && tm <= 33); nw += (tm >= 27 && tm <= 30); }
                                }
                                const int tv = tile >> 2;
                                if (tv >= 31 && tv <= 33) { if (nd == 3) C->pao37 = (uint8_t)ldp; }
                                else if (tv >= 27 && tv <= 30) { if (nw == 4) C->pao50 = (uint8_t)ldp; }
                            }
                            G->current_player = (uint8_t)claimer;
                            G->phase = RMJ_WAIT_ACT;
                            G->active_mask = (uint8_t)(1u << claimer);
                            C->forbidden[0] = (uint8_t)tile;
                            C->n_forbidden = 1;
                            if (ty != RMJ_PON) {
                                const int t34 = tile >> 2;
                                const int x = (int)c0 >> 2, y = (int)c1 >> 2;
                                const int lo = min(x, y), hi = max(x, y);
                                if (lo == t34 + 1 && hi == t34 + 2) {
                                    if (t34 % 9 <= 5) { C->forbidden[1] = (uint8_t)((t34 + 3) * 4); C->n_forbidden = 2; }
                                } else if (t34 >= 2 && hi == t34 - 1 && lo == t34 - 2 && t34 % 9 >= 3) {
                                    C->forbidden[1] = (uint8_t)((t34 - 3) * 4);
                                    C->n_forbidden = 2;
                                }
                            }
                            G->needs_tsumo = 0;
                            G->drawn_tile = 0xFF;
                        }
                        wave_sync();
                    }
                } else {
                    if (G->pending_kan_pid != 0xFF) {
                        R4BAIL(q, 26);     // a chankan / kita offer was passed: the kan resolves in the full path
                    } else {
                        if (r == 0) {
                            G->active_mask = 0;
                            G->ron_offer_mask = 0;
                        }
                        if (r < 4) G->stale_n[r] = 0;   // current_claims.clear() (state/mod.rs:1299)
                        wave_sync();
                        r4_accept_riichi(q);
                        const uint32_t tc = G->turn_count + 1u;
                        const int np_ = (G->current_player + 1) % KNP;
                        if (r == 0) {
                            G->turn_count = tc;
                            G->current_player = (uint8_t)np_;
                        }
                        wave_sync();
                        r4_deal_next(q, pf);
                        if (!q.bail) {
                            if (r == 0 && tc >= (uint32_t)KNP) G->is_first_turn = 0;
                            wave_sync();
                        }
                    }
                }
            }
        }
        R4M(43);
    }
    R4T(2);
    // rows whose round ended (an exhaustive draw above, a finished game under auto-reset) stop here in pass 1: ryukyoku, next round or end
    // of game and the deal happen between the passes (r4_round_end), their observation is pass 2's
    bool wait_deal = RMJ_ROW_ROUND_END && !pass2 && t0 && !q.bail && q.rend != 0u;
    if (q.rend) q.dirty = 0xFu;
    if (pass2 && RICH && __ballot(t0 && q.yk_mode == R4_RE_YAKU_CLAIMS)) {
        // pass 2 of a row that paused at its discard's Ron check: the rest of _resolve_discard with the evaluator's answers.  Whatever makes
        // it leave tier 0 now continues in the full path from HERE (the claims of the discard already made: STEP_F_CONT_CLAIMS), an
        // exhaustive draw at the exhaustive draw (STEP_F_CONT_RYU, set by r4_deal_next)
        if (t0 && q.yk_mode == R4_RE_YAKU_CLAIMS) {
            r4_resolve_discard<RICH, LOOP, true>(q, G->last_discard_pid, G->last_discard_tile, false, pf, nl_mine, w_mine);
            if (q.bail && q.cont == 0) q.cont = 3;
        }
    }
    if (t0 && !wait_deal) {
        // ---- the next observation: a WaitAct state needs the acting seat's list
        if (!q.bail && G->phase == RMJ_WAIT_ACT && !(q.rend && G->is_done)) {
            nl_mine = 0;
            w_mine = 0ull;
            // (an untouched state whose list was empty has an empty list; a holder of 13 tiles - poked states - also publishes its waits: r4_gen_act_legal decides)
            const PState& Pc = G->p[G->current_player & 3];
            if (!(noop && G->nlegal[G->current_player & 3] == 0 && Pc.hand_len + 3 * Pc.n_melds == 14)) r4_gen_act_legal<RICH>(q, nl_mine);
            if (q.rend && q.bail) q.cont = 2;   // (a dealt hand tier 0 has no list for: the full path publishes the state as it stands)
        }
        if (RMJ_ROW_ROUND_END && !pass2 && !q.bail && q.rend != 0u) wait_deal = true;   // (paused at the list's Tsumo check)
    }
    R4M(45);
    R4T(3);
    // ---- publication of the rows that completed in tier 0
    const bool done0 = q.live && t0 && !q.bail && !wait_deal;
    // INLR: the steps this call took for the row's game (a bailed row takes ONE in the full path, from the untouched record, whatever
    // tier 0 had got to), and the row's own publication flags: quiet unless this was the last step of the rollout
    const uint32_t used = (!q.live || pass2) ? 0u : ((INLR && second && (!q.bail || q.cont)) ? 2u : 1u);
    const uint32_t fl_pub = pass2 ? sh.rfl[row] : (INLR ? (flags | ((final_chunk && left == used) ? STEP_F_ALLROWS : STEP_F_QUIET)) : flags);
    if (RMJ_ROW_ROUND_END && r == 0) {
        sh.rmode[row] = wait_deal ? q.rend : 0u;
        sh.rfl[row] = fl_pub;
    }
    if (done0) {
        const uint32_t am = G->active_mask;
        const bool acts = r < 4 && ((am >> r) & 1u);
        const int n_me = acts ? nl_mine : 0;
        // masks: 82-bit id sets per seat by LDS atomics, rows of seats that had or have a list are rewritten.
        // Inside a fused rollout only the LAST step's observation can be read by anybody (the policy reads the lists and the
        // record): the steps before it (STEP_F_QUIET) publish lists, record and events only - no mask rows, no nlegal / waits /
        // status words - and the last one (STEP_F_ALLROWS) rewrites all four mask rows, whatever the quiet steps left behind.
        const bool quiet = LOOP && (fl_pub & STEP_F_QUIET) != 0u;
        const uint32_t rows = quiet ? 0u : ((LOOP && (fl_pub & STEP_F_ALLROWS)) ? 0xFu : ((rballot(r < 4 && G->nlegal[r & 3] != 0, rb) | am) & 0xFu));
        if (!quiet) q.T->mk[row][r >> 2][r & 3] = 0u;
        wave_sync();
        for (uint32_t m = am; m; m &= m - 1u) {
            const int p = __ffs((int)m) - 1;
            const int n = rbc(n_me, rb + p);
            if (r < n) {
                const uint64_t e = q.T->lst[row][p][r];
                (E.legal + (size_t)g * 4 * RMJ_MAX_LEGAL)[p * RMJ_MAX_LEGAL + r] = e & 0x00FFFFFFFFFFFFFFull;
                const int id = (int)(e >> 56);     // the id travels with the staged entry (r4_put)
                if (!quiet) atomicOr(&q.T->mk[row][p][id >> 5], 1u << (id & 31));
            }
            if (R4_ACT_CAP > R4_LIST && r + R4_LIST < n) {   // (3P: a drawer's list that runs on into the next seat's slots)
                const uint64_t e = (&q.T->lst[row][p][0])[r + R4_LIST];
                (E.legal + (size_t)g * 4 * RMJ_MAX_LEGAL)[p * RMJ_MAX_LEGAL + r + R4_LIST] = e & 0x00FFFFFFFFFFFFFFull;
                const int id = (int)(e >> 56);
                if (!quiet) atomicOr(&q.T->mk[row][p][id >> 5], 1u << (id & 31));
            }
        }
        wave_sync();
        uint16_t* mout = reinterpret_cast<uint16_t*>(E.mask + (size_t)g * 328);
        for (uint32_t m = rows; m; m &= m - 1u) {
            const int p = __ffs((int)m) - 1;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const int j = r + 16 * k;   // 16-bit unit = ids 2j, 2j + 1
                if (j < 41) {
                    const uint32_t w = q.T->mk[row][p][(2 * j) >> 5];
                    const uint32_t b0 = (w >> ((2 * j) & 31)) & 1u, b1 = (w >> ((2 * j + 1) & 31)) & 1u;
                    mout[41 * p + j] = (uint16_t)(b0 | (b1 << 8));
                }
            }
        }
        if (r < 4) {
            if (!quiet) E.nlegal[(size_t)g * 4 + r] = (uint8_t)n_me;
            G->nlegal[r] = (uint8_t)n_me;
            if (!quiet) E.waits[(size_t)g * 4 + r] = acts ? w_mine : 0ull;
            if (acts && !G->is_done) {
                G->obs_from[r] = G->obs_upto[r];
                G->obs_upto[r] = G->ev_count;
            }
        }
        if (r == 0 && !quiet) E.status[g] = (uint32_t)G->active_mask | ((uint32_t)G->phase << 8) | ((uint32_t)G->is_done << 16);
        // staged events -> ring
        if (r < 2 * q.evn) {
            const int e = r >> 1, hh = r & 1;
            uint4* dst = reinterpret_cast<uint4*>(E.events + (size_t)g * (E.ring_mask + 1u) + q.T->evidx[row][e]);
            dst[hh] = reinterpret_cast<const uint4*>(q.T->ev[row][e])[hh];
        }
    }
    if (q.live && (q.cont || wait_deal) && r < 2 * q.evn) {   // a row that continues in the full path or in r4_round_end: its staged events go out now (the full path's scratch overlays the staging area)
        const int e = r >> 1, hh = r & 1;
        uint4* dst = reinterpret_cast<uint4*>(E.events + (size_t)g * (E.ring_mask + 1u) + q.T->evidx[row][e]);
        dst[hh] = reinterpret_cast<const uint4*>(q.T->ev[row][e])[hh];
    }
    wave_sync();
    R4M(46);
    R4T(4);
    // ---- records of the completed rows back to HBM (globals + touched PState quarters)
    if (done0) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int off = r + 16 * k;
            if (off < (int)(sizeof(GState) / 16) && (off >= 32 || ((q.dirty >> (off >> 3)) & 1u)))
                reinterpret_cast<uint4*>(E.core + g)[off] = reinterpret_cast<const uint4*>(&sh.st[row])[off];
        }
    }
    wave_sync();
    R4M(47);
    R4T(5);
    // ---- bailed games: the complete state machine, one game at a time, from the untouched HBM record
    uint64_t bm = __ballot(q.live && q.bail && r == 0);
    const uint32_t fl_full = ((INLR || pass2) ? fl_pub : flags) |   // (pass 2 of an inline-response rollout: the row's publication flags come from pass 1 - the caller's `flags` carry neither STEP_F_QUIET nor _ALLROWS there)
                             (q.cont == 1 ? STEP_F_CONT_RYU : 0u) | (q.cont == 2 ? STEP_F_CONT_FIN : 0u) | (q.cont == 3 ? STEP_F_CONT_CLAIMS : 0u);   // (per row)
#ifdef RMJ_TL4
    if (!LOOP && lane == 0) rmj::g_tl4[(size_t)blockIdx.x * RMJ_TL4_ROW + 7] = (unsigned long long)__popcll(bm);
#endif
    while (bm) {
        const int br = (__ffsll((long long)bm) - 1) >> 4;
        bm &= bm - 1ull;
        const uint32_t gg = g0 + (uint32_t)br;
#ifdef RMJ_CENSUS
        {
            const int why_b = __builtin_amdgcn_readlane(q.cont == 2 ? 31 : q.why, 16 * br);
            if (lane == 0) atomicAdd(&rmj::g_bail_reason[why_b & 31], 1u);
        }
#endif
        // the seats' actions of that game move to lanes 0..3
        const uint64_t m_full = rbc64(mine, 16 * br + (lane & 3));
        Ctx c{sh.st[br], E, sh.u.x, gg, lane, E.wall + (size_t)gg * RMJ_WALL_STRIDE, E.legal + (size_t)gg * 4 * RMJ_MAX_LEGAL};
#ifdef RMJ_TL4
        if (lane < 16) sh.u.x.tl_acc[lane] = 0u;
        if (lane == 0) sh.u.x.tl_prev = __builtin_readcyclecounter();
        wave_sync();
#endif
        if (LOOP) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // the full path reads wall, lists and record with plain loads: drop what the L1 may hold of an earlier visit of this game to this CU (see the record fetch above)
#if RMJ_FULL_PRIO
        if (!LOOP || RMJ_FULL_PRIO > 1) __builtin_amdgcn_s_setprio(3);   // experiment: the few long waves of a per-step launch issue ahead of the others of their SIMD
#endif
        ol_step_full(ctx_pack(c), lane < 4 ? m_full : RMJ_NO_ACTION, (uint32_t)__builtin_amdgcn_readlane((int)fl_full, 16 * br));
#if RMJ_FULL_PRIO
        if (!LOOP || RMJ_FULL_PRIO > 1) __builtin_amdgcn_s_setprio(0);
#endif
        wave_sync();
#ifdef RMJ_TL4
        if (!LOOP && lane < 16) rmj::g_tl4[(size_t)blockIdx.x * RMJ_TL4_ROW + 16 + lane] += (unsigned long long)sh.u.x.tl_acc[lane];
        wave_sync();
#endif
    }
    R4T(6);
    if (LOOP) {
        // The next step reads back this step's lists / wall tile with agent-scope loads (served by the XCD's L2, past the
        // vector L1); the stores are this wave's own, to the same addresses and through the same L2 channel, so no cache
        // write-back is needed - an agent-scope release fence would write the whole L2 back (buffer_wbl2: 6x slower).
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        wave_sync();
    }
    return used | (__ballot(wait_deal) ? R4_RET_ROUND : 0u);
    }
}
// The step as an out-of-line function with its own static LDS: the rollout loop calls it once per step, so nothing of a
// step is hoisted out of the loop or kept live across it (the loop inlined: 48 VGPR + 37 SGPR spills).
template <bool LOOP, int POL>
__device__ __noinline__ uint32_t step4_call(const Env* Ep, uint64_t policy_seed, uint32_t flags, uint32_t g_base, uint32_t g_end, uint32_t load,
                                            uint64_t gs_row, uint32_t quad = 0xFFFFFFFFu) {
    Quad4Shared& sh = g_q4;
    return step4_body<LOOP, POL>(uni_ptr(Ep), sh, uni(policy_seed), uni(flags), uni(g_base), uni(g_end), uni(load) != 0u, gs_row, nullptr, uni(quad));
}
// Pass 2 of a step whose wave ended rounds (R4_RET_ROUND): r4_round_end has dealt them, their rows get the dealer's first list and are
// published (step4_body<.., PASS2>: no policy, no transitions - a small function, out of line, entered once in ~25 wave-steps)
template <bool LOOP, int POL>
__device__ __noinline__ void step4_pass2(const Env* Ep, uint32_t flags, uint32_t g_base, uint32_t g_end, uint32_t quad = 0xFFFFFFFFu) {
    Quad4Shared& sh = g_q4;
    step4_body<LOOP, POL, false, true>(uni_ptr(Ep), sh, 0ull, uni(flags), uni(g_base), uni(g_end), false, 0ull, nullptr, uni(quad));
}
template <bool LOOP, int POL>
__device__ __forceinline__ void step4_finish_rounds(const Env* Ep, uint32_t flags, uint32_t g_base, uint32_t g_end, uint32_t quad = 0xFFFFFFFFu) {
    constexpr bool RICH = (POL == 1) || !LOOP;   // (step4_body's tier)
    const uint32_t md = g_q4.rmode[(threadIdx.x & 63u) >> 4];
#ifdef RMJ_RE_PROF
    if ((threadIdx.x & 15u) == 0u) atomicAdd(&rmj::g_re_prof[8 + (md & 15u)], 1ull);
#endif
    if (__ballot(md == R4_RE_DRAW || md == R4_RE_RESTART || md == R4_RE_WIN_TSUMO || md == R4_RE_WIN_RON)) {
        if (LOOP && RICH) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // (the settlement reads the ura indicators off the wall slab with plain loads)
#ifdef RMJ_RE_PROF
        const unsigned long long pt0 = __builtin_amdgcn_s_memrealtime();
#endif
        r4_round_end<RICH>(Ep, g_base + (quad == 0xFFFFFFFFu ? blockIdx.x : quad) * r4_rows(flags));
#ifdef RMJ_RE_PROF
        if ((threadIdx.x & 63u) == 0u) { const unsigned long long tr = __builtin_amdgcn_s_memrealtime(); atomicAdd(&rmj::g_re_prof[0], tr - pt0); atomicAdd(&rmj::g_re_prof[1], 1ull); atomicAdd(&rmj::g_re_prof[17], pt0); atomicAdd(&rmj::g_re_prof[19], tr); }
#endif
    }
    if (RICH && __ballot(md == R4_RE_YAKU_CLAIMS || md == R4_RE_YAKU_TSUMO)) r4_yaku_answers();   // (the lean tier never pauses for a yaku check)
#ifdef RMJ_RE_PROF
    const unsigned long long pt1 = __builtin_amdgcn_s_memrealtime();
#endif
    step4_pass2<LOOP, POL>(Ep, flags, g_base, g_end, quad);
#ifdef RMJ_RE_PROF
    if ((threadIdx.x & 63u) == 0u) { atomicAdd(&rmj::g_re_prof[2], __builtin_amdgcn_s_memrealtime() - pt1); atomicAdd(&rmj::g_re_prof[3], 1ull); }
#endif
}
// ... with inline responses (step4_body<.., INLR>): `left` steps to go per row, returns the steps taken per row
#ifndef RMJ_INLINE_RESP
#define RMJ_INLINE_RESP 3   /* bit 0: the RandomAgent's rollouts, bit 1: the greedy policy's */
#endif
// Round 5: the step of the fused RandomAgent / greedy rollouts is INLINED into the rollout loop.  Out of line it paid, per call, 33
// callee-saved SGPRs through v_writelane / v_readlane (64 vector instructions of ~1 445) and 13 callee-saved VGPRs through scratch; the
// inlined loop of round 4 spilled far worse (45 VGPR + 66 SGPR) because everything derived from the LANE ID - row, rb, LDS addresses of the
// row's record - is loop invariant too and was hoisted and kept live across the step.  With the lane id laundered per iteration as well
// (step4_body) the loop spills 17 VGPR / 52 SGPR in the ticket kernel, mostly outside the hot sections: +5 % at every batch size
// (profiles/r05_inline_step_ab.txt).  RMJ_INLINE_STEP=0 brings the out-of-line step back (A/B).
#if RMJ_INLINE_STEP
#define R4_CALL_ATTR __forceinline__
#else
#define R4_CALL_ATTR __noinline__
#endif
template <int POL>
__device__ R4_CALL_ATTR uint32_t step4_call_inl(const Env* Ep, uint64_t policy_seed, uint32_t flags, uint32_t g_base, uint32_t g_end, uint32_t load,
                                                uint64_t gs_row, uint32_t quad, uint32_t left, uint32_t final_chunk) {
    Quad4Shared& sh = g_q4;
#if RMJ_INLINE_STEP
    // laundered as SCALAR registers (a vector-register launder made the Env pointer live in - and spill from - VGPRs: a scratch reload at
    // every event emission); readfirstlane first, so that the operands are scalar wherever this body is compiled
    Ep = uni_ptr(Ep); policy_seed = uni(policy_seed); flags = uni(flags); g_base = uni(g_base); g_end = uni(g_end); quad = uni(quad); final_chunk = uni(final_chunk);
    asm volatile("" : "+s"(Ep), "+s"(policy_seed), "+s"(flags), "+s"(g_base), "+s"(g_end), "+s"(quad), "+s"(final_chunk));
#endif
    return step4_body<true, POL, true>(uni_ptr(Ep), sh, uni(policy_seed), uni(flags), uni(g_base), uni(g_end), uni(load) != 0u, gs_row, nullptr, uni(quad),
                                       left, uni(final_chunk) != 0u);
}
// the steps [0, steps) of a quad's rollout / ticket: the RandomAgent answers claims inline (rows run ahead of each other by a step or
// two and wait at the end), the greedy policy steps all rows in lock-step
// max_calls: the ticket ends after that many calls of the step function even if rows have steps left (k_step4_queue: a ticket is a
// number of CALLS, the rows carry what is left of their rollout to the quad's next ticket - no row idles at the end of a ticket while
// the others catch up); the result = the row's steps still to take.
template <int POL>
__device__ __forceinline__ uint32_t step4_run(const Env* Ep, uint64_t policy_seed, uint32_t flags, uint32_t g_base, uint32_t g_end, uint64_t gs_row,
                                              uint32_t quad, uint32_t steps, bool final_chunk, uint32_t g_row, uint32_t max_calls = 0xFFFFFFFFu) {
    if (RMJ_INLINE_RESP & (POL == 0 ? 1 : 2)) {
        uint32_t left = g_row < g_end ? steps : 0u, load = 1u;
#ifdef RMJ_QTL
        if ((threadIdx.x & 63u) == 0u) { g_qtl_cnt[0] = 0u; g_qtl_cnt[1] = 0u; }
#endif
#pragma unroll 1
        while (__ballot(left != 0u) && max_calls-- != 0u) {
#ifdef RMJ_QTL
            { const uint64_t lv = __ballot(left != 0u && (threadIdx.x & 15u) == 0u); if ((threadIdx.x & 63u) == 0u) { g_qtl_cnt[0] += 1u; g_qtl_cnt[1] += (uint32_t)__popcll(lv); } }
#endif
            const uint32_t ret = step4_call_inl<POL>(Ep, policy_seed, flags, g_base, g_end, load, gs_row, quad, left, final_chunk ? 1u : 0u);
            left -= ret & 0xFFu;
            load = 0u;
            if (RMJ_ROW_ROUND_END && __ballot((ret & R4_RET_ROUND) != 0u)) step4_finish_rounds<true, POL>(Ep, flags, g_base, g_end, quad);   // (wave-uniform bit)
        }
        return left;
    } else {
        const uint32_t n = steps < max_calls ? steps : max_calls;   // (lock-step: every call is one step of every row)
#pragma unroll 1
        for (uint32_t it = 0; it < n; it++) {
            const uint32_t fl = flags | ((final_chunk && it + 1u == steps) ? STEP_F_ALLROWS : STEP_F_QUIET);
            const uint32_t ret = step4_call<true, POL>(Ep, policy_seed, fl, g_base, g_end, it == 0 ? 1u : 0u, gs_row, quad);
            if (RMJ_ROW_ROUND_END && __ballot((ret & R4_RET_ROUND) != 0u)) step4_finish_rounds<true, POL>(Ep, fl, g_base, g_end, quad);
        }
        return g_row < g_end ? steps - n : 0u;
    }
}
#ifndef RMJ_STEP4_WAVES
#define RMJ_STEP4_WAVES 6
#endif
// The greedy policy's fused rollouts at five waves per SIMD (96 VGPR): 49 instead of 59 vector registers spilled in the ticket kernel, +4.5 % (931-941 ->
// 977-979 M env.step/s); the RandomAgent's rollouts lose 10 % at five (1 890 -> 1 700 M) and stay at six.
#ifndef RMJ_STEP4_WAVES_GREEDY
#define RMJ_STEP4_WAVES_GREEDY 5
#endif
#define RMJ_STEP4_WAVES_OF(POL) ((POL) == 1 ? RMJ_STEP4_WAVES_GREEDY : RMJ_STEP4_WAVES)
// LOOP = false: one step per launch.  LOOP = true: games are independent, so a device-policy rollout needs no
// synchronisation between the steps of DIFFERENT games: the wave keeps its four records in LDS and steps its own games
// n_steps times (publishing every step's outputs exactly like n_steps launches would).  A launch per step ends with the
// slowest wave (the one that restarts a round), and at 65 536 games (16 384 waves = two generations of resident waves)
// that tail costs as much as the work; the loop pays it once per rollout.  POL: 0 = RandomAgent / the caller's actions, 1 = the
// greedy policy (r4_policy_greedy).
// Heavy-first launch order of the per-step kernel (LOOP = false, whole-batch launches).  A launch ends with its slowest wave, and
// which waves will be slow is known one step ahead: a game whose wall is exhausted ends its round with the next discard (r4_round_end:
// twice a plain wave's lifetime), a finished game restarts, an offered Ron may be taken (settlement: full path).  Every wave leaves a
// note for the next launch - its unit in a list (one atomic) and a flag - and the next launch starts with `front` blocks that take the
// listed units; the blocks behind them serve the others in place and leave at once where a front block has been.  Hints only: the two
// arrays are written by one launch and read by the next, so every unit is served exactly once whatever happened in between.
// (struct HeavyOrder: rmj_common.hip.h)
#ifndef RMJ_DEBUG_STEP_WAVES
#define RMJ_DEBUG_STEP_WAVES RMJ_STEP4_WAVES
#endif
template <bool LOOP, int POL>
__global__ __launch_bounds__(64, (LOOP && POL == 1) ? RMJ_STEP4_WAVES_GREEDY : RMJ_DEBUG_STEP_WAVES) void k_step4(const Env* __restrict__ Ep, uint64_t policy_seed, uint32_t flags, uint32_t g_base,
                                                                   uint32_t g_end, uint32_t n_steps, const uint64_t* __restrict__ actions, HeavyOrder ho) {
    if (LOOP) {
        const uint32_t row_ = (threadIdx.x & 63u) >> 4;
        const uint32_t g = row_ < r4_rows(flags) ? g_base + blockIdx.x * r4_rows(flags) + row_ : 0xFFFFFFFFu;   // (rows beyond the wave's games idle)
        const uint64_t gs_row = sm64(policy_seed + ((CEnv*)Ep)->game_offset + (uint64_t)g);   // policy key of the row's game
        step4_run<POL>(Ep, policy_seed, flags, g_base, g_end, gs_row, 0xFFFFFFFFu, n_steps, true, g);
    } else {
        Quad4Shared& sh = g_q4;
#ifdef RMJ_TL4
        const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
#endif
        uint32_t unit = blockIdx.x;
        if (ho.in_cnt) {
            if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) *ho.zero_cnt = 0u;
            if (blockIdx.x < ho.front) {
                if (blockIdx.x >= uni(*ho.in_cnt)) return;
                unit = uni(ho.in_list[blockIdx.x]);
            } else {
                unit = blockIdx.x - ho.front;
                if (uni((uint32_t)ho.in_flag[unit])) return;
            }
        }
        const uint32_t ret = step4_body<false, POL>(Ep, sh, policy_seed, flags, g_base, g_end, true, 0ull, actions, unit);
#ifdef RMJ_TL4
        if ((threadIdx.x & 63) == 0) rmj::g_tl4[(size_t)blockIdx.x * RMJ_TL4_ROW + 15] = __ballot((ret & R4_RET_ROUND) != 0u) ? 1ull : 0ull;   // the wave ends rounds
        if (__ballot((ret & R4_RET_ROUND) != 0u) && (threadIdx.x & 63) == 0)   // ... and why (bit R4_RE_* per row)
            rmj::g_tl4[(size_t)blockIdx.x * RMJ_TL4_ROW + 14] = (1ull << g_q4.rmode[0]) | (1ull << g_q4.rmode[1]) | (1ull << g_q4.rmode[2]) | (1ull << g_q4.rmode[3]);
#endif
        if (RMJ_ROW_ROUND_END && __ballot((ret & R4_RET_ROUND) != 0u)) step4_finish_rounds<false, POL>(Ep, flags, g_base, g_end, unit);
        if (ho.out_cnt) {   // the note for the next launch
            const int row = (threadIdx.x & 63) >> 4;
            const GState& S = sh.st[row];
            const uint32_t gq = g_base + unit * r4_rows(flags) + (uint32_t)row;
            // Round 5: ... and a seat that waits without a riichi - a discard into its wait or its own winning draw makes the wave pause for
            // the evaluator (R4_RE_YAKU_CLAIMS / _TSUMO: twice a plain wave's lifetime); served in place, 10-30 such waves per launch started
            // at 40 us and ended it at 75-88 us instead of 70 (scripts/timeline4.py detail)
            const PState& Pw = S.p[threadIdx.x & 3];
            // (a seat whose cache is stale - it has just discarded from its hand or called - stays unpredicted: 2-6 waves per launch.  Its old
            //  shanten bound is no predictor: "sh13 <= 1" holds for most hands - the list then takes an atomic from nearly every wave, 163 us per launch)
            const bool waits_open = RMJ_HEAVY_TENPAI && (threadIdx.x & 15) < 4 && (Pw.flags & PF_WAITS_VALID) && Pw.waits13 != 0ull && !(Pw.flags & PF_RIICHI_DECLARED);
            const bool hv = (uint32_t)row < r4_rows(flags) && gq < g_end &&
                            (S.is_done ? (flags & STEP_F_AUTORESET) != 0u : ((S.phase == RMJ_WAIT_ACT && S.drawable_count == 0) || S.ron_offer_mask != 0 || waits_open));
            const bool any = __ballot(hv) != 0ull;
            if ((threadIdx.x & 63) == 0) {
                uint32_t idx = 0xFFFFFFFFu;
                if (any) idx = atomicAdd(ho.out_cnt, 1u);
                if (idx < ho.front) ho.out_list[idx] = unit;
                ho.out_flag[unit] = idx < ho.front ? 1 : 0;
            }
        }
#ifdef RMJ_TL4
        const unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();
        if ((threadIdx.x & 63) == 0) {
            rmj::g_tl4[(size_t)blockIdx.x * RMJ_TL4_ROW + 8] = rt0;
            rmj::g_tl4[(size_t)blockIdx.x * RMJ_TL4_ROW + 9] = rt1;
        }
#endif
    }
}

// The fused rollout with the work handed out in pieces.  k_step4<true> gives every wave one quad for the whole rollout; at 65 536
// games that is 16 384 waves for ~7 168 wave slots, and the last 2 000 of them run on a chip that is three quarters empty
// (524 288 games, where that tail does not matter, step 10 % faster).  Here a grid that fits the chip once pulls (quad, chunk)
// tickets: chunk c of a quad = its steps [c * chunk, (c + 1) * chunk), in ticket order chunk-major, so chunk c - 1 of the same quad
// was handed out a full round earlier.  A quad's chunks stay on ONE XCD (queue per XCD, quads dealt by quad % 8; the XCD is read
// from the hardware register, not guessed from the block index): the per-XCD L2 is then the single point through which one
// wave's record / lists / wall reach the next - stores are write-through to it, the hand-over is "drain stores, publish the chunk
// count with an agent-scope atomic", the pick-up "poll it, invalidate this CU's L1 (acquire), load".  No cross-XCD traffic, no
// L2 write-back.  Results are those of k_step4<true>: every game steps n_steps times with the same policy keys.
// The queue keys (HW_REG_XCC_ID & 7, quad % 8) assume at most eight XCC ids: rmj_create probes the ids the device reports and the
// host uses this kernel only when none exceeds 7 (rollout_queued), otherwise every wave keeps its quad (k_step4<true>).
#define RMJ_Q_STRIDE 32u   /* u32 words per XCD queue head (its own 128-byte line) */
// (out of line: inlined into the ticket loop, the one-lane branches below were restructured into a loop nest that re-used a stale
// ticket - the kernel then ran quads twice at once)
__device__ __noinline__ uint32_t q_take_ticket(uint32_t* head) {
    uint32_t t = 0u;
    if ((threadIdx.x & 63u) == 0u) t = __hip_atomic_fetch_add(head, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
}
#define RMJ_Q_FIN 0x80000000u   /* done[quad]: every row of the quad has taken its n_steps - the quad's later tickets are empty */
// waits until the quad's ticket c - 1 has ended (done = chunks ended | RMJ_Q_FIN); returns the word
__device__ __noinline__ uint32_t q_wait_for(const uint32_t* slot, uint32_t want) {
    uint32_t v = 0u;
    if ((threadIdx.x & 63u) == 0u)
        for (;;) {
            v = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((v & RMJ_Q_FIN) || v >= want) break;
            __builtin_amdgcn_s_sleep(8);
        }
    wave_sync();
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
// A ticket = up to `chunk` CALLS of the step function on one quad.  With inline responses a row advances one or two game-steps per
// call, so a ticket that had to bring every row to the same step count ended with rows idling while the others caught up (8-step
// tickets: 3.62 of 4 rows live per call, 64-step tickets 3.85 - profiles/r05_ticket_timeline.txt); instead every game carries the
// steps it has taken in this rollout from ticket to ticket (prog[game], zeroed with the ticket counters) and a ticket simply ends
// after `chunk` calls.  A quad needs at most ceil(n_steps / chunk) tickets (a call advances every unfinished row); once all its
// rows are through, done[quad] carries RMJ_Q_FIN and the quad's remaining tickets are empty (one poll, no record fetched).
// (74 VGPRs: 6 waves per SIMD = 6 144 slots.  Compiled for 7 - 71 VGPRs, four scratch accesses in the ticket loop - it is not faster:
//  1 617 / 1 504 M env.step/s against 1 642 / 1 529 M in 4P / 3P; the seventh wave was worth its 10 % mostly because it shortened the tail.
//  Launch bounds for 4 / 5 / 8 waves: 1 532 / 1 279, 1 530 / 1 454, 1 426 / 1 382 - 6 is the optimum)
template <int POL>
__global__ __launch_bounds__(64, RMJ_STEP4_WAVES_OF(POL)) void k_step4_queue(const Env* __restrict__ Ep, uint64_t policy_seed, uint32_t flags, uint32_t n_games,
                                                                         uint32_t n_steps, uint32_t chunk, uint32_t* __restrict__ heads, uint32_t* __restrict__ done,
                                                                         uint32_t skip_xcds, uint32_t* __restrict__ prog, uint32_t tail) {
    // (the counters arrive zeroed: k_step4_fixup, which runs behind every ticket launch, re-arms the set it has looked at - no memset between
    //  rollouts, and the pair of launches is idempotent: a rollout captured in a HIP graph can be replayed.  Round 5 alternated two sets from
    //  the host and had each launch zero the other, which a replay of ONE captured launch found exhausted: ADVICE r5)
    const uint32_t xcd = (uint32_t)__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;   // HW_REG_XCC_ID[3:0]
    if ((skip_xcds >> xcd) & 1u) return;   // test hook (RMJ_QUEUE_TEST_SKIP_XCDS): pretend these XCDs received no block -> k_step4_fixup
    const uint32_t n_quads = (n_games + 3u) / 4u;
    const uint32_t mine = n_quads > xcd ? (n_quads - xcd + 7u) / 8u : 0u;             // quads xcd, xcd + 8, ...
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n_chunks = q_ticket_plan(n_steps, chunk, tail);   // (no table in LDS: the wave's 6 336 bytes are spoken for, see Quad4Shared)
    if (mine == 0u) return;
#ifdef RMJ_QTL   /* ticket timeline build (scripts/timeline_queue.py, never the shipped library): per wave and ticket [ticket | xcd << 32, taken, begun, ended] on the 100 MHz clock */
    uint32_t qtl_n = 0u;
    unsigned long long* const qtl = rmj::g_qtl + (size_t)blockIdx.x * RMJ_QTL_ROW;
    if (lane == 0u) { qtl[0] = __builtin_amdgcn_s_memrealtime(); qtl[3] = (unsigned long long)(uint32_t)__builtin_amdgcn_s_getreg((31 << 11) | 4); }   // HW_REG_HW_ID
#endif
#pragma unroll 1
    for (;;) {
        const uint32_t t = q_take_ticket(heads + xcd * RMJ_Q_STRIDE);
        if (t >= mine * n_chunks) break;
#ifdef RMJ_QTL
        const unsigned long long qt0 = __builtin_amdgcn_s_memrealtime();
#endif
        const uint32_t c = t / mine, quad = (t - c * mine) * 8u + xcd;
        if (c > 0u && (q_wait_for(done + quad, c) & RMJ_Q_FIN)) continue;   // the quad's previous ticket (handed out `mine` tickets ago) must have ended; nothing left: an empty ticket
#ifdef RMJ_Q_ACQ   /* (A/B only, profiles/r05_ticket_acquire_ab.txt: the agent-scope acquire = an L1 invalidate per pick-up, as in rounds 2-4) */
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
#ifdef RMJ_QTL
        const unsigned long long qt1 = __builtin_amdgcn_s_memrealtime();
#endif
        const uint32_t g = quad * 4u + (lane >> 4);
        const uint64_t gs_row = sm64(policy_seed + ((CEnv*)Ep)->game_offset + (uint64_t)g);
        uint32_t taken = 0u;
        if (c > 0u && g < n_games) taken = __hip_atomic_load(prog + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // (the quad's last possible ticket runs until every row is through, whatever a call achieved)
        const uint32_t left = step4_run<POL>(Ep, policy_seed, flags, 0u, n_games, gs_row, quad, n_steps - taken, true, g, c + 1u == n_chunks ? 0xFFFFFFFFu : (tail ? q_ticket_len(n_steps, chunk, c) : chunk));
        if ((lane & 15u) == 0u && g < n_games) prog[g] = n_steps - left;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this ticket's stores are in the XCD's L2
        wave_sync();
        const bool fin = __ballot(left != 0u) == 0ull;
        if (lane == 0u) __hip_atomic_store(done + quad, (c + 1u) | (fin ? RMJ_Q_FIN : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef RMJ_QTL
        if (lane == 0u && 4u + 4u * qtl_n + 3u < RMJ_QTL_ROW) {
            qtl[4u + 4u * qtl_n] = (unsigned long long)t | ((unsigned long long)xcd << 32) | ((unsigned long long)(g_qtl_cnt[0] & 0xFFu) << 40) | ((unsigned long long)(g_qtl_cnt[1] & 0xFFFFu) << 48);
            qtl[5u + 4u * qtl_n] = qt0; qtl[6u + 4u * qtl_n] = qt1; qtl[7u + 4u * qtl_n] = __builtin_amdgcn_s_memrealtime();
        }
        qtl_n += 1u;
#endif
    }
#ifdef RMJ_QTL
    if (lane == 0u) { qtl[1] = __builtin_amdgcn_s_memrealtime(); qtl[2] = qtl_n; }
#endif
}

// Safety net of k_step4_queue: where blocks run is not ours to decide - a quad whose XCD received no block at all (a partitioned
// device, a dispatcher that skips an XCD) has done[quad] == 0 and is stepped here, by one wave for the whole rollout like
// k_step4<true> (nothing of it has run yet, so no other cache holds newer data).  One wave looks at 64 quads (a launch of one
// block per quad cost the 20-step window as much as 1 % of its time to find nothing).
template <int POL>
__global__ __launch_bounds__(64, RMJ_STEP4_WAVES_OF(POL)) void k_step4_fixup(const Env* __restrict__ Ep, uint64_t policy_seed, uint32_t flags, uint32_t n_games,
                                                                         uint32_t n_steps, uint32_t* __restrict__ done) {
    const uint32_t n_quads = (n_games + 3u) / 4u, lane = threadIdx.x & 63u, mine = blockIdx.x * 64u + lane;
    uint64_t todo = __ballot(mine < n_quads && __hip_atomic_load(done + (mine < n_quads ? mine : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u);
    // re-arm the counter set for the next ticket rollout (the ticket kernel has ended: stream order): this wave's 64 quad words, and - block 0 -
    // the eight queue heads in front of them
    if (mine < n_quads) done[mine] = 0u;
    if (blockIdx.x == 0u)
        for (uint32_t i = lane; i < 8u * RMJ_Q_STRIDE; i += 64u) (done - 8u * RMJ_Q_STRIDE)[i] = 0u;
#pragma unroll 1
    while (todo) {
        const uint32_t quad = blockIdx.x * 64u + (uint32_t)(__ffsll((long long)todo) - 1);
        todo &= todo - 1ull;
        const uint32_t g = quad * 4u + (lane >> 4);
        const uint64_t gs_row = sm64(policy_seed + ((CEnv*)Ep)->game_offset + (uint64_t)g);
        (void)step4_run<POL>(Ep, policy_seed, flags, 0u, n_games, gs_row, quad, n_steps, true, g);
        wave_sync();
    }
}


// ---------------------------------------------------------------------------------------------------------------------------
// The fused rollout WITH feature output (BASELINE configs[4]: sanma with the feature tensor): after every step of its four games
// the wave writes Observation.encode() of the seats that are to act now into the resident tensor out[n][4][74][W] - the same rows
// with the same contents as rmj_step_random + rmj_encode_device(only_active = 2) per step, without a launch boundary between the
// issue-bound step and the store-bound encoder: while one wave streams its rows out, the others step.  The records are already in
// LDS (no second fetch), the encoder's byte staging lives in the union the step has finished with.  Compiled for five waves per
// SIMD: the encoder wants 85-96 registers.
// (six waves per SIMD since the value table shrank to 64 entries - 6 512 B of LDS per wave, 80 VGPR without a spill: trainer loop +1.5 %, the 3P
//  step + encode rollout the same: profiles/r05_enc_waves_ab.txt.  The launch is bound by its LDS and vector instruction streams together, not by occupancy.)
#ifndef RMJ_STEP4_ENC_WAVES
#define RMJ_STEP4_ENC_WAVES 6
#endif
#ifndef RMJ_INLINE_ENC
#define RMJ_INLINE_ENC 0     /* experiment: step + encode inlined into the rollout loops of k_step4_enc / k_step4_queue_enc (see step4_call_inl) */
#endif
template <bool LOOP, int POL>
__device__ __forceinline__ void step4_enc_impl(const Env* Ep, uint64_t policy_seed, uint32_t flags, uint32_t g_base, uint32_t g_end, uint32_t load,
                                               uint64_t gs_row, uint32_t quad, float* out, const uint64_t* actions = nullptr) {
    Quad4Shared& sh = g_q4;
    __shared__ float lut[ENC_LUT];
    constexpr int W = KSANMA ? ENC_W3 : ENC_W4;
    int lane_ = threadIdx.x & 63;
#if RMJ_INLINE_ENC
    if constexpr (LOOP) asm volatile("" : "+v"(lane_));
#endif
    const int lane = lane_;
    g_base = uni(g_base); g_end = uni(g_end); quad = uni(quad);
    out = uni_ptr(out);
    if (uni(load) != 0u) enc_lut_init(lut, lane);
    {
        const uint32_t ret = step4_body<LOOP, POL>(uni_ptr(Ep), sh, uni(policy_seed), uni(flags), g_base, g_end, uni(load) != 0u, gs_row, LOOP ? nullptr : uni_ptr(actions), quad);
        if (RMJ_ROW_ROUND_END && __ballot((ret & R4_RET_ROUND) != 0u)) step4_finish_rounds<LOOP, POL>(uni_ptr(Ep), uni(flags), g_base, g_end, quad);
    }
    wave_sync();
    const uint32_t g0 = g_base + (quad == 0xFFFFFFFFu ? blockIdx.x : quad) * 4u;
#pragma unroll 1
    for (int row = 0; row < 4; row++) {
        const uint32_t g = g0 + (uint32_t)row;
        if (g >= g_end) break;
        const GState& S = sh.st[row];
        uint32_t am = U((uint32_t)S.is_done) ? 0u : (U((uint32_t)S.active_mask) & (KSANMA ? 7u : 15u));
        while (am) {
            const int seat = __builtin_ctz(am);
            am &= am - 1u;
            float* dst = out + ((size_t)g * 4 + (size_t)seat) * (size_t)((CEnv*)Ep)->enc_stride;
            const int head = (int)(((16u - (uint32_t)(reinterpret_cast<uintptr_t>(dst) & 15u)) & 15u) >> 2);  // 0 or 2 floats
            EncByteSink<W> o{sh.u.e.raw + ((4 - head) & 3), lut, lane, -1.0f};
            encode_seat_to<KSANMA>(S, seat, lane, sh.u.e.hist, o, true);
            enc_emit_bytes<W>(dst, o.cells, lut, lane, head, o.big);
            wave_sync();
        }
    }
}
template <bool LOOP, int POL>
__device__ __noinline__ void step4_call_enc_ool(const Env* Ep, uint64_t policy_seed, uint32_t flags, uint32_t g_base, uint32_t g_end, uint32_t load,
                                                uint64_t gs_row, uint32_t quad, float* out, const uint64_t* actions = nullptr) {
    step4_enc_impl<LOOP, POL>(Ep, policy_seed, flags, g_base, g_end, load, gs_row, quad, out, actions);
}
// what the rollout loops call: out of line (rounds 2-4), or - RMJ_INLINE_ENC - inlined with the loop-invariant inputs laundered per iteration
template <bool LOOP, int POL>
__device__ __forceinline__ void step4_call_enc(const Env* Ep, uint64_t policy_seed, uint32_t flags, uint32_t g_base, uint32_t g_end, uint32_t load,
                                               uint64_t gs_row, uint32_t quad, float* out, const uint64_t* actions = nullptr) {
#if RMJ_INLINE_ENC
    if constexpr (LOOP) {
        asm volatile("" : "+v"(Ep), "+v"(policy_seed), "+v"(g_base), "+v"(g_end), "+v"(quad), "+v"(out));
        step4_enc_impl<LOOP, POL>(Ep, policy_seed, flags, g_base, g_end, load, gs_row, quad, out, actions);
        return;
    }
#endif
    step4_call_enc_ool<LOOP, POL>(Ep, policy_seed, flags, g_base, g_end, load, gs_row, quad, out, actions);
}
template <int POL>
__global__ __launch_bounds__(64, RMJ_STEP4_ENC_WAVES) void k_step4_enc(const Env* __restrict__ Ep, uint64_t policy_seed, uint32_t flags, uint32_t g_base,
                                                                           uint32_t g_end, uint32_t n_steps, float* __restrict__ out) {
    const uint32_t g = g_base + blockIdx.x * 4u + ((threadIdx.x & 63u) >> 4);
    const uint64_t gs_row = sm64(policy_seed + ((CEnv*)Ep)->game_offset + (uint64_t)g);
#pragma unroll 1
    for (uint32_t it = 0; it < n_steps; it++)
        step4_call_enc<true, POL>(Ep, policy_seed, flags | (it + 1u < n_steps ? STEP_F_QUIET : STEP_F_ALLROWS), g_base, g_end, it == 0 ? 1u : 0u, gs_row, 0xFFFFFFFFu, out);
}
// One step driven by the caller's action ids / packed actions + the rows of the seats that are to act next: what a trainer loop
// issues per iteration (rmj_step_ids_encode_device) - one launch instead of a step launch and an encoder launch, the encoder's
// stores under the tail of the step (the last third of a per-step launch belongs to the few waves that carry a full-path game).
__global__ __launch_bounds__(64, RMJ_STEP4_ENC_WAVES) void k_step4_act_enc(const Env* __restrict__ Ep, uint32_t flags, uint32_t g_base, uint32_t g_end,
                                                                               const uint64_t* __restrict__ actions, float* __restrict__ out) {
#ifdef RMJ_DEBUG_LDS_FILL   /* debugging aid: what does the kernel read of LDS it has not written? */
    for (int i = threadIdx.x & 63; i < (int)(sizeof(Quad4Shared) / 4); i += 64) reinterpret_cast<uint32_t*>(&g_q4)[i] = RMJ_DEBUG_LDS_FILL;
    wave_sync();
#endif
#ifdef RMJ_ACT_ENC_STAGGER   /* experiment: every other wave starts late - do the waves of a generation run their step and store phases in lock-step? */
    if (blockIdx.x & 1u)
        for (int k = 0; k < RMJ_ACT_ENC_STAGGER; k++) __builtin_amdgcn_s_sleep(127);
#endif
#ifdef RMJ_DEBUG_PAD_VGPR   /* debugging aid (round 6, journal r06 section 1): -DRMJ_DEBUG_PAD_VGPR=95 makes the kernel NAME v95, so its waves are allocated 96 vector registers while the out-of-line step compiles to the same instructions as without it */
#define RMJ_STR2(x) #x
#define RMJ_STR(x) RMJ_STR2(x)
    asm volatile("v_mov_b32 v" RMJ_STR(RMJ_DEBUG_PAD_VGPR) ", 0" ::: "v" RMJ_STR(RMJ_DEBUG_PAD_VGPR));
#endif
#ifdef RMJ_DEBUG_HWID   /* debugging aid (round 6, scripts/debug_scratch_poison.py --hwid; value = blocks recorded): where and when every wave of the launch ran - taken around the call, the out-of-line function itself is untouched */
    const unsigned long long hw_t0 = __builtin_amdgcn_s_memrealtime();
#endif
#ifdef RMJ_DEBUG_ACT_ENC_INLINE
    step4_enc_impl<false, 0>(Ep, 0ull, flags, g_base, g_end, 1u, 0ull, 0xFFFFFFFFu, out, actions);
#else
    step4_call_enc<false, 0>(Ep, 0ull, flags, g_base, g_end, 1u, 0ull, 0xFFFFFFFFu, out, actions);
#endif
#ifdef RMJ_DEBUG_HWID
    if ((threadIdx.x & 63) == 0 && blockIdx.x < RMJ_DEBUG_HWID) {
        unsigned long long* o = rmj::g_dbg_hwid + (size_t)blockIdx.x * 4;
        o[0] = hw_t0;
        o[1] = __builtin_amdgcn_s_memrealtime();
        o[2] = (unsigned long long)(uint32_t)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)(uint32_t)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);   // HW_ID | XCC_ID
        o[3] = (unsigned long long)(uint32_t)__builtin_amdgcn_s_getreg((31 << 11) | 5) | ((unsigned long long)(uint32_t)__builtin_amdgcn_s_getreg((31 << 11) | 6) << 32);    // GPR_ALLOC | LDS_ALLOC
    }
#endif
}
// ... with the policy's draw in front (rmj_step_sample_encode_device): the wave samples one id per acting seat of its own four games from the
// caller's logits (sample_ids_row: the keyed Gumbel draw of k_sample_ids, the same ids), hands them to the caller (d_ids) and steps under
// them - every lane reads back the id it has just stored itself (program order), no second launch and no pass over all games in between.
__global__ __launch_bounds__(64, RMJ_STEP4_ENC_WAVES) void k_step4_sample_enc(const Env* __restrict__ Ep, uint32_t flags, uint32_t g_base, uint32_t g_end,
                                                                                  const float* __restrict__ logits, uint32_t stride, uint64_t seed,
                                                                                  int32_t* __restrict__ ids, float* __restrict__ out) {
    {
        CEnv& E = *(CEnv*)Ep;
        const int lane = threadIdx.x & 63;
        const uint32_t g = g_base + blockIdx.x * 4u + (uint32_t)(lane >> 4);
        const bool in = g < g_end;
        const int32_t res = sample_ids_row(E.status, E.core, E.nlegal, E.mask, E.game_offset, (int)E.game_mode, g, in, logits, stride, seed, lane);
        if (in && (lane & 15) < 4) ids[(size_t)g * 4 + (lane & 15)] = res;
    }
    step4_call_enc<false, 0>(Ep, 0ull, flags, g_base, g_end, 1u, 0ull, 0xFFFFFFFFu, out, reinterpret_cast<const uint64_t*>(ids));
}
// the same as tickets (see k_step4_queue): a quad's chunks - its records, lists and tensor rows - stay on one XCD
template <int POL>
__global__ __launch_bounds__(64, RMJ_STEP4_ENC_WAVES) void k_step4_queue_enc(const Env* __restrict__ Ep, uint64_t policy_seed, uint32_t flags, uint32_t n_games,
                                                                                 uint32_t n_steps, uint32_t chunk, uint32_t* __restrict__ heads, uint32_t* __restrict__ done,
                                                                                 uint32_t skip_xcds, float* __restrict__ out) {
    const uint32_t xcd = (uint32_t)__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;   // HW_REG_XCC_ID[3:0]
    if ((skip_xcds >> xcd) & 1u) return;
    const uint32_t n_quads = (n_games + 3u) / 4u;
    const uint32_t mine = n_quads > xcd ? (n_quads - xcd + 7u) / 8u : 0u;
    const uint32_t n_chunks = (n_steps + chunk - 1u) / chunk;
    const uint32_t lane = threadIdx.x & 63u;
    if (mine == 0u) return;
#pragma unroll 1
    for (;;) {
        const uint32_t t = q_take_ticket(heads + xcd * RMJ_Q_STRIDE);
        if (t >= mine * n_chunks) break;
        const uint32_t c = t / mine, quad = (t - c * mine) * 8u + xcd;
        if (c > 0u) (void)q_wait_for(done + quad, c);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        const uint32_t steps = n_steps - c * chunk < chunk ? n_steps - c * chunk : chunk;
        const uint32_t g = quad * 4u + (lane >> 4);
        const uint64_t gs_row = sm64(policy_seed + ((CEnv*)Ep)->game_offset + (uint64_t)g);
        const bool last_chunk = c + 1u == n_chunks;
#pragma unroll 1
        for (uint32_t it = 0; it < steps; it++)
            step4_call_enc<true, POL>(Ep, policy_seed, flags | ((last_chunk && it + 1u == steps) ? STEP_F_ALLROWS : STEP_F_QUIET), 0u, n_games, it == 0 ? 1u : 0u, gs_row, quad, out);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wave_sync();
        if (lane == 0u) __hip_atomic_store(done + quad, c + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
template <int POL>
__global__ __launch_bounds__(64, RMJ_STEP4_ENC_WAVES) void k_step4_fixup_enc(const Env* __restrict__ Ep, uint64_t policy_seed, uint32_t flags, uint32_t n_games,
                                                                                 uint32_t n_steps, const uint32_t* __restrict__ done, float* __restrict__ out) {
    if (uni(__hip_atomic_load(done + blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0u) return;
    const uint32_t g = blockIdx.x * 4u + ((threadIdx.x & 63u) >> 4);
    const uint64_t gs_row = sm64(policy_seed + ((CEnv*)Ep)->game_offset + (uint64_t)g);
#pragma unroll 1
    for (uint32_t it = 0; it < n_steps; it++)
        step4_call_enc<true, POL>(Ep, policy_seed, flags | (it + 1u < n_steps ? STEP_F_QUIET : STEP_F_ALLROWS), 0u, n_games, it == 0 ? 1u : 0u, gs_row, 0xFFFFFFFFu, out);
}

}  // namespace RMJ_NS
