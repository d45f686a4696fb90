// Shared (variant-independent) device definitions of the step path: environment slabs, per-wave LDS scratch,
// packed-action helpers (action.rs), small tile helpers.  Included once; rmj_step.hip.h is included once per variant.
#pragma once
#include "../../include/riichi_mi355x.h"
#include "rmj_hand.hip.h"
#include "rmj_eval4.hip.h"
#include "rmj_shanten.hip.h"
#include "rmj_state.h"

namespace rmj {

struct Env {
    GState* core;
    uint8_t* wall;        // [B][RMJ_WALL_STRIDE]
    uint64_t* legal;      // [B][4][64]
    uint8_t* nlegal;      // [B][4]
    uint8_t* mask;        // [B][4][82]
    uint64_t* waits;      // [B][4]
    uint32_t* status;     // [B]  active_mask | phase<<8 | done<<16
    RmjEvent* events;     // [B][ring]
    RmjWinResult* win;    // [B][4]  win_results of the round that ended the game (rare path only)
    uint32_t* wall_dg;    // [B][8]  RMJ_RULE_REFERENCE_RNG: wall_digest of a wall that is no longer in the wall slab (GState::wall_meta == 2)
    uint32_t ring_mask;   // ring-1
    uint32_t n_games;
    uint32_t rule_bits;
    uint32_t game_mode;
    uint32_t skip_log;
    uint32_t ctor_round_wind;
    uint64_t game_offset;
    uint32_t enc_stride;  // floats from one (game, seat) row of the base encoder's output to the next (rmj_set_encode_row_stride; default 74 x W)
    uint32_t pad_;
    ShantenTables sh;     // replacement-number tables (prefilter of the riichi probe)
};

// The Env record is immutable while a kernel runs.  Reading it through the CONSTANT address space tells the compiler so:
// its fields are fetched with scalar loads (s_load, cached, no vmcnt stall) even after the kernel has stored to global
// memory - through a generic `const Env&` every field access after a store was a vector global_load followed by
// s_waitcnt vmcnt(0), i.e. a full memory round trip (and a wait for all outstanding stores) at every event emission.
typedef const __attribute__((address_space(4))) Env CEnv;
__device__ __forceinline__ ShantenTables sh_tables_of(CEnv& E) {
    ShantenTables T;
    T.rank9 = E.sh.rank9; T.rank7 = E.sh.rank7; T.suit = E.sh.suit; T.honor = E.sh.honor; T.r2 = E.sh.r2; T.v6 = E.sh.v6;
    return T;
}

// Ticket lengths (in calls of the step function) of the fused ticket rollout, k_step4_queue: `cap` calls while much is left, a
// geometric descent (x 0.6) towards the call at which a quad is EXPECTED to finish (0.89 calls per step: the RandomAgent's inline
// responses advance a row 1.15 steps per call, the rows of a quad end within a call or two of each other), so that the tickets handed
// out last are one or two calls long and the launch does not end with a few waves on long tickets; behind that point 1, 2, 4, ... for
// the few quads that are not through yet, until the lengths cover n_steps calls (a call advances every unfinished row by at least one
// step).  `tail` = 0: equal tickets of `cap` calls.  Returns the number of tickets per quad; *len_out = the length of ticket `want`.
__host__ __device__ inline uint32_t q_ticket_plan(uint32_t n_steps, uint32_t cap, uint32_t tail, uint32_t want = 0xFFFFFFFFu, uint32_t* len_out = nullptr) {
    uint32_t n = 0, covered = 0;
    if (!tail) {
        while (covered < n_steps && n < 64u) { covered += cap; n++; }
        if (len_out) *len_out = cap;
        return n ? n : 1u;
    }
    uint32_t rem = (n_steps * 89u + 99u) / 100u, grow = 1u;
    while (covered < n_steps && n < 64u) {
        uint32_t s;
        if (rem > 0u) {
            s = (rem * 2u + 4u) / 5u;              // ceil(0.4 rem)
            if (s > cap) s = cap;
            if (s < 1u) s = 1u;
            rem -= s < rem ? s : rem;
        } else {
            s = grow > cap ? cap : grow;
            grow *= 2u;
        }
        if (n == want && len_out) *len_out = s;   // (the host keeps cap >= n_steps / 40: 64 tickets cover any rollout; the quad's last ticket runs unbounded anyway)
        covered += s;
        n++;
    }
    return n ? n : 1u;
}
__host__ __device__ inline uint32_t q_ticket_len(uint32_t n_steps, uint32_t cap, uint32_t c) {   // length of ticket c of the descending plan (scalar loop, <= 64 rounds)
    uint32_t len = cap;
    (void)q_ticket_plan(n_steps, cap, 1u, c, &len);
    return len;
}

// heavy-first launch order of the per-step kernel (k_step4<false>, rmj_step4.hip.h)
struct HeavyOrder {        // device pointers (rmj_env::d_heavy), nullptr members = plain order
    const uint32_t* in_cnt;
    const uint32_t* in_list;
    const uint8_t* in_flag;
    uint32_t* out_cnt;
    uint32_t* zero_cnt;   // the counter the launch AFTER the next one will fill: cleared here (no memset between the launches)
    uint32_t* out_list;
    uint8_t* out_flag;
    uint32_t front;
};

#define RMJ_EV_STAGE 4
struct WaveScratch {      // per-wave LDS scratch
    uint8_t tiles[144];
    uint8_t maskbuf[4 * 82 + 8];
    uint64_t legal[4][RMJ_MAX_LEGAL];  // lists produced this launch (copied to HBM by finalize_outputs)
    uint64_t wout[4];                  // waits produced this launch
    int nl[4];                         // list lengths produced this launch
    // events of the fast path, staged until the step has succeeded (k_step): a store in flight would make every later
    // s_waitcnt vmcnt(0) of the step - table lookups, the wall draw - wait for its acknowledgement as well
    alignas(16) uint32_t evbuf[RMJ_EV_STAGE][8];
    uint32_t evidx[RMJ_EV_STAGE];      // ring slot of each staged event
#ifdef RMJ_PROFILE
    uint64_t tprev;                    // section timer (profiling build only, scripts/prof_sections.py)
    uint32_t pacc[64];
#endif
#ifdef RMJ_TL4
    uint64_t tl_prev;                  // timeline build: cycles between the marks of the full path (TLF), per wave
    uint32_t tl_acc[16];
#endif
};

// Timeline build of k_step4 (-DRMJ_TL4, scripts/timeline4.py, never the shipped library): every wave of a launch owns a row of
// 32 u64: core cycles between the outer marks of the step [0..6], start / end on the 100 MHz clock [8], [9]
#ifdef RMJ_TL4
__device__ unsigned long long* g_tl4;
#define RMJ_TL4_ROW 32   /* u64 per wave: [0..15] as above, [16..31] cycles between the marks of the full path (TLF) */
#define TLF(c, k) do { const uint64_t t__ = __builtin_readcyclecounter(); \
        if ((c).lane == 0) { (c).X.tl_acc[k] += (uint32_t)(t__ - (c).X.tl_prev); (c).X.tl_prev = t__; } } while (0)
#else
#define TLF(c, k) do {} while (0)
#endif
#ifdef RMJ_RE_PROF   /* round-end timing build (scripts/r06_round_end_prof.py, never the shipped library): ticks of the 100 MHz clock summed by lane 0 of every wave:
                        [0] r4_round_end, [1] its calls, [2] step4_pass2, [3] its calls, [4] r4_round_end up to the wall loop, [5] the wall / deal / hand-sort / event loop, [6] games dealt */
__device__ unsigned long long g_re_prof[24];   // [8 + mode]: rows by R4_RE_* mode at the calls of step4_finish_rounds
#endif
#ifdef RMJ_DEBUG_HWID   /* value = blocks recorded; see k_step4_act_enc */
__device__ unsigned long long g_dbg_hwid[4 * RMJ_DEBUG_HWID];
#endif
#ifdef RMJ_QTL   /* ticket timeline build of k_step4_queue (-DRMJ_QTL, scripts/timeline_queue.py, never the shipped library) */
__device__ unsigned long long* g_qtl;
#define RMJ_QTL_ROW 256   /* u64 per wave: [0] kernel entry, [1] exit, [2] tickets, then 4 per ticket */
#endif
// Section timing of the step kernel (profiling build only: -DRMJ_PROFILE, never the shipped library): wave cycles
// between consecutive PROF marks are accumulated per section id by lane 0.
#ifdef RMJ_PROFILE
__device__ uint32_t* g_prof_buf;  // [n_games][64]: 0..31 cycles, 32..63 visit counts (each wave owns its row)
#define PROF(X, lane, id)                                                         \
    do {                                                                          \
        uint64_t t__ = __builtin_readcyclecounter();                              \
        if ((lane) == 0) {                                                        \
            (X).pacc[id] += (uint32_t)(t__ - (X).tprev);                          \
            (X).pacc[32 + (id)] += 1u;                                            \
        }                                                                         \
        (X).tprev = t__;                                                          \
    } while (0)
#define PROF_START(X, lane) do { (X).pacc[lane] = 0u; (X).tprev = __builtin_readcyclecounter(); } while (0)
#define PROF_FLUSH(X, lane, g) do { rmj::g_prof_buf[(size_t)(g) * 64 + (lane)] += (X).pacc[lane]; } while (0)
#elif defined(RMJ_CUTS)
// Instruction accounting build (scripts/valu_sections.py, never the shipped library): a wave that reaches mark `g_cut`
// ends there, before anything is stored.  The difference of the launch's PMC instruction counters between two cuts is
// the number of instructions executed between the two marks.
__device__ int g_cut = -1, g_cut2 = -1, g_cut3 = -1;
__device__ uint32_t g_bail_reason[32];  // bail census of k_step4 (R4BAIL sites)
#define PROF(X, lane, id) do { if (rmj::g_cut == (id) || rmj::g_cut2 == (id) || rmj::g_cut3 == (id)) asm volatile("s_endpgm" ::: "memory"); } while (0)   /* (asm, not the noreturn builtin: see R4M) */
#define PROF_START(X, lane) do {} while (0)
#define PROF_FLUSH(X, lane, g) do {} while (0)
#else
#ifdef RMJ_CENSUS
__device__ uint32_t g_bail_reason[32];  // bail census of k_step4 (R4BAIL sites) without the accounting marks
#endif
#define PROF(X, lane, id) do {} while (0)
#define PROF_START(X, lane) do {} while (0)
#define PROF_FLUSH(X, lane, g) do {} while (0)
#endif

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ uint64_t lanemask_lt(int lane) { return (1ull << lane) - 1ull; }
}  // namespace rmj
#include "rmj_refrng.hip.h"
namespace rmj {

// ---------------------------------------------------------------- packed actions
__device__ __forceinline__ uint64_t mk_action(uint32_t type, uint32_t tile, uint32_t n, uint32_t c0 = 0, uint32_t c1 = 0, uint32_t c2 = 0,
                                              uint32_t c3 = 0) {
    return (uint64_t)type | ((uint64_t)tile << 8) | ((uint64_t)n << 16) | ((uint64_t)c0 << 24) | ((uint64_t)c1 << 32) |
           ((uint64_t)c2 << 40) | ((uint64_t)c3 << 48);
}
__device__ __forceinline__ uint32_t a_type(uint64_t a) { return (uint32_t)(a & 0xFF); }
__device__ __forceinline__ uint32_t a_tile(uint64_t a) { return (uint32_t)((a >> 8) & 0xFF); }
__device__ __forceinline__ uint32_t a_n(uint64_t a) { return (uint32_t)((a >> 16) & 0xFF); }
__device__ __forceinline__ uint32_t a_c(uint64_t a, int i) { return (uint32_t)((a >> (24 + 8 * i)) & 0xFF); }
// canonical form: consume tiles ascending (Action::new, action.rs:97-98), unused bytes zero
__device__ inline uint64_t a_canon(uint64_t a) {
    if (a == RMJ_NO_ACTION) return a;
    uint32_t n = a_n(a);
    if (n > 4) n = 4;
    uint32_t c[4];
#pragma unroll
    for (int i = 0; i < 4; i++) c[i] = (uint32_t)i < n ? a_c(a, i) : 0xFFFFu;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            if (c[j] > c[j + 1]) { uint32_t t = c[j]; c[j] = c[j + 1]; c[j + 1] = t; }
#pragma unroll
    for (int i = 0; i < 4; i++)
        if ((uint32_t)i >= n) c[i] = 0;
    return mk_action(a_type(a), a_tile(a), n, c[0], c[1], c[2], c[3]);
}
// action.rs:158-227
__device__ inline int a_encode(uint64_t a) {
    uint32_t ty = a_type(a), tile = a_tile(a);
    switch (ty) {
        case RMJ_DISCARD: return tile / 4;
        case RMJ_RIICHI: return 37;
        case RMJ_CHI: {
            uint32_t tt = tile / 4, x = a_c(a, 0) / 4, y = a_c(a, 1) / 4;
            uint32_t lo = min(tt, min(x, y)), hi = max(tt, max(x, y));
            return tt == lo ? 38 : (tt == hi ? 40 : 39);
        }
        case RMJ_PON: return 41;
        case RMJ_DAIMINKAN: return 42 + tile / 4;
        case RMJ_ANKAN:
        case RMJ_KAKAN: return 42 + a_c(a, 0) / 4;
        case RMJ_RON:
        case RMJ_TSUMO: return 79;
        case RMJ_KYUSHU: return 80;
        case RMJ_PASS: return 81;
        default: return -1;
    }
}
// action.rs:262-346 (3P compact ids, 60-wide)
__device__ inline int a_encode_3p(uint64_t a) {
    uint32_t ty = a_type(a), tile = a_tile(a);
    auto compact = [](uint32_t t34) -> int { return t34 == 0 ? 0 : (t34 == 8 ? 1 : (t34 >= 9 && t34 < 34 ? (int)t34 - 7 : -1)); };
    switch (ty) {
        case RMJ_DISCARD: return compact(tile / 4);
        case RMJ_RIICHI: return 27;
        case RMJ_PON: return 28;
        case RMJ_DAIMINKAN: { int k = compact(tile / 4); return k < 0 ? -1 : 29 + k; }
        case RMJ_ANKAN:
        case RMJ_KAKAN: { int k = compact(a_c(a, 0) / 4); return k < 0 ? -1 : 29 + k; }
        case RMJ_RON:
        case RMJ_TSUMO: return 56;
        case RMJ_KYUSHU: return 57;
        case RMJ_PASS: return 58;
        case RMJ_KITA: return 59;
        default: return -1;
    }
}
// validation match, state/mod.rs:344-393 (quirk Q13)
__device__ inline bool a_match(uint64_t l, uint64_t act) {
    uint32_t lt = a_type(l);
    if (lt != a_type(act)) return false;
    bool tiles_match = a_tile(l) == a_tile(act);
    bool cons_match = (l >> 16) == (act >> 16);
    bool act_empty = a_n(act) == 0;
    if (tiles_match) {
        if (cons_match) return true;
        if (act_empty && lt == RMJ_KAKAN) return true;
        if (act_empty && (lt == RMJ_DISCARD || lt == RMJ_RIICHI || lt == RMJ_TSUMO || lt == RMJ_RON || lt == RMJ_PASS)) return true;
    }
    if (cons_match && (lt == RMJ_ANKAN || lt == RMJ_KAKAN)) return true;
    if (a_tile(act) == RMJ_TILE_NONE) return lt == RMJ_TSUMO || lt == RMJ_RON || lt == RMJ_RIICHI || lt == RMJ_KYUSHU || lt == RMJ_KITA;
    return false;
}

__device__ __forceinline__ uint64_t sm64(uint64_t x) {  // state/wall.rs:83-88
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// The device policies' key of (game, step, seat) - round 6 - and what they draw from it.  gs = splitmix64(policy_seed + global game), once per
// game and call; per seat and step a 32-bit finaliser (murmur3's fmix32) over gs's halves and the counter, and a multiply-high instead of a
// modulo: 13 vector instructions per pick where splitmix64 + a four-digit modulo by magic numbers took ~55 (the RandomAgent's pick was 112
// vector cycles at 17 live lanes: profiles/r05_lane_use_random.txt).  The policy is this build's own definition (the reference's RandomAgent
// is Python's random.choice); the oracle's twin: oracle_capi.cpp policy_key32 / policy_choice.
__host__ __device__ __forceinline__ uint32_t policy_key32(uint64_t gs, uint32_t step, uint32_t seat) {
    uint32_t x = ((uint32_t)gs ^ ((step * 4u + seat) * 0x9E3779B1u)) + (uint32_t)(gs >> 32);
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}
__host__ __device__ __forceinline__ uint32_t policy_pick(uint32_t key, uint32_t n) { return (uint32_t)(((uint64_t)key * (uint64_t)n) >> 32); }   // uniform over [0, n)
__host__ __device__ __forceinline__ bool policy_calls(uint32_t key, uint32_t call_rate_256) { return (key >> 24) < call_rate_256; }                // the greedy policy's Pon / Chi coin
__host__ __device__ __forceinline__ uint32_t policy_tie(uint32_t key, uint32_t n) { return policy_pick(key * 0x9E3779B1u, n); }                      // ... and its tie break

// x mod n for 1 <= n <= 64, exact, without the 64-bit software division: four 16-bit digits, each step reduces a value
// below 2^22 with an fp32 reciprocal estimate (quotient off by at most one) and one correction either way.
__device__ __forceinline__ uint32_t mod_small(uint64_t x, uint32_t n) {
    const float inv = __builtin_amdgcn_rcpf((float)n);
    int r = 0;
#pragma unroll
    for (int d = 3; d >= 0; d--) {
        int v = (r << 16) | (int)((x >> (16 * d)) & 0xFFFFull);
        int q = (int)((float)v * inv);
        int rem = v - q * (int)n;
        rem = rem < 0 ? rem + (int)n : rem;
        rem = rem >= (int)n ? rem - (int)n : rem;
        r = rem;
    }
    return (uint32_t)r;
}

// Per-lane variant by magic multiplication: digits of 16 bits keep every intermediate below 2^22, where
// floor(v / n) = mulhi(v, ceil(2^32 / n)) exactly (error term v * (M n - 2^32) < 2^22 * 64 < 2^32); M from the table.
__device__ __forceinline__ uint32_t mod_small_magic(uint64_t x, uint32_t n);
// The same for wave-uniform operands, on the scalar unit only (the step is bound by VALU issue): with a_i = 2^(16 i) mod n
// the digit sum d0 + d1 a1 + d2 a2 + d3 a3 is below 2^24 and congruent to x, and for v < 2^24, n <= 64 the quotient is
// exactly mulhi(v, ceil(2^32 / n)) (error term v * (M n - 2^32) < 2^24 * 64 < 2^32).  Both tables are compile-time data
// in the constant address space: scalar loads.
struct ModTab { uint32_t pw[65]; uint32_t inv[65]; };
constexpr ModTab make_modtab() {
    ModTab t{};
    for (uint32_t n = 1; n <= 64; n++) {
        const uint64_t a1 = 65536ull % n, a2 = (a1 * 65536ull) % n, a3 = (a2 * 65536ull) % n;
        t.pw[n] = (uint32_t)(a1 | (a2 << 8) | (a3 << 16));
        t.inv[n] = n == 1 ? 0u : (uint32_t)((0x100000000ull + n - 1) / n);
    }
    return t;
}
__constant__ const ModTab g_modtab = make_modtab();
__device__ __forceinline__ uint32_t mod_small_magic(uint64_t x, uint32_t n) {  // 1 <= n <= 64, per lane
    const uint32_t inv = g_modtab.inv[n];
    uint32_t r = 0;
#pragma unroll
    for (int d = 3; d >= 0; d--) {
        const uint32_t v = (r << 16) | (uint32_t)((x >> (16 * d)) & 0xFFFFull);
        const uint32_t qd = __umulhi(v, inv);
        r = v - qd * n;
    }
    return n == 1u ? 0u : r;
}
__device__ __forceinline__ uint32_t mod_small_uniform(uint64_t x, uint32_t n) {  // x, n in SGPRs; 1 <= n <= 64
    const uint32_t pw = g_modtab.pw[n], inv = g_modtab.inv[n];
    const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    const uint32_t v = (lo & 0xFFFFu) + (lo >> 16) * (pw & 0xFFu) + (hi & 0xFFFFu) * ((pw >> 8) & 0xFFu) + (hi >> 16) * (pw >> 16);
    const uint32_t q = (uint32_t)(((uint64_t)v * inv) >> 32);
    return n == 1 ? 0u : v - q * n;
}

// ---------------------------------------------------------------- small helpers
__device__ __forceinline__ bool is_terminal_tile136(int t) {  // types.rs:362-367
    int tt = t / 4;
    return tt >= 27 || (tt % 9) == 0 || (tt % 9) == 8;
}
__device__ __forceinline__ int next_dora34(int t, bool sanma) {  // hand_evaluator.rs:286-300 / _3p.rs:300-311
    if (sanma) {
        if (t == 0) return 8;
        if (t == 8) return 0;
        if (t < 9) return t;
    }
    if (t < 27) return (t % 9 == 8) ? t - 8 : t + 1;
    if (t < 31) return t == 30 ? 27 : t + 1;
    return t == 33 ? 31 : t + 1;
}

// concealed histogram of a seat (optionally skipping hand index `skip`)
__device__ __forceinline__ PH build_ph(const PState& P, int skip = -1) {
    PH h = {0, 0, 0, 0};
    int n = P.hand_len;
    for (int j = 0; j < n; j++)
        if (j != skip) ph_add(h, P.hand[j] >> 2);
    return h;
}
// Wave-cooperative build_ph: lane j contributes the one-hot field of hand[j]; four row sums; the result is
// wave-uniform (scalar registers).  Must be called by all 64 lanes.
__device__ __forceinline__ PH build_ph_wave(const PState& P, int lane, int skip = -1) {
    const int n = P.hand_len;
    uint32_t a = 0, b = 0, c = 0, d = 0;
    if (lane < n && lane != skip) {
        int t = P.hand[lane] >> 2;
        int s = t_suit(t);
        uint32_t one = 1u << (3 * (t - 9 * s));
        a = s == 0 ? one : 0u; b = s == 1 ? one : 0u; c = s == 2 ? one : 0u; d = s == 3 ? one : 0u;
    }
    a = row_sum16(a); b = row_sum16(b); c = row_sum16(c); d = row_sum16(d);
    PH h;
    h.a = (uint32_t)__builtin_amdgcn_readlane((int)a, 15); h.b = (uint32_t)__builtin_amdgcn_readlane((int)b, 15);
    h.c = (uint32_t)__builtin_amdgcn_readlane((int)c, 15); h.d = (uint32_t)__builtin_amdgcn_readlane((int)d, 15);
    return h;
}
__device__ inline MeldAgg build_meld_agg(const PState& P) {
    MeldAgg m;
    m.n = P.n_melds;
    m.n_kan = m.n_ankan = m.n_nonchi = 0;
    m.menzen = true;
    m.types = 0;
    m.fu = 0;
    m.aka = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        m.mtypes[i] = 0;
        m.mtype[i] = 0;
        m.t0[i] = 0;
        if (i < m.n) {
            uint8_t ty = P.meld_type[i];
            int nt = (ty >= RMJ_MELD_DAIMINKAN) ? 4 : 3;
            uint64_t mm = 0;
            for (int k = 0; k < nt; k++) {
                int t = P.meld_tiles[i][k];
                mm |= 1ull << (t >> 2);
                m.aka += is_aka(t);
            }
            m.mtypes[i] = mm;
            m.types |= mm;
            m.mtype[i] = ty;
            int t0 = P.meld_tiles[i][0] >> 2;  // tiles sorted by id -> lowest type first (== chi sort)
            m.t0[i] = (uint8_t)t0;
            bool opened = ty != RMJ_MELD_ANKAN;
            if (opened) m.menzen = false;
            bool kan = ty >= RMJ_MELD_DAIMINKAN;
            m.n_kan += kan;
            m.n_ankan += (ty == RMJ_MELD_ANKAN);
            m.n_nonchi += (ty != RMJ_MELD_CHI);
            if (ty != RMJ_MELD_CHI) {  // tiles[0] == tiles[1]
                int f = opened ? 2 : 4;
                if (t_is_terminal(t0)) f *= 2;
                if (kan) f *= 4;
                m.fu += f;
            }
        }
    }
    return m;
}

}  // namespace rmj
