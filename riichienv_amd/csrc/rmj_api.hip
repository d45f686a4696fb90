// Kernels + C-ABI (include/riichi_mi355x.h) of the MI355X-native batched Riichi step path.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC rmj_api.hip -o libriichi_mi355x.so
//
// Launch geometry: 256-thread workgroups = 4 wavefronts = 4 games; grid = ceil(B/4) (16 384
// workgroups at B = 65 536, i.e. 64 per CU: far above the ">>256 workgroups" rule).  The
// blockIdx -> game map is launch-invariant, so a game is always served by the same XCD
// (block b runs on XCD b % 8) and its 640-byte record is re-read from that XCD's L2 / MALL.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/riichi_mi355x.h"
#include "../../include/riichi_mi355x_bench.h"
#include "rmj_common.hip.h"
#include "rmj_eval4.hip.h"
#include "rmj_encode.hip.h"
#include "rmj_seq.hip.h"
#include "rmj_host.h"

using namespace rmj;

#define WPB 4
#define STEP_F_RANDOM 1u
#define STEP_F_AUTORESET 2u
#define STEP_F_IDS 4u /* `actions` holds int32 action ids [n][4] (Observation.find_action semantics) */
#define STEP_F_QUIET 0x10000u   /* fused rollouts, every step but the last: no mask rows, no nlegal / waits / status words (nobody can read them) */
#define STEP_F_ALLROWS 0x20000u /* fused rollouts, last step: all four mask rows are rewritten (the quiet steps left them stale) */
#define STEP_F_CONT_RYU 0x40000u /* ol_step_full: continue at the exhaustive draw on the record k_step4's tier 0 left in LDS (no reload, no replay of the discard) */
#define STEP_F_CONT_FIN 0x80000u /* ol_step_full: the step is complete on the record in LDS, only the observation outputs are produced */
#define STEP_F_CONT_CLAIMS 0x400000u /* ol_step_full: continue behind the dahai event of the discard made on the record in LDS (claim generation, then the rest of _resolve_discard) */
#define STEP_F_GREEDY 8u /* with STEP_F_RANDOM: the greedy policy (rmj_step_greedy, r4_policy_greedy) instead of the RandomAgent; bits 8..15 = call rate / 256 */

template <int N>
struct BlockSharedT {
    GState st[N];
    WaveScratch x[N];
};
typedef BlockSharedT<WPB> BlockShared;
#ifndef RMJ_STEP_WPB
#define RMJ_STEP_WPB 1 /* games (= waves) per block of the step kernel: single-wave blocks release their LDS as soon as the game is done (a block of four waited for its slowest game) */
#endif
static inline dim3 step_grid(uint32_t n) { return dim3((n + RMJ_STEP_WPB - 1) / RMJ_STEP_WPB); }
// smallest batch that a multi-step device rollout splits over several streams of a handle (rmj_step_random)
#define RMJ_SPLIT_MIN_GAMES 16384u
#define RMJ_SPLIT_MIN_PART 8192u   // games per part at least
#define RMJ_MAX_ROLLOUT_STREAMS 8
// games per wave by batch size (STEP_F_ROWS_SHIFT; profiles/r04_rows_sweep.txt, fused 4p-red-single rollouts, M env.step/s at 4 | 2 | 1 games
// per wave: 2 048 games 226 | 254 | 278, 4 096: 435 | 478 | 456, 8 192: 814 | 770 | 522, 16 384: 1 288 | 882 | 599; round 5, profiles/r05_rows_sweep.txt:
// 2 048: 236 | 272 | 301, 3 072: 345 | 373 | 416, 4 096: 455 | 505 | 492, 6 144: 632 | 692 | 589, 8 192: 845 | 823 | 567)
#define RMJ_ROWS1_MAX_GAMES 3584u
#define RMJ_ROWS2_MAX_GAMES 7168u

__device__ __forceinline__ void load_state(GState& S, const GState* src, int lane) {
    if (lane < (int)(sizeof(GState) / 16)) reinterpret_cast<uint4*>(&S)[lane] = reinterpret_cast<const uint4*>(src)[lane];
    wave_sync();
}
__device__ __forceinline__ void store_state(const GState& S, GState* dst, int lane) {
    wave_sync();
    if (lane < (int)(sizeof(GState) / 16)) reinterpret_cast<uint4*>(dst)[lane] = reinterpret_cast<const uint4*>(&S)[lane];
}

// fast path of k_step: the 128 B of globals and the PState quarters named by `dirty` (bit = seat)
__device__ __forceinline__ void store_state_partial(const GState& S, GState* dst, int lane, uint32_t dirty) {
    wave_sync();
    if (lane < (int)(sizeof(GState) / 16) && (lane >= 32 || ((dirty >> (lane >> 3)) & 1u)))
        reinterpret_cast<uint4*>(dst)[lane] = reinterpret_cast<const uint4*>(&S)[lane];
}

// Device policy without stepping (rmj_random_actions)
__global__ void k_random_actions(Env E, uint64_t policy_seed, uint64_t* out) {
    uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= E.n_games) return;
    const GState& S = E.core[g];
    uint64_t gs = sm64(policy_seed + E.game_offset + g);
    for (int p = 0; p < 4; p++) {
        uint64_t a = RMJ_NO_ACTION;
        int n = E.nlegal[(size_t)g * 4 + p];
        if (((S.active_mask >> p) & 1u) && n > 0 && !S.is_done) {
            const uint32_t ch = policy_pick(policy_key32(gs, S.step_count, (uint32_t)p), (uint32_t)n);
            a = E.legal[((size_t)g * 4 + p) * RMJ_MAX_LEGAL + ch];
        }
        out[(size_t)g * 4 + p] = a;
    }
}

// Trainer-side masked categorical sampler (rmj_sample_ids_device): one wave per game; for every seat that is to act the
// lanes hold ids lane and lane + 64 of the seat's mask row, add Gumbel noise to the policy's logits (Gumbel-max = a draw
// from softmax(logits) restricted to the legal ids; no logits = uniform over the legal ids) and a wave arg-max picks the id.
// The noise is counter-based: splitmix64(seed, global game, the game's step count, seat, id).
// Round 5: four games per wave (one 16-lane row each; lane r of a row judges the ids r, r + 16, ...), like the step kernels - the keyed Gumbel
// draw of an id costs the same wherever it runs, but a wave per game left 64 lanes to 82 ids of (mostly) one seat.  The same keys, the same
// arg-max rule (ties to the lower id) as the wave-per-game kernel of rounds 3-4: identical ids.
// One id per acting seat of the row's game g (in: the row has a game): lane p of the row returns seat p's id, -1 where nobody acts.
__device__ __forceinline__ int32_t sample_ids_row(const uint32_t* status, const GState* core, const uint8_t* nlegal, const uint8_t* mask, uint64_t game_offset,
                                                  int game_mode, uint32_t g, bool in, const float* __restrict__ logits, uint32_t stride, uint64_t seed, int lane) {
    const int r = lane & 15;
    const uint32_t gi = in ? g : 0u;
    const uint32_t st = in ? status[gi] : 0x10000u;
    const uint32_t am = (st >> 16) & 0xFFu ? 0u : (st & 0xFu);   // done games have nobody to act
    const int A = game_mode >= 3 ? RMJ_ACTION_SPACE_3P : RMJ_ACTION_SPACE_4P;
    const uint64_t base = sm64(seed ^ sm64(game_offset + gi)) + ((uint64_t)core[gi].step_count << 10);
    const uint32_t nl4 = in ? *reinterpret_cast<const uint32_t*>(nlegal + (size_t)gi * 4) : 0u;   // the four list lengths of the game
    int32_t res = -1;
    for (int p = 0; p < 4; p++) {
        const bool act = ((am >> p) & 1u) && ((nl4 >> (8 * p)) & 0xFFu) != 0u;   // (row-uniform)
        if (!__ballot(act)) continue;
        const uint8_t* m = mask + ((size_t)gi * 4 + p) * 82;
        const float* lg = logits ? logits + ((size_t)gi * 4 + p) * stride : nullptr;
        float best = -INFINITY;
        int bid = -1;
        if (act) {
            for (int id = r; id < A; id += 16) {
                if (m[id]) {
                    const uint64_t h = sm64(base + ((uint64_t)p << 8) + (uint64_t)id);
                    const float u = ((float)(uint32_t)(h >> 40) + 0.5f) * (1.0f / 16777216.0f);   // (0, 1), 24 bits
                    const float key = (lg ? lg[id] : 0.0f) - __logf(-__logf(u));
                    if (key > best || bid < 0) { best = key; bid = id; }
                }
            }
        }
        // row arg-max (ties to the lower id)
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) {
            const float ob = __shfl_xor(best, off, 64);
            const int oi = __shfl_xor(bid, off, 64);
            if (oi >= 0 && (bid < 0 || ob > best || (ob == best && oi < bid))) { best = ob; bid = oi; }
        }
        if (act && r == p) res = bid;
    }
    return res;
}
__global__ __launch_bounds__(256) void k_sample_ids(Env E, const float* __restrict__ logits, uint32_t stride, uint64_t seed,
                                                    int32_t* __restrict__ out) {
    const int lane = threadIdx.x & 63, r = lane & 15;
    const uint32_t g = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4 + (uint32_t)(lane >> 4);
    const bool in = g < E.n_games;
    const int32_t res = sample_ids_row(E.status, E.core, E.nlegal, E.mask, E.game_offset, E.game_mode, g, in, logits, stride, seed, lane);
    if (in && r < 4) out[(size_t)g * 4 + r] = res;
}

struct ResetArgs {
    const uint8_t* select;
    const uint8_t* walls;       // [n][136] reference orientation (draw order)
    const uint8_t* oya;
    const uint8_t* round_wind;
    const int32_t* scores;      // [n][4]
    const uint8_t* honba;
    const uint32_t* kyotaku;
    const uint64_t* seeds;      // ctor only
    uint64_t base_seed;
    uint32_t is_ctor;
};


#define RMJ_NS rmj4
#define RMJ_SANMA 0
#include "rmj_step.hip.h"
#include "rmj_kernels.hip.h"
#include "rmj_step4.hip.h"
#undef RMJ_NS
#undef RMJ_SANMA
#define RMJ_NS rmj3
#define RMJ_SANMA 1
#include "rmj_step.hip.h"
#include "rmj_kernels.hip.h"
#include "rmj_step4.hip.h"
#undef RMJ_NS
#undef RMJ_SANMA

// largest raw HW_REG_XCC_ID[3:0] over the waves of the launch (rmj_create: is the per-XCD queue assumption of k_step4_queue valid?)
__global__ void k_probe_xcc(unsigned long long* out) {
    const unsigned long long id = (unsigned long long)((uint32_t)__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u);
    if ((threadIdx.x & 63u) == 0u) atomicMax(out, id);
}
__global__ void k_sum_steps(const GState* core, uint32_t n, unsigned long long* out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long v = 0;
    for (; i < n; i += gridDim.x * blockDim.x) v += core[i].step_count;
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(out, v);
}
__global__ void k_sum_full(const GState* core, uint32_t n, unsigned long long* out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long v = 0;
    for (; i < n; i += gridDim.x * blockDim.x) v += core[i].full_count;
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(out, v);
}
__global__ void k_gather_steps(const GState* core, uint32_t n, uint64_t* out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = core[i].step_count;
}
// ---- compact legal lists for a host agent loop (rmj_get_legal_compact): rows = the seats that are to act, in (game, seat) order,
// entries = their lists one after the other.  Pass 1: per game the rows / entries it contributes, block prefix sums; pass 2: the
// block totals scanned by one block; pass 3: every game writes its rows.  Deterministic order, no atomics.
#define LC_BLOCK 256
__device__ __forceinline__ uint32_t lc_rows_of(uint32_t status, const uint8_t* nl, uint32_t& entries) {
    const uint32_t am = (status >> 16) & 1u ? 0u : (status & 0xFu);
    uint32_t rows = 0;
    entries = 0;
    for (int p = 0; p < 4; p++)
        if (((am >> p) & 1u) && nl[p]) { rows++; entries += nl[p]; }
    return rows;
}
__global__ __launch_bounds__(LC_BLOCK) void k_lc_count(const uint32_t* __restrict__ status, const uint8_t* __restrict__ nlegal, uint32_t n, uint32_t* __restrict__ pre /*[n][2]*/,
                                                       uint32_t* __restrict__ blk /*[blocks][2]*/) {
    __shared__ uint32_t sr[LC_BLOCK], se[LC_BLOCK];
    const uint32_t g = blockIdx.x * LC_BLOCK + threadIdx.x;
    uint32_t e = 0, r = 0;
    if (g < n) r = lc_rows_of(status[g], nlegal + (size_t)g * 4, e);
    sr[threadIdx.x] = r; se[threadIdx.x] = e;
    __syncthreads();
    for (int off = 1; off < LC_BLOCK; off <<= 1) {     // inclusive Hillis-Steele scan
        uint32_t ar = 0, ae = 0;
        if ((int)threadIdx.x >= off) { ar = sr[threadIdx.x - off]; ae = se[threadIdx.x - off]; }
        __syncthreads();
        sr[threadIdx.x] += ar; se[threadIdx.x] += ae;
        __syncthreads();
    }
    if (g < n) { pre[2 * (size_t)g] = sr[threadIdx.x] - r; pre[2 * (size_t)g + 1] = se[threadIdx.x] - e; }
    if (threadIdx.x == LC_BLOCK - 1) { blk[2 * blockIdx.x] = sr[threadIdx.x]; blk[2 * blockIdx.x + 1] = se[threadIdx.x]; }
}
__global__ void k_lc_scan(uint32_t* blk, uint32_t blocks, uint32_t* totals /*[2]*/) {   // one thread: a few thousand blocks at most
    if (blockIdx.x || threadIdx.x) return;
    uint32_t r = 0, e = 0;
    for (uint32_t b = 0; b < blocks; b++) {
        const uint32_t cr = blk[2 * b], ce = blk[2 * b + 1];
        blk[2 * b] = r; blk[2 * b + 1] = e;
        r += cr; e += ce;
    }
    totals[0] = r; totals[1] = e;
}
__global__ __launch_bounds__(LC_BLOCK) void k_lc_gather(const uint32_t* __restrict__ status, const uint8_t* __restrict__ nlegal, const uint64_t* __restrict__ legal, uint32_t n,
                                                        const uint32_t* __restrict__ pre, const uint32_t* __restrict__ blk, uint32_t cap_rows, uint32_t cap_entries,
                                                        uint32_t* __restrict__ index, uint32_t* __restrict__ offs, uint64_t* __restrict__ entries) {
    const uint32_t g = blockIdx.x * LC_BLOCK + threadIdx.x;
    if (g >= n) return;
    const uint32_t st = status[g];
    const uint32_t am = (st >> 16) & 1u ? 0u : (st & 0xFu);
    uint32_t row = blk[2 * blockIdx.x] + pre[2 * (size_t)g], ent = blk[2 * blockIdx.x + 1] + pre[2 * (size_t)g + 1];
    for (int p = 0; p < 4; p++) {
        const uint32_t k = nlegal[(size_t)g * 4 + p];
        if (!((am >> p) & 1u) || !k) continue;
        if (row < cap_rows) { index[row] = g * 4u + (uint32_t)p; offs[row] = ent; offs[row + 1] = ent + k; }   // (the next row writes the same value at row + 1)
        for (uint32_t j = 0; j < k; j++)
            if (ent + j < cap_entries) entries[ent + j] = legal[((size_t)g * 4 + p) * RMJ_MAX_LEGAL + j];
        row++;
        ent += k;
    }
}
// RiichiEnv.points (env.rs:691-727) with ranks (env.rs:673-689: by score, ties by seat) for every game: f64 like the reference
__global__ void k_points(const GState* core, uint32_t n, int np, double weight, double base, double u0, double u1, double u2, double u3, double* out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int32_t sc[4];
    for (int p = 0; p < 4; p++) sc[p] = core[i].p[p].score;
    for (int p = 0; p < 4; p++) {
        double v = 0.0;
        if (p < np) {
            int rank = 0;   // seats ahead: a higher score, or the same score and a lower seat index
            for (int o = 0; o < np; o++) rank += (sc[o] > sc[p]) || (sc[o] == sc[p] && o < p);
            const double uma = rank == 0 ? u0 : (rank == 1 ? u1 : (rank == 2 ? u2 : u3));
            v = ((double)sc[p] - base) / 1000.0 * weight + uma;
        }
        out[(size_t)i * 4 + p] = v;
    }
}
__global__ void k_gather_scores(const GState* core, uint32_t n, int32_t* out, uint32_t* evc) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        for (int p = 0; p < 4; p++) out[(size_t)i * 4 + p] = core[i].p[p].score;
        if (evc) evc[i] = core[i].ev_count - core[i].ev_base;   // len(mjai_log) of the current game
    }
}
__global__ void k_track_mark(uint8_t* mark, const uint8_t* __restrict__ select, uint32_t first, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && (!select || select[first + i])) mark[first + i] = 1;
}
__global__ void k_log_positions(const GState* core, uint32_t n, uint32_t* base, uint32_t* pos) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { base[i] = core[i].ev_base; pos[i] = core[i].ev_count; }
}

// WallState.salt / wall_digest (state/wall.rs:15-16, 48-55) of games [first, first + n): out[i] = {valid, salt (u64), SHA-256 (8 x u32, big
// endian words)} as 11 dwords; one lane per game.  valid = GState::wall_meta (RMJ_RULE_REFERENCE_RNG shuffles only).
__global__ __launch_bounds__(64) void k_wall_digest(const GState* __restrict__ core, const uint8_t* __restrict__ wall, const uint32_t* __restrict__ frozen,
                                                    uint32_t first, uint32_t n, int tiles, uint32_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t g = first + i;
    const uint8_t* W = wall + (size_t)g * RMJ_WALL_STRIDE;
    uint32_t* o = out + (size_t)i * 11;
    const uint32_t valid = core[g].wall_meta;
    uint64_t salt = 0;
    uint32_t dg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (valid) {
        for (int k = 0; k < 8; k++) salt |= (uint64_t)W[136 + k] << (8 * k);
        if (valid == 2) { for (int k = 0; k < 8; k++) dg[k] = frozen[(size_t)g * 8 + k]; }   // the wall it belonged to is gone
        else sha256_wall(W, tiles, salt, dg);
    }
    o[0] = valid; o[1] = (uint32_t)salt; o[2] = (uint32_t)(salt >> 32);
    for (int k = 0; k < 8; k++) o[3 + k] = dg[k];
}

// ---------------------------------------------------------------- batched hand math kernels (one wave per case)
__device__ inline MeldAgg agg_from_views(const RmjHandCase& hc) {
    MeldAgg m;
    m.n = hc.n_melds > 4 ? 4 : hc.n_melds;
    m.n_kan = m.n_ankan = m.n_nonchi = 0;
    m.menzen = true;
    m.types = 0;
    m.fu = 0;
    m.aka = 0;
    for (int i = 0; i < 4; i++) {
        m.mtypes[i] = 0;
        m.mtype[i] = 0;
        m.t0[i] = 0;
        if (i < m.n) {
            const RmjMeldView& v = hc.melds[i];
            int nt = v.n_tiles > 4 ? 4 : v.n_tiles;
            uint64_t mm = 0;
            int tmin = 99;
            for (int k = 0; k < nt; k++) {
                int t = v.tiles[k];
                mm |= 1ull << (t >> 2);
                m.aka += is_aka(t);
                tmin = min(tmin, t >> 2);
            }
            m.mtypes[i] = mm;
            m.types |= mm;
            m.mtype[i] = v.meld_type;
            int t0 = (v.meld_type == RMJ_MELD_CHI) ? tmin : (v.tiles[0] >> 2);  // chi tiles are sorted (hand_evaluator.rs:63-65)
            m.t0[i] = (uint8_t)t0;
            if (v.opened) m.menzen = false;
            bool kan = v.meld_type >= RMJ_MELD_DAIMINKAN;
            m.n_kan += kan;
            m.n_ankan += (v.meld_type == RMJ_MELD_ANKAN);
            m.n_nonchi += (v.meld_type != RMJ_MELD_CHI);
            bool trip = nt >= 3 && v.meld_type != RMJ_MELD_CHI && (v.tiles[0] >> 2) == (v.tiles[1] >> 2);
            if (v.meld_type == RMJ_MELD_CHI && nt >= 3) {  // sorted types: equal first two only for degenerate input
                int a = 99, b = 99;
                for (int k = 0; k < nt; k++) {
                    int t = v.tiles[k] >> 2;
                    if (t < a) { b = a; a = t; } else if (t < b) b = t;
                }
                trip = a == b;
            }
            if (trip) {
                int f = v.opened ? 2 : 4;
                if (t_is_terminal(t0)) f *= 2;
                if (kan) f *= 4;
                m.fu += f;
            }
        }
    }
    return m;
}

// Round 4: HandEvaluator::calc + waits for FOUR hands per wave, one 16-lane row per hand (e4_calc, rmj_eval4.hip.h).  The wave's four
// 88-byte cases arrive as one contiguous 352-byte block (coalesced dword loads into LDS), lane r of a row is tile r / meld r / dora
// indicator r while the case is parsed, the 64-byte results leave as one dword per lane (256 contiguous bytes per wave).  Round 3's
// kernel - one wave per hand, lane = candidate head walking every division x winning group serially, 236 registers squeezed into 80
// with 576 B of scratch per lane - ran at 71 M hands/s.
struct EvalShared {
    alignas(16) uint32_t in[4 * sizeof(RmjHandCase) / 4];
    alignas(16) uint32_t out[4][16];
};
static_assert(sizeof(RmjHandCase) == 88 && sizeof(RmjHandResult) == 64, "k_eval_hands stages cases / results by these sizes");
#ifndef RMJ_EVAL_WAVES
#define RMJ_EVAL_WAVES 6   /* 129 VGPR left alone = three waves per SIMD: 0.72 G hands/s; four 0.84, five 0.906, six 0.91-0.92, seven 0.905, eight 0.84 */
#endif
#if RMJ_EVAL_WAVES > 0
#define RMJ_EVAL_OCC __attribute__((amdgpu_waves_per_eu(RMJ_EVAL_WAVES, RMJ_EVAL_WAVES)))
#else
#define RMJ_EVAL_OCC
#endif
__global__ __launch_bounds__(256) RMJ_EVAL_OCC void k_eval_hands(const RmjHandCase* cases, uint32_t n, RmjHandResult* out) {
    __shared__ EvalShared shw[WPB];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, rb = lane & 48, row = lane >> 4;
    const uint32_t k0 = (blockIdx.x * WPB + wave) * 4u;   // first hand of the wave
    if (k0 >= n) return;
    EvalShared& sh = shw[wave];
    const uint32_t k = k0 + (uint32_t)row;
    const bool live = k < n;
    {
        const uint32_t words = (n - k0 < 4u ? n - k0 : 4u) * (uint32_t)(sizeof(RmjHandCase) / 4);
        const uint32_t* src = reinterpret_cast<const uint32_t*>(cases + k0);
        for (uint32_t i = lane; i < words; i += 64) sh.in[i] = src[i];
    }
    wave_sync();
    const RmjHandCase& hc = *reinterpret_cast<const RmjHandCase*>(reinterpret_cast<const uint8_t*>(sh.in) + (size_t)row * sizeof(RmjHandCase));
    const int nt = live ? (hc.n_tiles > 14 ? 14 : hc.n_tiles) : 0;
    const int nm = live ? (hc.n_melds > 4 ? 4 : hc.n_melds) : 0;
    const bool sanma = hc.is_sanma != 0;
    // ---- lane = concealed tile: histogram of the given tiles, red fives
    uint32_t ca = 0, cb = 0, cc = 0, cd = 0;
    const int tile = r < nt ? (int)hc.tiles[r] : 0;
    if (r < nt) {
        const int t = tile >> 2, s = t_suit(t);
        const uint32_t one = 1u << (3 * (t - 9 * s));
        ca = s == 0 ? one : 0u; cb = s == 1 ? one : 0u; cc = s == 2 ? one : 0u; cd = s == 3 ? one : 0u;
    }
    PH conc;
    conc.a = e4_rsum(ca, rb); conc.b = e4_rsum(cb, rb); conc.c = e4_rsum(cc, rb); conc.d = e4_rsum(cd, rb);
    int aka = __popc(e4_ballot(r < nt && is_aka(tile), rb));
    // ---- lane = meld: the packed aggregate (agg_from_views), the tiles it adds to the dora histogram, HandEvaluator::new's kan fix
    //      (hand_evaluator.rs:43-62: a concealed hand that still lists all four tiles of a kan loses one)
    E4Meld mp;
    uint32_t ma_ = 0, mb_ = 0, mc_ = 0, md_ = 0;   // meld tiles (one-hot sums)
    uint32_t fa_ = 0, fb_ = 0, fc_ = 0, fd_ = 0;   // kan fix
    {
        const bool mv = r < nm;
        const RmjMeldView& v = hc.melds[r & 3];
        const int ntm = mv ? (v.n_tiles > 4 ? 4 : v.n_tiles) : 0;
        const uint32_t t0id = v.tiles[0], t1id = v.tiles[1], t2id = v.tiles[2], t3id = v.tiles[3];
        const bool chi = v.meld_type == RMJ_MELD_CHI;
        int lo1 = 99, lo2 = 99;   // the two lowest types among the meld's tiles
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int t = (int)(q == 0 ? t0id : (q == 1 ? t1id : (q == 2 ? t2id : t3id))) >> 2;
            if (q < ntm) {
                if (t < lo1) { lo2 = lo1; lo1 = t; } else if (t < lo2) lo2 = t;
                const int s = t_suit(t);
                const uint32_t one = 1u << (3 * (t - 9 * s));
                ma_ += s == 0 ? one : 0u; mb_ += s == 1 ? one : 0u; mc_ += s == 2 ? one : 0u; md_ += s == 3 ? one : 0u;
            }
        }
        const int t0 = chi ? lo1 : (int)(t0id >> 2);
        const bool trip = ntm >= 3 && (chi ? lo1 == lo2 : (t0id >> 2) == (t1id >> 2));
        mp = e4_meld_lane(mv, v.meld_type, ntm, t0id, t1id, t2id, t3id, t0, trip, v.opened != 0);
        if (mv && v.meld_type >= RMJ_MELD_DAIMINKAN) {
            const int t = (int)(t0id >> 2);
            if (t < 34 && ph_cnt(conc, t) == 4) {
                const int s = t_suit(t);
                const uint32_t one = 1u << (3 * (t - 9 * s));
                fa_ = s == 0 ? one : 0u; fb_ = s == 1 ? one : 0u; fc_ = s == 2 ? one : 0u; fd_ = s == 3 ? one : 0u;
            }
        }
    }
    const E4Meld ma = e4_meld_reduce(mp, rb);
    aka += e4m_aka(ma);
    PH hand = conc, full = conc;
    hand.a -= e4_rsum(fa_, rb); hand.b -= e4_rsum(fb_, rb); hand.c -= e4_rsum(fc_, rb); hand.d -= e4_rsum(fd_, rb);
    full.a += e4_rsum(ma_, rb); full.b += e4_rsum(mb_, rb); full.c += e4_rsum(mc_, rb); full.d += e4_rsum(md_, rb);
    const int total = ph_total(hand) + 3 * nm;
    uint64_t waits = 0ull;
    if (__ballot(live && total == 13)) {
        if (live && total == 13) waits = rmj4::r4_waits_probe(hand.a, hand.b, hand.c, hand.d);
    }
    const int win34 = (hc.win_tile >> 2) < 34 ? (hc.win_tile >> 2) : 33;
    PH h14 = hand, f14 = full;
    if (total == 13) {
        ph_add(h14, win34);
        ph_add(f14, win34);
        aka += is_aka(hc.win_tile);
    }
    // ---- lane = indicator: dora (lanes 0..4) and ura (lanes 8..12) counts over the full histogram
    int dora, ura;
    {
        const bool is_d = r < 5 && r < hc.n_dora, is_u = r >= 8 && r < 13 && r - 8 < hc.n_ura;
        int cnt = 0;
        if (is_d || is_u) {
            const int ind = is_d ? hc.dora[r & 7] : hc.ura[(r - 8) & 7];
            const int nt34 = next_dora34((ind >> 2) < 34 ? (ind >> 2) : 33, sanma);
            cnt = ph_cnt(f14, nt34);
            if (sanma && nt34 == 30) cnt += hc.kita_count;
        }
        dora = (int)e4_rsum(is_d ? (uint32_t)cnt : 0u, rb);
        ura = (int)e4_rsum(is_u ? (uint32_t)cnt : 0u, rb);
    }
    E4In in;
    in.on = live;
    in.hand14 = h14;
    in.ma = ma;
    in.win34 = win34;
    uint32_t cf = 0;
    if (hc.tsumo) cf |= CF_TSUMO;
    if (hc.riichi) cf |= CF_RIICHI;
    if (hc.double_riichi) cf |= CF_DOUBLE_RIICHI;
    if (hc.ippatsu) cf |= CF_IPPATSU;
    if (hc.haitei) cf |= CF_HAITEI;
    if (hc.houtei) cf |= CF_HOUTEI;
    if (hc.rinshan) cf |= CF_RINSHAN;
    if (hc.chankan) cf |= CF_CHANKAN;
    if (hc.tsumo_first_turn) cf |= CF_FIRST_TURN;
    in.cf = cf;
    in.dora = dora & 0xFF; in.aka = aka & 0xFF; in.ura = ura & 0xFF;
    in.nuki = sanma ? hc.kita_count : 0;
    in.round_wind34 = 27 + (hc.round_wind & 3);
    in.seat_wind34 = 27 + (hc.player_wind & 3);
    in.sanma = sanma;
    in.honba = hc.honba;
    const E4Out o = e4_calc(in, r, rb);
    // ---- the 64-byte result: dword r by lane r, the ordered yaku list through LDS bytes
    uint32_t w = 0u;
    if (r == 6) w = o.shape ? (uint32_t)o.han : 0u;
    if (r == 7) w = o.shape ? (uint32_t)o.fu : 0u;
    if (r == 8) w = o.ron;
    if (r == 9) w = o.tsumo_oya;
    if (r == 10) w = o.tsumo_ko;
    if (r == 12) w = (uint32_t)waits;
    if (r == 13) w = (uint32_t)(waits >> 32);
    if (r == 14) w = (uint32_t)(waits != 0ull) | ((uint32_t)o.shape << 8);
    sh.out[row][r] = w;
    wave_sync();
    int ny = 0;
    if (__ballot(live && o.shape)) {
        if (live && o.shape) ny = e4_yaku_list(o.kind, o.ym, reinterpret_cast<uint8_t*>(&sh.out[row][1]), r, rb);
    }
    if (r == 0) sh.out[row][0] = (uint32_t)o.is_win | ((uint32_t)o.yakuman << 8) | ((uint32_t)o.shape << 16) | ((uint32_t)ny << 24);
    wave_sync();
    if (live) reinterpret_cast<uint32_t*>(out + k)[r] = sh.out[row][r];
}

// agari.rs:65-73 + hand_evaluator.rs:178-213 over raw histograms.  Round 3: FOUR hands per wave - one 16-lane row per hand like
// the step kernel's tier 0: the hand's 34 counts arrive as three coalesced byte loads per row (round 2: one hand per wave, every lane
// walked the same 34 bytes one by one), the row OR-reduces them into the packed histogram, is_agari is closed-form per row and the
// waits come from the row-form probe of the step kernel (r4_waits_probe).
__global__ __launch_bounds__(256) void k_agari_counts(const uint8_t* counts, uint32_t n, uint8_t* agari, uint8_t* tenpai, uint64_t* waits) {
    const int lane = threadIdx.x & 63, r = lane & 15, rb = lane & 48;
    const uint32_t k = (blockIdx.x * 4u + (threadIdx.x >> 6)) * 4u + (uint32_t)(lane >> 4);
    const bool live = k < n;
    PH h = {0, 0, 0, 0};
    {
        uint32_t w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const int t = r + 16 * j;
            if (live && t < 34) {
                const int s = t_suit(t);
                const uint32_t f = ((uint32_t)counts[(size_t)k * 34 + t] & 7u) << (3 * (t - 9 * s));
                w[0] |= s == 0 ? f : 0u; w[1] |= s == 1 ? f : 0u; w[2] |= s == 2 ? f : 0u; w[3] |= s == 3 ? f : 0u;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; q++) w[q] = (uint32_t)rmj4::rbc((int)rmj4::row_or16(w[q]), rb + 15);
        h.a = w[0]; h.b = w[1]; h.c = w[2]; h.d = w[3];
    }
    const bool ag = is_agari(h);
    uint64_t w = 0ull;
    if (__ballot(live && ph_total(h) == 13)) {
        if (live && ph_total(h) == 13) w = rmj4::r4_waits_probe(h.a, h.b, h.c, h.d);
    }
    if (live && r == 0) {
        agari[k] = ag;
        tenpai[k] = w != 0ull;
        waits[k] = w;
    }
}

// Observation.encode() / encode_extended() for every (game, seat): one wave (= one block) per (game, seat).
// out[g][seat][C][W] f32, C = 74 or 215.  The tensor is assembled in an 84-channel LDS staging buffer (11 KB: 13 blocks per
// CU) and streamed out group by group: base channels, then the two extended groups (rmj_encode.hip.h).
// only_active: 0 = every seat, 1 = acting seats (other rows zeroed), 2 = acting seats (other rows untouched).
template <int W>
__device__ __forceinline__ void enc_stream_out(float* dst, const float* buf, int n_floats, int lane) {
    // both even: 16-byte rows are not guaranteed (215 x 27 is odd), 8-byte pairs are when the offset and count are even
    if ((n_floats & 1) == 0 && ((reinterpret_cast<uintptr_t>(dst) & 7u) == 0)) {
        for (int i = lane; i < n_floats / 2; i += 64) reinterpret_cast<float2*>(dst)[i] = reinterpret_cast<const float2*>(buf)[i];
    } else {
        for (int i = lane; i < n_floats; i += 64) dst[i] = buf[i];
    }
}
// n_floats floats from LDS to global memory in 16-byte stores: `dst` is 8-byte aligned (every row of the tensors is an even
// number of floats from a 16-byte aligned base), so at most two floats precede the first 16-byte boundary and at most
// three follow the last; the body goes out as dwordx4 (the epilogue of a wave is store-issue bound: half the instructions
// of the 8-byte version).  The LDS side is read as two 8-byte halves (its offset is only 8-byte aligned after the head).
__device__ __forceinline__ void enc_stream_out16(float* dst, const float* buf, int n_floats, int lane) {
    const int head = (int)(((16u - (uint32_t)(reinterpret_cast<uintptr_t>(dst) & 15u)) & 15u) >> 2);  // 0 or 2
    const int body = (n_floats - head) >> 2, tail0 = head + 4 * body;
    if (lane < head) dst[lane] = buf[lane];
    float4* d4 = reinterpret_cast<float4*>(dst + head);
    for (int i = lane; i < body; i += 64) {
        const float2 lo = *reinterpret_cast<const float2*>(buf + head + 4 * i), hi = *reinterpret_cast<const float2*>(buf + head + 4 * i + 2);
        d4[i] = make_float4(lo.x, lo.y, hi.x, hi.y);
    }
    if (lane < n_floats - tail0) dst[tail0 + lane] = buf[tail0 + lane];
}
__device__ __forceinline__ void enc_zero16(float* dst, int n_floats, int lane) {
    const int head = (int)(((16u - (uint32_t)(reinterpret_cast<uintptr_t>(dst) & 15u)) & 15u) >> 2);
    const int body = (n_floats - head) >> 2, tail0 = head + 4 * body;
    if (lane < head) dst[lane] = 0.0f;
    float4* d4 = reinterpret_cast<float4*>(dst + head);
    for (int i = lane; i < body; i += 64) d4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lane < n_floats - tail0) dst[tail0 + lane] = 0.0f;
}
// Slots of the compact observation batch, level one: a block of 1024 threads scans the acting-seat counts of its 1024 games
// (finished games have none): offs[g] = acting seats in the block's games before g, totals[block] = the block's sum.  The
// encoder adds the totals of the blocks before its game's block (at most 512 numbers, one wave reduction).
#define OBS_SCAN_BLOCK 1024
__global__ __launch_bounds__(OBS_SCAN_BLOCK) void k_obs_offsets(const uint32_t* __restrict__ status, uint32_t n, uint32_t* __restrict__ offs,
                                                               uint32_t* __restrict__ totals) {
    __shared__ uint32_t wsum[OBS_SCAN_BLOCK / 64];
    const uint32_t t = threadIdx.x, g = blockIdx.x * OBS_SCAN_BLOCK + t, lane = t & 63u, wv = t >> 6;
    uint32_t c = 0u;
    if (g < n) {
        const uint32_t w = status[g];
        c = ((w >> 16) & 0xFFu) ? 0u : (uint32_t)__popc(w & 0xFu);
    }
    uint32_t inc = c;   // inclusive scan inside the wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t v = (uint32_t)__shfl_up((int)inc, d, 64);
        if ((int)lane >= d) inc += v;
    }
    if (lane == 63u) wsum[wv] = inc;
    __syncthreads();
    uint32_t before = 0u;
    for (uint32_t k = 0; k < wv; k++) before += wsum[k];
    if (g < n) offs[g] = before + inc - c;
    if (t == OBS_SCAN_BLOCK - 1) totals[blockIdx.x] = before + inc;
}
// sum of totals[0 .. nb) by one wave (nb <= 512 for 524 288 games)
__device__ __forceinline__ uint32_t obs_block_prefix(const uint32_t* __restrict__ totals, uint32_t nb, int lane) {
    uint32_t s = 0u;
    for (uint32_t k = (uint32_t)lane; k < nb; k += 64u) s += totals[k];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += (uint32_t)__shfl_xor((int)s, d, 64);
    return s;
}
// Observation.encode() of the games [g0, g0 + gridDim.x): ONE block (= one wave) per game, which walks the seats it has to
// encode - with only_active that is the acting seat (one, rarely two or three), so the launch has a quarter of the blocks of
// a (game, seat) grid and no early-exit blocks.  The tensor of a seat is staged as one byte per cell (EncByteSink: 2.5 KB,
// 4.9 KB of LDS per block with the record, the histograms and the value table) and leaves as a stream of 16-byte stores.
// Round 6: four waves per SIMD.  Left alone the kernel takes 100 VGPR (four waves) under the default flags and 60 (seven) under -disable-machine-licm; alone
// it runs the same either way (3P 0.1347 -> 0.1378 ms, occupancy 5 -> 8 changed nothing in round 5), but next to step and sampler kernels of other
// shards on other streams the seven-wave form crowds them out: the trainer loop as 4 shards on 4 streams 304 -> 345 M env.step/s, 2 shards 290 -> 315 M,
// compact batch 305 -> 334 M with the cap (five waves: 336 / 292 / 320; round-5 binary: 340-350 / 312-320 / 311-316).
#ifndef RMJ_ENC_WAVES
#define RMJ_ENC_WAVES 4
#endif
#if RMJ_ENC_WAVES > 0
#define RMJ_ENC_OCC __attribute__((amdgpu_waves_per_eu(RMJ_ENC_WAVES, RMJ_ENC_WAVES)))
#else
#define RMJ_ENC_OCC
#endif
// `offs` != nullptr: compact output (rmj_encode_compact_device) - the observations of the acting seats, one after the other in
// (game, seat) order: observation offs[g] + j is the j-th acting seat of game g, `index` receives game * 4 + seat.
template <bool SANMA, bool COMPACT>
__global__ __launch_bounds__(64) RMJ_ENC_OCC void k_encode_base(Env E, int only_active, float* __restrict__ out, uint32_t g0,
                                                               const uint32_t* __restrict__ offs, int32_t* __restrict__ index, uint32_t capacity,
                                                               const uint32_t* __restrict__ totals, uint32_t* __restrict__ count) {
    constexpr int W = SANMA ? ENC_W3 : ENC_W4, NPP = SANMA ? 3 : 4;
    __shared__ GState st;
    __shared__ __attribute__((aligned(16))) uint8_t raw[(ENC_CH * W + 4 + 15) / 16 * 16];
    __shared__ float lut[ENC_LUT];
    __shared__ uint32_t hist[ENC_HIST_WORDS];
    const int lane = threadIdx.x & 63;
    const uint32_t g = g0 + blockIdx.x;
    // the record is requested together with the status word (nearly every game has a seat to act): one memory round trip
    uint4 rec = make_uint4(0u, 0u, 0u, 0u);
    if (lane < (int)(sizeof(GState) / 16)) rec = reinterpret_cast<const uint4*>(E.core + g)[lane];
    const uint32_t stw = E.status[g];
    const uint32_t am = ((stw >> 16) & 0xFFu) ? 0u : (stw & 0xFu);
    const size_t RS = E.enc_stride;   // row stride in floats (>= 74 x W; rows padded to a multiple of 256 B leave at 1.3-1.4 x the rate)
    float* base = out + (size_t)g * 4 * RS;
    uint32_t slot = 0u;
    if (COMPACT) {
        if (blockIdx.x == 0) {   // the size of the batch: all block totals
            const uint32_t all = obs_block_prefix(totals, (E.n_games + OBS_SCAN_BLOCK - 1) / OBS_SCAN_BLOCK, lane);
            if (lane == 0) *count = all;
        }
        if (am == 0u) return;
        slot = offs[g] + obs_block_prefix(totals, g / OBS_SCAN_BLOCK, lane);
    } else if (only_active && am == 0u) {
        if (only_active == 1)
            for (int z = 0; z < 4; z++) enc_zero16(base + (size_t)z * RS, ENC_CH * W, lane);
        return;
    }
    enc_lut_init(lut, lane);
    if (lane < (int)(sizeof(GState) / 16)) reinterpret_cast<uint4*>(&st)[lane] = rec;
    wave_sync();
    const GState& S = st;
    for (int seat = 0; seat < 4; seat++) {
        float* dst = base + (size_t)seat * RS;
        const bool acts = (am >> seat) & 1u;
        if (COMPACT) {
            if (seat >= NPP || !acts) continue;
            if (slot >= capacity) return;                  // (the count tells the caller that the buffer was too small)
            dst = out + (size_t)slot * RS;
            if (lane == 0) index[slot] = (int32_t)(g * 4u + (uint32_t)seat);
            slot += 1u;
        } else if (seat >= NPP || (only_active && !acts)) {
            if (only_active != 2) enc_zero16(dst, ENC_CH * W, lane);
            continue;
        }
        const int head = (int)(((16u - (uint32_t)(reinterpret_cast<uintptr_t>(dst) & 15u)) & 15u) >> 2);  // 0 or 2 floats
        EncByteSink<W> o{raw + ((4 - head) & 3), lut, lane, -1.0f};
#ifdef RMJ_ENC_NOCOMPUTE   /* experiment: the memory side alone (record in, 74 x W floats out) */
        o.zero();
        wave_sync();
#else
        encode_seat_to<SANMA>(S, seat, lane, hist, o, true);
#endif
        enc_emit_bytes<W>(dst, o.cells, lut, lane, head, o.big);
        wave_sync();
    }
}
// n floats computed per element into 16-byte stores (4-byte aligned dst: up to three floats before the first boundary)
template <class F>
__device__ __forceinline__ void enc_emit_fn(float* dst, int n_floats, int lane, F f) {
    int head = (int)(((16u - (uint32_t)(reinterpret_cast<uintptr_t>(dst) & 15u)) & 15u) >> 2);
    if (head > n_floats) head = n_floats;
    const int body = (n_floats - head) >> 2, tail0 = head + 4 * body;
    if (lane < head) dst[lane] = f(lane);
    float4* d4 = reinterpret_cast<float4*>(dst + head);
    for (int i = lane; i < body; i += 64) {
        const int e = head + 4 * i;
        d4[i] = make_float4(f(e), f(e + 1), f(e + 2), f(e + 3));
    }
    if (lane < n_floats - tail0) dst[tail0 + lane] = f(tail0 + lane);
}
// encode_extended() of every (game, seat): one wave per seat.  The 215 x W tensor leaves in three groups that share one
// staging area of bytes: the 74 base channels (EncByteSink), the extended scalars (four per-column channels as floats and a
// table of the 53 channels that are one value per row) and the 84 meld-overview channels (a 0/1 pattern).  6 KB of LDS per
// block instead of 12 KB: the kernel waits on table lookups (the ukeire walk), and its duration is inversely proportional to
// the resident waves (measured by capping them: 13 / 8 / 5 / 3 blocks per CU -> 1.05 / 1.63 / 2.24 / 3.67 ms).
#ifndef RMJ_ENCX_WAVES
#define RMJ_ENCX_WAVES 6   /* 3P (85 VGPR left alone = five waves): six waves 691 -> 661 us, seven 667, eight 755; 4P (63 VGPR) the same at any */
#endif
#if RMJ_ENCX_WAVES > 0
#define RMJ_ENCX_OCC __attribute__((amdgpu_waves_per_eu(RMJ_ENCX_WAVES, RMJ_ENCX_WAVES)))
#else
#define RMJ_ENCX_OCC
#endif
template <bool SANMA>
__global__ __launch_bounds__(64) RMJ_ENCX_OCC void k_encode_ext(Env E, int only_active, const float* __restrict__ decay, float* __restrict__ out) {
    constexpr int W = SANMA ? ENC_W3 : ENC_W4;
    constexpr int CH = ENC_EXT_CH;
    __shared__ GState st;
    __shared__ __attribute__((aligned(16))) uint8_t raw[(ENC_EXT_C_SLOTS * W + 4 + 15) / 16 * 16];
    __shared__ float lut[ENC_LUT];
    __shared__ float tab[ENC_EXT_B_SLOTS];
    __shared__ float col4[4 * W];
    __shared__ uint32_t hist[ENC_HIST_WORDS];
    const int lane = threadIdx.x & 63;
    const uint32_t g = blockIdx.x >> 2;
    const int seat = blockIdx.x & 3;
    float* dst = out + ((size_t)g * 4 + seat) * CH * W;
    if (only_active) {  // cheap early-out from the 4-byte status word, before the record is fetched
        const uint32_t stw = E.status[g];
        const bool acts = ((stw >> seat) & 1u) && !((stw >> 16) & 0xFFu);
        if (!acts || seat >= (SANMA ? 3 : 4)) {
            if (only_active == 1)
                for (int i = lane; i < CH * W; i += 64) dst[i] = 0.0f;
            return;
        }
    }
    if (lane < (int)(sizeof(GState) / 16)) reinterpret_cast<uint4*>(&st)[lane] = reinterpret_cast<const uint4*>(E.core + g)[lane];
    enc_lut_init(lut, lane);
    wave_sync();
    const GState& S = st;
    if (seat >= (SANMA ? 3 : 4)) {
        for (int i = lane; i < CH * W; i += 64) dst[i] = 0.0f;
        return;
    }
    auto head_of = [](const float* p) { return (int)(((16u - (uint32_t)(reinterpret_cast<uintptr_t>(p) & 15u)) & 15u) >> 2); };
    {   // channels 0..73 (encode_base_into: its own tiles-left count, see encode_seat_to)
        const int head = head_of(dst);
        EncByteSink<W> o{raw + ((4 - head) & 3), lut, lane, -1.0f};
        encode_seat_to<SANMA>(S, seat, lane, hist, o, true, true);
        enc_emit_bytes<W>(dst, o.cells, lut, lane, head, o.big);
        wave_sync();
    }
    {   // channels 74..93 and 178..214
        const int n_legal = (((S.active_mask >> seat) & 1u) && !S.is_done) ? (int)E.nlegal[(size_t)g * 4 + seat] : 0;
        encode_ext_scalars<SANMA>(S, seat, tab, col4, lane, E.sh, decay, E.legal + ((size_t)g * 4 + seat) * RMJ_MAX_LEGAL, n_legal);
        enc_emit_fn(dst + 74 * W, 20 * W, lane, [&](int e) { return e < 4 * W ? col4[e] : tab[e / W]; });
        enc_emit_fn(dst + 178 * W, 37 * W, lane, [&](int e) { return tab[20 + e / W]; });
        wave_sync();
    }
    {   // channels 94..177
        float* d = dst + 94 * W;
        const int head = head_of(d);
        uint8_t* cells = raw + ((4 - head) & 3);
        for (int i = lane; i < (int)sizeof(raw) / 16; i += 64) reinterpret_cast<uint4*>(raw)[i] = make_uint4(0u, 0u, 0u, 0u);
        wave_sync();
        encode_ext_melds<SANMA>(S, seat, cells, lane);
        enc_emit_bytes<W, ENC_EXT_C_SLOTS>(d, cells, lut, lane, head);
    }
}

// ---- auxiliary encoders (row N3): kawa overview, yaku possibility, furiten-ron possibility ----------------------
// One wave per game; absolute seat order, public information only (the same for every observing seat).
//   which 0  Observation.encode_kawa_overview           (observation/python.rs:881-925, observation_3p/python.rs:759-810)
//   which 1  Observation.encode_yaku_possibility        (observation/python.rs:327-455 over yaku_checker.rs:27-412)
//   which 2  Observation.encode_furiten_ron_possibility (observation/python.rs:251-293)
template <bool SANMA>
__global__ __launch_bounds__(64) void k_encode_aux(Env E, int which, float* __restrict__ out) {
    constexpr int W = SANMA ? ENC_W3 : ENC_W4, NP = SANMA ? 3 : 4;
    __shared__ GState st;
    const int lane = threadIdx.x & 63;
    const uint32_t g = blockIdx.x;
    if (which == 2) {  // tsumogiri_flags is never filled by the reference (observation/mod.rs:105): every row stays 1.0
        for (int i = lane; i < NP * 21; i += 64) out[(size_t)g * NP * 21 + i] = 1.0f;
        return;
    }
    if (lane < (int)(sizeof(GState) / 16)) reinterpret_cast<uint4*>(&st)[lane] = reinterpret_cast<const uint4*>(E.core + g)[lane];
    wave_sync();
    const GState& S = st;
    if (which == 0) {
        // lane = tile column.  Channel k (< 4) is set iff the seat has discarded more than k tiles of the type; channels 4..6 are
        // the red-five flags with the reference's ids and columns (20 / 24 / 28; 4P column 5 + 9 i, 3P (5, 6) and (6, 15))
        float* dst = out + (size_t)g * NP * 7 * W;
        for (int p = 0; p < NP; p++) {
            const PState& P = S.p[p];
            int cnt = 0;
            bool aka0 = false, aka1 = false, aka2 = false;
            for (int k = 0; k < P.n_discards; k++) {
                const int t = P.discards[k];
                cnt += (lane < W && enc_col<SANMA>(t >> 2) == lane);
                aka0 |= t == 20;
                aka1 |= t == 24;
                aka2 |= t == 28;
            }
            if (lane < W) {
                for (int k = 0; k < 4; k++) dst[(p * 7 + k) * W + lane] = cnt > k ? 1.0f : 0.0f;
                if (!SANMA) {
                    dst[(p * 7 + 4) * W + lane] = (aka0 && lane == 5) ? 1.0f : 0.0f;
                    dst[(p * 7 + 5) * W + lane] = (aka1 && lane == 14) ? 1.0f : 0.0f;
                    dst[(p * 7 + 6) * W + lane] = (aka2 && lane == 23) ? 1.0f : 0.0f;
                } else {
                    dst[(p * 7 + 4) * W + lane] = 0.0f;
                    dst[(p * 7 + 5) * W + lane] = (aka1 && lane == 6) ? 1.0f : 0.0f;
                    dst[(p * 7 + 6) * W + lane] = (aka2 && lane == 15) ? 1.0f : 0.0f;
                }
            }
        }
        return;
    }
    // which == 1.  lane = tile type: visible[type] = own discards + dora indicators (yaku_checker.rs:42-58); the meld facts
    // are wave-uniform loops over <= 4 melds x <= 4 tiles.
    float* dst = out + (size_t)g * NP * 21 * 2;
    for (int p = 0; p < NP; p++) {
        const PState& P = S.p[p];
        int vis = 0;
        for (int k = 0; k < P.n_discards; k++) vis += (P.discards[k] >> 2) == lane;
        for (int k = 0; k < S.n_dora; k++) vis += (S.dora[k] >> 2) == lane;
        const uint64_t vis2 = __ballot(lane < 34 && vis >= 2), vis3 = __ballot(lane < 34 && vis >= 3), vis4 = __ballot(lane < 34 && vis >= 4);
        uint64_t set_types = 0;  // types with a meld of >= 3 tiles starting with that type (yaku_checker.rs:68-75)
        bool any_yaochu = false, simple_tile = false, any_number = false, any_honor = false, any_non_terminal = false;
        bool suit0 = false, suit1 = false, suit2 = false, has_run = false, no_yaochu_meld = false, junchan_bad = false;
        const int nm = P.n_melds;
        for (int m = 0; m < nm; m++) {
            const int len = (P.meld_type[m] == RMJ_MELD_CHI || P.meld_type[m] == RMJ_MELD_PON) ? 3 : 4;
            const int t0 = P.meld_tiles[m][0] >> 2, t1 = P.meld_tiles[m][1] >> 2, t2 = P.meld_tiles[m][2] >> 2;
            set_types |= 1ull << t0;
            if (len == 3 && t0 + 1 == t1 && t1 + 1 == t2 && t0 < 27) has_run = true;
            bool m_yaochu = false, m_terminal = false, m_honor = false;
            for (int k = 0; k < len; k++) {
                const int tt = P.meld_tiles[m][k] >> 2;
                const bool honor = tt >= 27, terminal = !honor && (tt % 9 == 0 || tt % 9 == 8);
                m_yaochu |= honor || terminal;
                m_terminal |= terminal;
                m_honor |= honor;
                any_number |= !honor;
                any_non_terminal |= !terminal;
                simple_tile |= !honor && !terminal;
                if (!honor) { suit0 |= tt < 9; suit1 |= tt >= 9 && tt < 18; suit2 |= tt >= 18; }
            }
            any_yaochu |= m_yaochu;
            any_honor |= m_honor;
            if (!m_yaochu) no_yaochu_meld = true;
            if (m_honor || !m_terminal) junchan_bad = true;
        }
        const int ns = (int)suit0 + (int)suit1 + (int)suit2;
        const int round_t = 27 + S.round_wind, seat_t = 27 + (p + NP - S.oya) % NP;
        auto yakuhai_imp = [&](int tt) { return !((set_types >> tt) & 1ull) && ((vis3 >> tt) & 1ull); };
        const uint64_t koku_req = 0x101ull | (0x101ull << 9) | (0x101ull << 18) | (0x7Full << 27);
        bool imp = false;
        switch (lane) {
            case 0: imp = any_yaochu; break;
            case 1: imp = yakuhai_imp(31); break;
            case 2: imp = yakuhai_imp(32); break;
            case 3: imp = yakuhai_imp(33); break;
            case 4: imp = yakuhai_imp(round_t); break;
            case 5: imp = yakuhai_imp(seat_t); break;
            case 6: imp = nm > 0 && ns >= 2; break;
            case 7: imp = nm > 0 && (ns >= 2 || (ns == 1 && any_honor)); break;
            case 8: imp = has_run; break;
            case 9: imp = nm > 0; break;
            case 10: imp = ((vis4 >> 31) & 7ull) != 0ull; break;
            case 11: imp = (((vis2 & ~set_types) >> 31) & 7ull) != 0ull; break;
            case 12: imp = any_number; break;
            case 13: imp = any_non_terminal; break;
            case 14: imp = simple_tile; break;
            case 15: imp = nm > 0 || (vis4 & koku_req) != 0ull; break;
            case 16: imp = no_yaochu_meld; break;
            case 17: imp = junchan_bad; break;
            case 19: imp = nm > 0; break;
            default: break;  // 18 sanshoku, 20 ittsu: never impossible
        }
        if (lane < 21) {
            const float v = imp ? 0.0f : 1.0f;
            reinterpret_cast<float2*>(dst)[p * 21 + lane] = make_float2(v, v);
        }
    }
}


// shanten.rs:244-261 / :470-484 (calculate_shanten / calculate_shanten_3p over raw histograms): one thread per hand
// one thread per hand; the block's 256 hands (8 704 contiguous bytes) are fetched as coalesced 16-byte loads into LDS first
// (round 2: every thread read its own 34 bytes at a 34-byte stride)
__global__ __launch_bounds__(256) void k_shanten(ShantenTables T, const uint8_t* counts, uint32_t n, int sanma, int8_t* out) {
    __shared__ __attribute__((aligned(16))) uint8_t tile[256 * 34 + 16];
    const uint32_t base = blockIdx.x * 256u;
    const uint32_t here = n - base < 256u ? n - base : 256u;
    const size_t off0 = (size_t)base * 34;                       // 8 704 * block: 16-byte aligned when `counts` is
    const uint32_t bytes = here * 34u;
    if ((reinterpret_cast<uintptr_t>(counts) & 15u) == 0u) {
        for (uint32_t i = threadIdx.x; i * 16u < bytes; i += 256u) {
            if (i * 16u + 16u <= bytes) reinterpret_cast<uint4*>(tile)[i] = reinterpret_cast<const uint4*>(counts + off0)[i];
            else for (uint32_t b = i * 16u; b < bytes; b++) tile[b] = counts[off0 + b];
        }
    } else {
        for (uint32_t b = threadIdx.x; b < bytes; b += 256u) tile[b] = counts[off0 + b];
    }
    __syncthreads();
    const uint32_t i = base + threadIdx.x;
    if (i >= n) return;
    PH h = {0, 0, 0, 0};
    int total = 0;
    const uint8_t* mine = tile + threadIdx.x * 34;
#pragma unroll
    for (int t = 0; t < 34; t++) {
        const uint32_t c = mine[t];
        total += (int)c;
        const int s = t_suit(t);
        ph_addv(h, s, (c & 7u) << (3 * (t - 9 * s)));
    }
    out[i] = (int8_t)sh_shanten(h, total / 3, sanma != 0, T);
}
// (round 4's walk: 82 VGPRs = five waves per SIMD left alone; compiled for six: +5 %, eight: the same.  Round 5's pair-dense walk, 74 VGPRs left alone:
//  five waves 0.388, six 0.412, seven 0.425, eight 0.430 G hands/s of best ukeire on random hands)
#ifndef RMJ_UKE_WAVES
#define RMJ_UKE_WAVES 8
#endif
#define RMJ_UKE_OCC __attribute__((amdgpu_waves_per_eu(RMJ_UKE_WAVES, RMJ_UKE_WAVES)))
__global__ __launch_bounds__(256) RMJ_UKE_OCC void k_ukeire(ShantenTables T, const uint8_t* counts, const uint8_t* visible, uint32_t n, int sanma,
                                                int mode, uint32_t* out) {
    const int lane = threadIdx.x & 63;
    const uint32_t i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const bool sm = sanma != 0;
    const int t = lane;                                       // tile type of this lane
    const uint32_t my_cnt = t < 34 ? counts[(size_t)i * 34 + t] : 0u;
    const uint32_t my_vis = (t < 34 && visible) ? visible[(size_t)i * 34 + t] : 0u;
    // wave-uniform histogram: lane t contributes its field, the four words are OR-reduced over the wave
    PH h = {0, 0, 0, 0};
    {
        const int s = t < 34 ? t_suit(t) : 0;
        uint32_t f = t < 34 ? (my_cnt & 7u) << (3 * (t - 9 * s)) : 0u;
        uint32_t w[4] = {s == 0 ? f : 0u, s == 1 ? f : 0u, s == 2 ? f : 0u, s == 3 ? f : 0u};
#pragma unroll
        for (int k = 0; k < 4; k++) {
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) w[k] |= (uint32_t)__shfl_xor((int)w[k], off, 64);
        }
        h.a = w[0]; h.b = w[1]; h.c = w[2]; h.d = w[3];
    }
    const uint32_t res = sh_ukeire_wave(T, h, my_cnt, my_vis, sm, mode, lane);
    if (lane == 0) out[i] = res;
}

__global__ void k_score(const uint8_t* han, const uint8_t* fu, const uint8_t* oya, const uint8_t* tsumo, const uint32_t* honba,
                        const uint8_t* np, uint32_t n, uint32_t* out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    ScoreOut s = calc_score(han[i], fu[i], oya[i] != 0, tsumo[i] != 0, honba[i], np[i]);
    out[4 * i] = s.total; out[4 * i + 1] = s.ron; out[4 * i + 2] = s.tsumo_oya; out[4 * i + 3] = s.tsumo_ko;
}

// ================================================================= host side
static thread_local std::string g_err;
static int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
#define HIPCHK(x)                                                                          \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) return fail(RMJ_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(e_)); \
    } while (0)

struct rmj_env {
    RmjConfig cfg;
    Env d;
    hipStream_t stream = nullptr;      // the stream every entry point works on: own_stream, or the caller's (rmj_set_stream)
    hipStream_t own_stream = nullptr;
    // extra streams of multi-step device rollouts (rmj_step_random): the games are stepped as k parts, one per stream,
    // so that the draining tail of one part's launch overlaps the bodies of the others'.  Forked from and joined back
    // into `stream` inside the call: every other entry point sees one ordered stream.  Measured at 65 536 games:
    // 1 stream 504 M env.step/s, 2: 605 M, 3: 648 M, 4: 678 M, 6: 469 M, 8: 498 M, 16: 343 M.
    hipStream_t xstream[RMJ_MAX_ROLLOUT_STREAMS - 1] = {};
    hipEvent_t ev_fork = nullptr, ev_join[RMJ_MAX_ROLLOUT_STREAMS - 1] = {};
    Env* d_env = nullptr;  // device-resident copy of `d` (kernels take a pointer, see rmj_kernels.hip.h)
    uint64_t* d_actions = nullptr;
    unsigned long long* d_counter = nullptr;
    uint32_t* d_obs_offs = nullptr;    // [n_games] + [blocks] slots of the compact observation batch (rmj_encode_compact_device)
    uint32_t ring = 0;
    float* d_decay = nullptr;  // expf(-0.2f * age), age 0..31, computed on the host (encode_extended)
    void* d_scratch = nullptr; // staging buffer of the host-copy entry points (grown on demand, never per call)
    size_t scratch_bytes = 0;
    int want_streams = 4;      // parts a multi-step device rollout is cut into (rmj_set_rollout_streams; RMJ_STEP_STREAMS at create)
    int quad = 2;              // device-policy steps: 0 = one game per wave (k_step), 1 = four games per wave (k_step4), 2 = and a
                               // rollout of >= 2 steps is ONE launch in which every wave steps its own games (k_step4<true>); RMJ_STEP4 at create
    // long fused rollouts hand the work out in (quad, chunk) tickets to a grid that fits the chip once (k_step4_queue)
    int queue_chunk = 32;      // calls per ticket (round 5: a ticket is a number of calls of the step function, profiles/r05_ticket_schedule_sweep.txt); RMJ_QUEUE_CHUNK at create, 0 = off (every wave keeps one quad for the rollout)
    hipEvent_t ev_time[2] = {nullptr, nullptr};   // rmj_time_rollout* / rmj_bench_rollout: created with the handle, so that a timed region holds no event create / destroy
    uint32_t* d_qheads = nullptr;   // [8][RMJ_Q_STRIDE] ticket counters, one line per XCD
    uint32_t q_slots_pol[2] = {0, 0};   // waves of k_step4_queue<policy> the device holds at once (the greedy instantiation is compiled for fewer)
    int queue_force = 0;            // RMJ_QUEUE_FORCE at create (tests): tickets for every batch of >= 64 quads
    uint32_t queue_skip_xcds = 0;   // test hook: XCDs whose waves leave the queue kernel at once (RMJ_QUEUE_TEST_SKIP_XCDS at create)
    void* d_heavy = nullptr;        // heavy-first launch order of whole-batch per-step launches (HeavyOrder): two counters, lists, flag arrays
    uint32_t heavy_phase = 0;       // which half the next launch reads
    int heavy_first = 1;            // RMJ_HEAVY_FIRST at create (0: plain block order)
    uint32_t rows_pw = 4;           // games per wave of the non-ticket four-games-per-wave kernels: 4, or 2 / 1 for batches that leave the chip
                                    // latency bound (chosen at create from the batch size; RMJ_ROWS overrides)
    int queue_tail = 0;             // ticket lengths descend towards the expected end of a quad's rollout (q_ticket_plan); RMJ_QUEUE_TAIL=0: equal tickets
    int queue_min_chunk = 5;        // shortest ticket (steps): a rollout of >= 2 tickets per quad runs as tickets; RMJ_QUEUE_MIN_CHUNK at create
    uint32_t max_xcc_id = 0;        // largest HW_REG_XCC_ID seen by a probe launch at create: the ticket rollout assumes ids 0..7 (one L2 per queue)
    uint32_t* d_ev_lost = nullptr;  // [n_games] records a game's ring lost to a late drain (rmj_drain_events), cumulative
    void* d_track = nullptr;        // round tracker (rmj_round_track_device): hand index / scores / meta where every game's round began
    // staging of rmj_drain_format's size call (the records sit in h_pin): reused by the call that brings the text buffer
    bool stage_valid = false;
    int stage_seat = 0;
    uint32_t stage_events = 0;
    double stage_ms[2] = {0, 0};
    std::vector<uint32_t> stage_cursor;
    void* h_pin = nullptr;          // pinned host staging of the host-buffer entry points (rmj_get_legal_compact, rmj_drain_*), grown on demand
    size_t pin_bytes = 0;
    int enc_streams = 0;            // RMJ_ENC_STREAMS at create (0: want_streams): parts of the step + encode rollout
    int enc_parts_quad = -1;        // RMJ_ENC_PARTS_QUAD at create (-1: follow `quad`)
    int enc_fused = 1;              // RMJ_ENC_FUSED at create: the step + encode rollout as ONE launch (k_step4_enc / k_step4_queue_enc); 0 = parts on streams
    uint32_t q_slots_enc = 0;       // waves of k_step4_queue_enc the device holds at once
};
// device staging memory of at least `bytes` bytes, owned by the handle
static int scratch_for(rmj_env* h, size_t bytes, void** out) {
    if (bytes > h->scratch_bytes) {
        HIPCHK(hipStreamSynchronize(h->stream));
        if (h->d_scratch) hipFree(h->d_scratch);
        h->d_scratch = nullptr;
        h->scratch_bytes = 0;
        HIPCHK(hipMalloc(&h->d_scratch, bytes));
        h->scratch_bytes = bytes;
    }
    *out = h->d_scratch;
    return RMJ_OK;
}

static int ensure_device(int device) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(RMJ_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= n) return fail(RMJ_ERR_NO_DEVICE, "device ordinal out of range");
    if (hipSetDevice(device) != hipSuccess) return fail(RMJ_ERR_NO_DEVICE, "hipSetDevice failed");
    return RMJ_OK;
}
static inline dim3 game_grid(uint32_t n) { return dim3((n + WPB - 1) / WPB); }

// Temporary device buffers (and timing events) of one entry point: released on EVERY return path, so that an error
// in the middle of a call (HIPCHK returns at once) strands nothing on the device.
struct DevTmp {
    std::vector<void*> bufs;
    std::vector<hipEvent_t> events;
    ~DevTmp() {
        for (void* q : bufs) if (q) hipFree(q);
        for (hipEvent_t e : events) if (e) hipEventDestroy(e);
    }
    template <typename T>
    hipError_t alloc(T** dst, size_t bytes) {
        *dst = nullptr;
        hipError_t e = hipMalloc((void**)dst, bytes);
        if (e == hipSuccess) bufs.push_back((void*)*dst);
        return e;
    }
    template <typename T>
    int upload(const T* src, size_t count, T** dst) {  // NULL source = optional array not given
        *dst = nullptr;
        if (!src) return RMJ_OK;
        HIPCHK(alloc(dst, count * sizeof(T)));
        HIPCHK(hipMemcpy(*dst, src, count * sizeof(T), hipMemcpyHostToDevice));
        return RMJ_OK;
    }
    hipError_t event(hipEvent_t* e) {
        hipError_t r = hipEventCreate(e);
        if (r == hipSuccess) events.push_back(*e);
        return r;
    }
};

static int shanten_tables_for(int device, ShantenTables* out);
static void launch_encode_base_range(rmj_env* h, hipStream_t st, int only_active, float* d_out, uint32_t g0, uint32_t g1);

extern "C" {

const char* rmj_version(void) { return "riichi_mi355x 0.1 (gfx950)"; }
const char* rmj_last_error(void) { return g_err.c_str(); }
int rmj_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// body of rmj_create; on any failure the caller destroys the partially built handle (rmj_destroy tolerates null members)
static int create_impl(rmj_env* h, const RmjConfig* cfg, uint64_t** d_seeds_out) {
    uint32_t ring = cfg->event_ring ? cfg->event_ring : 64;
    uint32_t r2 = 64;
    while (r2 < ring) r2 <<= 1;
    h->ring = r2;
    if (const char* e = getenv("RMJ_STEP_STREAMS")) h->want_streams = atoi(e);
    if (const char* e = getenv("RMJ_STEP4")) h->quad = atoi(e);
    if (const char* e = getenv("RMJ_QUEUE_CHUNK")) h->queue_chunk = atoi(e);
    if (const char* e = getenv("RMJ_QUEUE_FORCE")) h->queue_force = atoi(e);
    if (const char* e = getenv("RMJ_QUEUE_TEST_SKIP_XCDS")) h->queue_skip_xcds = (uint32_t)strtoul(e, nullptr, 0) & 0xFFu;
    if (const char* e = getenv("RMJ_QUEUE_TAIL")) h->queue_tail = atoi(e) != 0;
    if (const char* e = getenv("RMJ_QUEUE_MIN_CHUNK")) h->queue_min_chunk = atoi(e) > 0 ? atoi(e) : 1;
    if (const char* e = getenv("RMJ_HEAVY_FIRST")) h->heavy_first = atoi(e);
    h->rows_pw = cfg->n_games <= RMJ_ROWS1_MAX_GAMES ? 1u : (cfg->n_games <= RMJ_ROWS2_MAX_GAMES ? 2u : 4u);
    if (const char* e = getenv("RMJ_ROWS")) { const int r = atoi(e); if (r == 1 || r == 2 || r == 4) h->rows_pw = (uint32_t)r; }
    if (const char* e = getenv("RMJ_ENC_STREAMS")) h->enc_streams = atoi(e);
    if (const char* e = getenv("RMJ_ENC_PARTS_QUAD")) h->enc_parts_quad = atoi(e) != 0;
    if (const char* e = getenv("RMJ_ENC_FUSED")) h->enc_fused = atoi(e);
    const size_t B = cfg->n_games;
    Env& d = h->d;
    HIPCHK(hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
    h->stream = h->own_stream;
    HIPCHK(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
    // (the side streams of the split paths are created when such a path first runs: ensure_side_streams)
    // the two events of rmj_time_rollout* / rmj_bench_rollout: here, not at their first use - the driver's timed region IS their first use, and two
    // hipEventCreate calls between its host timestamps cost the 20-step window ~4 % (round 6, scripts/r06_window_order.py: first window 1.65-1.69 G, later ones 1.76 G)
    for (int i = 0; i < 2; i++) HIPCHK(hipEventCreate(&h->ev_time[i]));
    HIPCHK(hipMalloc(&d.core, B * sizeof(GState)));
    HIPCHK(hipMalloc(&d.wall, B * RMJ_WALL_STRIDE));
    HIPCHK(hipMalloc(&d.legal, B * 4 * RMJ_MAX_LEGAL * sizeof(uint64_t)));
    HIPCHK(hipMalloc(&d.nlegal, B * 4));
    HIPCHK(hipMalloc(&d.mask, B * 4 * 82));
    HIPCHK(hipMalloc(&d.waits, B * 4 * sizeof(uint64_t)));
    HIPCHK(hipMalloc(&d.status, B * sizeof(uint32_t)));
    HIPCHK(hipMalloc(&d.events, B * (size_t)r2 * sizeof(RmjEvent)));
    HIPCHK(hipMalloc(&d.win, B * 4 * sizeof(RmjWinResult)));
    HIPCHK(hipMemsetAsync(d.win, 0, B * 4 * sizeof(RmjWinResult), h->stream));
    HIPCHK(hipMalloc(&d.wall_dg, B * 32));
    HIPCHK(hipMemsetAsync(d.wall_dg, 0, B * 32, h->stream));
    HIPCHK(hipMalloc(&h->d_actions, B * 4 * sizeof(uint64_t)));
    HIPCHK(hipMalloc(&h->d_counter, sizeof(unsigned long long)));
    {   // The ticket rollout (k_step4_queue) hands a quad from wave to wave through ONE XCD's L2 and keys its eight queues by
        // HW_REG_XCC_ID & 7: on a part or partition mode that reports ids >= 8 two XCDs would share a queue and the hand-over
        // would cross L2s without a write-back.  A probe launch (more blocks than any dispatcher keeps on one XCD) records the
        // largest id; tickets are used only when it is <= 7 (rollout_queued), k_step4<true> otherwise.
        HIPCHK(hipMemsetAsync(h->d_counter, 0, sizeof(unsigned long long), h->stream));
        hipLaunchKernelGGL(k_probe_xcc, dim3(4096), dim3(64), 0, h->stream, h->d_counter);
        HIPCHK(hipGetLastError());
        unsigned long long m = 0;
        HIPCHK(hipMemcpyAsync(&m, h->d_counter, sizeof(m), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        h->max_xcc_id = (uint32_t)m;
    }
    HIPCHK(hipMalloc(&h->d_obs_offs, (B + (B + 1023) / 1024) * sizeof(uint32_t)));   // per game + per scan block
    HIPCHK(hipMemsetAsync(d.legal, 0, B * 4 * RMJ_MAX_LEGAL * sizeof(uint64_t), h->stream));
    HIPCHK(hipMemsetAsync(d.nlegal, 0, B * 4, h->stream));
    HIPCHK(hipMemsetAsync(d.mask, 0, B * 4 * 82, h->stream));
    HIPCHK(hipMemsetAsync(d.events, 0, B * (size_t)r2 * sizeof(RmjEvent), h->stream));
    d.ring_mask = r2 - 1;
    d.n_games = cfg->n_games;
    d.rule_bits = cfg->rule_bits;
    d.game_mode = cfg->game_mode;
    d.skip_log = cfg->skip_mjai_logging;
    d.ctor_round_wind = cfg->round_wind;
    d.game_offset = cfg->game_offset;
    d.enc_stride = (uint32_t)(ENC_CH * (cfg->game_mode >= 3 ? ENC_W3 : ENC_W4));
    d.pad_ = 0;
    int rc;
    if ((rc = shanten_tables_for(cfg->device, &d.sh))) return rc;
    HIPCHK(hipMalloc(&h->d_env, sizeof(Env)));
    HIPCHK(hipMemcpy(h->d_env, &d, sizeof(Env), hipMemcpyHostToDevice));
    {
        float decay[RMJ_MAX_DISCARDS];
        for (int a = 0; a < RMJ_MAX_DISCARDS; a++) decay[a] = expf(-0.2f * (float)a);  // same call as the reference's f32::exp
        HIPCHK(hipMalloc(&h->d_decay, sizeof(decay)));
        HIPCHK(hipMemcpy(h->d_decay, decay, sizeof(decay), hipMemcpyHostToDevice));
    }
    ResetArgs A;
    memset(&A, 0, sizeof(A));
    A.is_ctor = 1;
    A.base_seed = cfg->base_seed;
    if (cfg->seeds) {
        HIPCHK(hipMalloc(d_seeds_out, B * sizeof(uint64_t)));
        HIPCHK(hipMemcpy(*d_seeds_out, cfg->seeds, B * sizeof(uint64_t), hipMemcpyHostToDevice));
        A.seeds = *d_seeds_out;
    }
    if (cfg->game_mode >= 3) hipLaunchKernelGGL(rmj3::k_reset, game_grid(cfg->n_games), dim3(256), 0, h->stream, (const Env*)h->d_env, A);
    else hipLaunchKernelGGL(rmj4::k_reset, game_grid(cfg->n_games), dim3(256), 0, h->stream, (const Env*)h->d_env, A);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    return RMJ_OK;
}
int rmj_create(const RmjConfig* cfg, rmj_handle* out) {
    if (!cfg || !out || cfg->n_games == 0) return fail(RMJ_ERR_ARG, "bad config");
    if (cfg->game_mode > 5) return fail(RMJ_ERR_ARG, "game_mode must be 0..5");
    int rc = ensure_device(cfg->device);
    if (rc) return rc;
    rmj_env* h = new rmj_env();
    h->cfg = *cfg;
    memset(&h->d, 0, sizeof(h->d));
    uint64_t* d_seeds = nullptr;
    rc = create_impl(h, cfg, &d_seeds);
    if (d_seeds) hipFree(d_seeds);
    h->cfg.seeds = nullptr;
    if (rc) {  // nothing allocated so far outlives a failed constructor (an OOM at 524 288 games would strand GBs)
        const std::string keep = g_err;
        rmj_destroy(h);
        g_err = keep;
        return rc;
    }
    *out = h;
    return RMJ_OK;
}

int rmj_destroy(rmj_handle h) {
    if (!h) return RMJ_OK;
    hipSetDevice(h->cfg.device);
    if (h->stream) hipStreamSynchronize(h->stream);
    if (h->h_pin) hipHostFree(h->h_pin);
    hipFree(h->d.core); hipFree(h->d.wall); hipFree(h->d.wall_dg); hipFree(h->d.legal); hipFree(h->d.nlegal); hipFree(h->d_decay); if (h->d_scratch) hipFree(h->d_scratch); hipFree(h->d.mask);
    hipFree(h->d.waits); hipFree(h->d.status); hipFree(h->d.events); hipFree(h->d.win); hipFree(h->d_actions); hipFree(h->d_counter); hipFree(h->d_obs_offs); hipFree(h->d_env); hipFree(h->d_qheads);   // (d_qdone lives in the same allocation)
    hipFree(h->d_ev_lost); hipFree(h->d_track); hipFree(h->d_heavy);
    for (int i = 0; i < 2; i++) if (h->ev_time[i]) hipEventDestroy(h->ev_time[i]);
    if (h->own_stream) hipStreamDestroy(h->own_stream);
    for (int i = 0; i < RMJ_MAX_ROLLOUT_STREAMS - 1; i++) {
        if (h->xstream[i]) { hipStreamSynchronize(h->xstream[i]); hipStreamDestroy(h->xstream[i]); }
        if (h->ev_join[i]) hipEventDestroy(h->ev_join[i]);
    }
    if (h->ev_fork) hipEventDestroy(h->ev_fork);
    delete h;
    return RMJ_OK;
}

int rmj_clone(rmj_handle h, rmj_handle* out) {
    if (!h || !out) return fail(RMJ_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(hipStreamSynchronize(h->stream));   // (before anything is allocated: an early return below must not strand the copy)
    RmjConfig cfg = h->cfg;
    cfg.seeds = nullptr;
    cfg.event_ring = h->d.ring_mask + 1u;
    rmj_handle c = nullptr;
    int rc = rmj_create(&cfg, &c);
    if (rc) return rc;
    c->want_streams = h->want_streams;
    c->quad = h->quad;
    c->queue_chunk = h->queue_chunk;
    c->queue_force = h->queue_force;
    c->queue_min_chunk = h->queue_min_chunk;
    c->queue_tail = h->queue_tail;
    c->rows_pw = h->rows_pw;
    c->heavy_first = h->heavy_first;
    c->enc_streams = h->enc_streams;
    c->enc_parts_quad = h->enc_parts_quad;
    c->enc_fused = h->enc_fused;
    if (h->d.enc_stride != c->d.enc_stride) { int rc2 = rmj_set_encode_row_stride(c, h->d.enc_stride); if (rc2) { rmj_destroy(c); return rc2; } }
    const size_t B = h->cfg.n_games, ring = (size_t)h->d.ring_mask + 1u;
    const struct { void* dst; const void* src; size_t bytes; } slabs[] = {
        {c->d.core, h->d.core, B * sizeof(GState)}, {c->d.wall, h->d.wall, B * RMJ_WALL_STRIDE},
        {c->d.legal, h->d.legal, B * 4 * RMJ_MAX_LEGAL * sizeof(uint64_t)}, {c->d.nlegal, h->d.nlegal, B * 4}, {c->d.mask, h->d.mask, B * 4 * 82},
        {c->d.waits, h->d.waits, B * 4 * sizeof(uint64_t)}, {c->d.status, h->d.status, B * sizeof(uint32_t)},
        {c->d.events, h->d.events, B * ring * sizeof(RmjEvent)}, {c->d.win, h->d.win, B * 4 * sizeof(RmjWinResult)},
        {c->d.wall_dg, h->d.wall_dg, B * 32}};
    for (const auto& s : slabs) {
        if (hipMemcpyAsync(s.dst, s.src, s.bytes, hipMemcpyDeviceToDevice, c->stream) != hipSuccess) {
            rmj_destroy(c);
            return fail(RMJ_ERR_HIP, "rmj_clone: device copy failed");
        }
    }
    if (hipStreamSynchronize(c->stream) != hipSuccess) {
        rmj_destroy(c);
        return fail(RMJ_ERR_HIP, "rmj_clone: device copy failed");
    }
    *out = c;
    return RMJ_OK;
}

int rmj_copy_games(rmj_handle dst, const uint32_t* dst_idx, rmj_handle src, const uint32_t* src_idx, uint32_t n) {
    DevTmp tmp;
    if (!dst || !src || (n && (!dst_idx || !src_idx))) return fail(RMJ_ERR_ARG, "null argument");
    if (dst->cfg.device != src->cfg.device || (dst->cfg.game_mode >= 3) != (src->cfg.game_mode >= 3) || dst->d.ring_mask != src->d.ring_mask)
        return fail(RMJ_ERR_ARG, "rmj_copy_games: the handles must share device, player count and event ring size");
    if (n == 0) return RMJ_OK;
    for (uint32_t i = 0; i < n; i++)
        if (dst_idx[i] >= dst->cfg.n_games || src_idx[i] >= src->cfg.n_games) return fail(RMJ_ERR_ARG, "rmj_copy_games: game index out of range");
    HIPCHK(hipSetDevice(dst->cfg.device));
    uint32_t *d_a, *d_b;
    int rc;
    if ((rc = tmp.upload(dst_idx, n, &d_a)) || (rc = tmp.upload(src_idx, n, &d_b))) return rc;
    if (src != dst) HIPCHK(hipStreamSynchronize(src->stream));
    if (dst->cfg.game_mode >= 3) hipLaunchKernelGGL(rmj3::k_copy_games, dim3(n), dim3(64), 0, dst->stream, (const Env*)dst->d_env, (const Env*)src->d_env, d_a, d_b, n);
    else hipLaunchKernelGGL(rmj4::k_copy_games, dim3(n), dim3(64), 0, dst->stream, (const Env*)dst->d_env, (const Env*)src->d_env, d_a, d_b, n);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(dst->stream));
    return RMJ_OK;
}

int rmj_copy_games_device(rmj_handle dst, const uint32_t* d_dst_idx, rmj_handle src, const uint32_t* d_src_idx, uint32_t n) {
    if (!dst || !src || (n && (!d_dst_idx || !d_src_idx))) return fail(RMJ_ERR_ARG, "null argument");
    if (dst->cfg.device != src->cfg.device || (dst->cfg.game_mode >= 3) != (src->cfg.game_mode >= 3) || dst->d.ring_mask != src->d.ring_mask)
        return fail(RMJ_ERR_ARG, "rmj_copy_games_device: the handles must share device, player count and event ring size");
    if (n == 0) return RMJ_OK;
    HIPCHK(hipSetDevice(dst->cfg.device));
    if (src != dst && src->stream != dst->stream) HIPCHK(hipStreamSynchronize(src->stream));
    if (dst->cfg.game_mode >= 3) hipLaunchKernelGGL(rmj3::k_copy_games, dim3(n), dim3(64), 0, dst->stream, (const Env*)dst->d_env, (const Env*)src->d_env, d_dst_idx, d_src_idx, n);
    else hipLaunchKernelGGL(rmj4::k_copy_games, dim3(n), dim3(64), 0, dst->stream, (const Env*)dst->d_env, (const Env*)src->d_env, d_dst_idx, d_src_idx, n);
    HIPCHK(hipGetLastError());
    return RMJ_OK;
}

int rmj_reset(rmj_handle h, const uint8_t* select, const uint8_t* walls, const uint8_t* oya, const uint8_t* round_wind,
              const int32_t* scores, const uint8_t* honba, const uint32_t* kyotaku) {
    DevTmp tmp;
    if (!h) return fail(RMJ_ERR_ARG, "null handle");
    HIPCHK(hipSetDevice(h->cfg.device));
    const size_t B = h->cfg.n_games;
    uint8_t *d_sel, *d_walls, *d_oya, *d_rw, *d_honba;
    int32_t* d_sc;
    uint32_t* d_ky;
    int rc;
    if ((rc = tmp.upload(select, B, &d_sel))) return rc;
    if ((rc = tmp.upload(walls, B * 136, &d_walls))) return rc;
    if ((rc = tmp.upload(oya, B, &d_oya))) return rc;
    if ((rc = tmp.upload(round_wind, B, &d_rw))) return rc;
    if ((rc = tmp.upload(scores, B * 4, &d_sc))) return rc;
    if ((rc = tmp.upload(honba, B, &d_honba))) return rc;
    if ((rc = tmp.upload(kyotaku, B, &d_ky))) return rc;
    ResetArgs A;
    memset(&A, 0, sizeof(A));
    A.select = d_sel; A.walls = d_walls; A.oya = d_oya; A.round_wind = d_rw; A.scores = d_sc; A.honba = d_honba; A.kyotaku = d_ky;
    if (h->cfg.game_mode >= 3) hipLaunchKernelGGL(rmj3::k_reset, game_grid(h->cfg.n_games), dim3(256), 0, h->stream, (const Env*)h->d_env, A);
    else hipLaunchKernelGGL(rmj4::k_reset, game_grid(h->cfg.n_games), dim3(256), 0, h->stream, (const Env*)h->d_env, A);
    if (h->d_track)   // the round tracker must not read the reset as the end of a round (rmj_round_track_device)
        hipLaunchKernelGGL(k_track_mark, dim3((h->cfg.n_games + 255u) / 256u), dim3(256), 0, h->stream, (uint8_t*)h->d_track + B * 37, (const uint8_t*)d_sel, 0u, h->cfg.n_games);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    return RMJ_OK;
}

// one k_step launch over games [g0, g1) of the handle
static inline void launch_step_range(rmj_env* h, hipStream_t st, const uint64_t* d_actions, uint64_t policy_seed, uint32_t flags,
                                     uint32_t g0, uint32_t g1, bool allow_quad = true) {
    const bool greedy = (flags & STEP_F_GREEDY) != 0u;   // (four-games-per-wave kernels only: the callers check h->quad)
#ifdef RMJ_TUNE_LDS
    static const unsigned extra_lds = getenv("RMJ_EXTRA_LDS") ? (unsigned)atoi(getenv("RMJ_EXTRA_LDS")) : 0u;  // occupancy experiments
#else
    const unsigned extra_lds = 0u;
#endif
    if (h->quad && allow_quad) {   // four games per wave (device policy, packed actions or action ids); small batches: two or one (rows_pw)
        const uint32_t rows = h->rows_pw;
        flags |= (rows == 4u ? 0u : rows) << STEP_F_ROWS_SHIFT;
        const uint32_t units = (g1 - g0 + rows - 1u) / rows;
        // whole-batch launches in heavy-first order (HeavyOrder): the previous launch's notes name the units that will end a round,
        // restart or may settle a Ron - they get the first blocks
        HeavyOrder ho = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0u};
        if (h->heavy_first && g0 == 0u && g1 == h->cfg.n_games && units >= 2048u) {
            const uint32_t front = units / 4u;
            const size_t o_list = 384, o_flag = o_list + 2 * (size_t)front * 4, bytes = o_flag + 2 * (size_t)units;   // three counters, a line each
            if (!h->d_heavy) {
                if (hipMalloc(&h->d_heavy, bytes) == hipSuccess) hipMemsetAsync(h->d_heavy, 0, bytes, st);
                else h->d_heavy = nullptr;
            }
            if (h->d_heavy) {
                uint8_t* b = (uint8_t*)h->d_heavy;
                const uint32_t k = h->heavy_phase, in = k & 1u, out = in ^ 1u;
                ho.in_cnt = (const uint32_t*)(b + 128 * (k % 3u)); ho.out_cnt = (uint32_t*)(b + 128 * ((k + 1u) % 3u)); ho.zero_cnt = (uint32_t*)(b + 128 * ((k + 2u) % 3u));
                ho.in_list = (const uint32_t*)(b + o_list) + (size_t)front * in; ho.out_list = (uint32_t*)(b + o_list) + (size_t)front * out;
                ho.in_flag = b + o_flag + (size_t)units * in; ho.out_flag = b + o_flag + (size_t)units * out;
                ho.front = front;
                h->heavy_phase = (k + 1u) % 6u;   // (period of the counter and the list rotation)
            }
        }
        const dim3 grid(units + ho.front);
        if (h->cfg.game_mode >= 3) {
            if (greedy) hipLaunchKernelGGL((rmj3::k_step4<false, 1>), grid, dim3(64), 0, st, (const Env*)h->d_env, policy_seed, flags, g0, g1, 1u, d_actions, ho);
            else hipLaunchKernelGGL((rmj3::k_step4<false, 0>), grid, dim3(64), 0, st, (const Env*)h->d_env, policy_seed, flags, g0, g1, 1u, d_actions, ho);
        } else {
            if (greedy) hipLaunchKernelGGL((rmj4::k_step4<false, 1>), grid, dim3(64), 0, st, (const Env*)h->d_env, policy_seed, flags, g0, g1, 1u, d_actions, ho);
            else hipLaunchKernelGGL((rmj4::k_step4<false, 0>), grid, dim3(64), 0, st, (const Env*)h->d_env, policy_seed, flags, g0, g1, 1u, d_actions, ho);
        }
        return;
    }
    if (h->cfg.game_mode >= 3) hipLaunchKernelGGL(rmj3::k_step, step_grid(g1 - g0), dim3(64 * RMJ_STEP_WPB), extra_lds, st, (const Env*)h->d_env, d_actions, policy_seed, flags, g0, g1);
    else hipLaunchKernelGGL(rmj4::k_step, step_grid(g1 - g0), dim3(64 * RMJ_STEP_WPB), extra_lds, st, (const Env*)h->d_env, d_actions, policy_seed, flags, g0, g1);
}
int rmj_step_device(rmj_handle h, const rmj_action_t* d_actions) {
    if (!h || !d_actions) return fail(RMJ_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(h->cfg.device));
    launch_step_range(h, h->stream, (const uint64_t*)d_actions, 0ull, 0u, 0u, h->cfg.n_games);
    HIPCHK(hipGetLastError());
    return RMJ_OK;
}
// Trainer-side entry (row N4): the policy's categorical outputs (action ids of the 82- / 60-way space, -1 = no action),
// resident on the device, are mapped to the first legal action with that id (Observation.find_action,
// observation/python.rs:119-122) inside the step kernel.  An id without a legal action is an illegal action (chombo).
int rmj_step_ids_device(rmj_handle h, const int32_t* d_action_ids, int auto_reset) {
    if (!h || !d_action_ids) return fail(RMJ_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(h->cfg.device));
    const uint32_t flags = STEP_F_IDS | (auto_reset ? STEP_F_AUTORESET : 0u);
    launch_step_range(h, h->stream, reinterpret_cast<const uint64_t*>(d_action_ids), 0ull, flags, 0u, h->cfg.n_games);
    HIPCHK(hipGetLastError());
    return RMJ_OK;
}
// rmj_step_ids_device + rmj_encode_device(only_active = 2) as ONE launch (k_step4_act_enc): the step of every game under the policy's
// action ids, then Observation.encode() of the seats that are to act next into the resident tensor d_out [n][4][74][W].
int rmj_step_ids_encode_device(rmj_handle h, const int32_t* d_action_ids, int auto_reset, float* d_out) {
    if (!h || !d_action_ids || !d_out) return fail(RMJ_ERR_ARG, "null argument");
    if (!h->quad) return fail(RMJ_ERR_ARG, "rmj_step_ids_encode_device runs in the four-games-per-wave kernels (RMJ_STEP4=0 selects the one-game kernel)");
    HIPCHK(hipSetDevice(h->cfg.device));
    const uint32_t flags = STEP_F_IDS | (auto_reset ? STEP_F_AUTORESET : 0u);
    const uint32_t n = h->cfg.n_games;
    const dim3 grid((n + 3u) / 4u);
    if (h->cfg.game_mode >= 3) hipLaunchKernelGGL(rmj3::k_step4_act_enc, grid, dim3(64), 0, h->stream, (const Env*)h->d_env, flags, 0u, n, reinterpret_cast<const uint64_t*>(d_action_ids), d_out);
    else hipLaunchKernelGGL(rmj4::k_step4_act_enc, grid, dim3(64), 0, h->stream, (const Env*)h->d_env, flags, 0u, n, reinterpret_cast<const uint64_t*>(d_action_ids), d_out);
    HIPCHK(hipGetLastError());
    return RMJ_OK;
}
// rmj_sample_ids_device + rmj_step_ids_encode_device as ONE launch (k_step4_sample_enc): every wave draws the ids of its own four games
// (the same keyed draw: identical ids, written to d_ids for the caller), steps them and encodes the seats that act next.
int rmj_step_sample_encode_device(rmj_handle h, const float* d_logits, uint32_t stride, uint64_t seed, int auto_reset, int32_t* d_ids, float* d_out) {
    if (!h || !d_ids || !d_out) return fail(RMJ_ERR_ARG, "null argument");
    if (!h->quad) return fail(RMJ_ERR_ARG, "rmj_step_sample_encode_device runs in the four-games-per-wave kernels (RMJ_STEP4=0 selects the one-game kernel)");
    const uint32_t A = h->cfg.game_mode >= 3 ? RMJ_ACTION_SPACE_3P : RMJ_ACTION_SPACE_4P;
    if (d_logits && stride < A) return fail(RMJ_ERR_ARG, "logits row shorter than the action space");
    HIPCHK(hipSetDevice(h->cfg.device));
    const uint32_t flags = STEP_F_IDS | (auto_reset ? STEP_F_AUTORESET : 0u);
    const uint32_t n = h->cfg.n_games;
    const dim3 grid((n + 3u) / 4u);
    if (h->cfg.game_mode >= 3) hipLaunchKernelGGL(rmj3::k_step4_sample_enc, grid, dim3(64), 0, h->stream, (const Env*)h->d_env, flags, 0u, n, d_logits, stride, seed, d_ids, d_out);
    else hipLaunchKernelGGL(rmj4::k_step4_sample_enc, grid, dim3(64), 0, h->stream, (const Env*)h->d_env, flags, 0u, n, d_logits, stride, seed, d_ids, d_out);
    HIPCHK(hipGetLastError());
    return RMJ_OK;
}
int rmj_sample_ids_device(rmj_handle h, const float* d_logits, uint32_t stride, uint64_t seed, int32_t* d_ids) {
    if (!h || !d_ids) return fail(RMJ_ERR_ARG, "null argument");
    const uint32_t A = h->cfg.game_mode >= 3 ? RMJ_ACTION_SPACE_3P : RMJ_ACTION_SPACE_4P;
    if (d_logits && stride < A) return fail(RMJ_ERR_ARG, "logits row shorter than the action space");
    HIPCHK(hipSetDevice(h->cfg.device));
    const uint32_t n = h->cfg.n_games;
    hipLaunchKernelGGL(k_sample_ids, dim3((n + 15) / 16), dim3(256), 0, h->stream, h->d, d_logits, stride, seed, d_ids);
    HIPCHK(hipGetLastError());
    return RMJ_OK;
}
int rmj_device_views(rmj_handle h, RmjDeviceViews* out) {
    if (!h || !out) return fail(RMJ_ERR_ARG, "null argument");
    out->n_games = h->cfg.n_games;
    out->status = h->d.status;
    out->nlegal = h->d.nlegal;
    out->legal = h->d.legal;
    out->mask = h->d.mask;
    out->waits = h->d.waits;
    out->stream = (void*)h->stream;
    return RMJ_OK;
}
int rmj_scores_device(rmj_handle h, int32_t* d_scores, uint32_t* d_event_counts) {
    if (!h || !d_scores) return fail(RMJ_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(h->cfg.device));
    const uint32_t n = h->cfg.n_games;
    hipLaunchKernelGGL(k_gather_scores, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->d.core, n, d_scores, d_event_counts);
    HIPCHK(hipGetLastError());
    return RMJ_OK;
}
// RiichiEnv.points(rule_name) (env.rs:691-727) of every game on the device: rule 0 = "basic", 1 = "ouza-tyoujyo", 2 = "ouza-normal"
// (3P knows "basic" only, like the reference); d_points [n][4] f64, seats beyond the player count get 0.  Asynchronous on the
// handle's stream: the reward a trainer-side loop reads without leaving the GPU.
int rmj_points_device(rmj_handle h, int rule, double* d_points) {
    if (!h || !d_points) return fail(RMJ_ERR_ARG, "null argument");
    const bool sanma = h->cfg.game_mode >= 3;
    if (rule < 0 || rule > (sanma ? 0 : 2)) return fail(RMJ_ERR_ARG, sanma ? "Unknown preset rule for 3P" : "Unknown preset rule");
    HIPCHK(hipSetDevice(h->cfg.device));
    const uint32_t n = h->cfg.n_games;
    static const double UMA4[3][4] = {{50.0, 10.0, -10.0, -50.0}, {100.0, 40.0, -40.0, -100.0}, {50.0, 20.0, -20.0, -50.0}};
    const double w = sanma ? 1.0 : (rule == 0 ? 1.0 : 0.0), base = sanma ? 35000.0 : 25000.0;
    const double u0 = sanma ? 40.0 : UMA4[rule][0], u1 = sanma ? 0.0 : UMA4[rule][1], u2 = sanma ? -40.0 : UMA4[rule][2], u3 = sanma ? 0.0 : UMA4[rule][3];
    hipLaunchKernelGGL(k_points, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->d.core, n, sanma ? 3 : 4, w, base, u0, u1, u2, u3, d_points);
    HIPCHK(hipGetLastError());
    return RMJ_OK;
}
int rmj_get_points(rmj_handle h, int rule, double* points) {
    DevTmp tmp;
    if (!h || !points) return fail(RMJ_ERR_ARG, "null argument");
    double* d;
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(tmp.alloc(&d, (size_t)h->cfg.n_games * 4 * sizeof(double)));
    int rc = rmj_points_device(h, rule, d);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(points, d, (size_t)h->cfg.n_games * 4 * sizeof(double), hipMemcpyDeviceToHost));
    return RMJ_OK;
}
// Row stride of the base encoder's outputs (rmj_encode(_device), rmj_encode_compact_device, rmj_step_random_encode, rmj_step_ids_encode_device):
// every (game, seat) row - 74 x W floats - starts `floats` floats after the previous one; 0 restores the dense layout (74 x W).  Padding
// the rows to a multiple of 256 B (2 048 floats in 3P, 2 560 in 4P) costs 2 % more memory and lets the acting seats' rows - one row in
// four of the tensor - leave at 1.3-1.4 x the rate: unaligned rows of 7 992 / 10 064 B are written at 3.8 TB/s, aligned ones at 5.0-5.3
// (torch fills of the same pattern, scripts/micro/row_stride_fill.py).  The pad floats are never written.
int rmj_set_encode_row_stride(rmj_handle h, uint32_t floats) {
    if (!h) return fail(RMJ_ERR_ARG, "null handle");
    const uint32_t dense = (uint32_t)(ENC_CH * (h->cfg.game_mode >= 3 ? ENC_W3 : ENC_W4));
    if (floats == 0) floats = dense;
    if (floats < dense || (floats & 1u)) return fail(RMJ_ERR_ARG, "row stride must be an even number of floats >= 74 x W");
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->d.enc_stride = floats;
    HIPCHK(hipMemcpy(h->d_env, &h->d, sizeof(Env), hipMemcpyHostToDevice));
    return RMJ_OK;
}
int rmj_set_stream(rmj_handle h, void* stream, int own) {
    if (!h) return fail(RMJ_ERR_ARG, "null handle");
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(hipStreamSynchronize(h->stream));  // everything issued so far is complete before the order changes hands
    h->stream = own ? h->own_stream : (hipStream_t)stream;  // (a NULL stream is the device's default stream)
    return RMJ_OK;
}
int rmj_sync(rmj_handle h) {
    if (!h) return fail(RMJ_ERR_ARG, "null handle");
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(hipStreamSynchronize(h->stream));
    return RMJ_OK;
}
int rmj_step(rmj_handle h, const rmj_action_t* actions) {
    if (!h || !actions) return fail(RMJ_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(h->d_actions, actions, (size_t)h->cfg.n_games * 4 * sizeof(uint64_t), hipMemcpyHostToDevice));
    return rmj_step_device(h, h->d_actions);
}
// streams a device rollout of n_steps steps uses (RMJ_STEP_STREAMS=1 keeps everything on one stream)
static int rollout_streams(const rmj_env* h, uint32_t n_steps) {
    const int want = h->want_streams;
    if (h->quad >= 2 && n_steps >= 2 && want >= 2) return 1;   // the fused rollout: one launch, no parts
    if (want < 2 || n_steps < 2 || h->cfg.n_games < RMJ_SPLIT_MIN_GAMES) return 1;
    int k = want > RMJ_MAX_ROLLOUT_STREAMS ? RMJ_MAX_ROLLOUT_STREAMS : want;
    const int fit = (int)(h->cfg.n_games / RMJ_SPLIT_MIN_PART);
    return k > fit ? fit : k;
}
// Side streams (and their join events) of the paths that split a batch over k streams - the per-step rollouts of RMJ_STEP4=0/1 builds and the unfused
// step + encode rollout.  Round 6: created on first use, not in rmj_create: the default paths (fused rollouts, one stream) never touch them, and every live
// stream makes hipDeviceSynchronize slower - with seven idle side streams the synchronisation behind the driver's 20-step window took ~45 us of a 0.78 ms
// region (scripts/r06_window_order.py: the same window between stream synchronisations ran at 1.76 G env.step/s, between device synchronisations at 1.66 G).
static int ensure_side_streams(rmj_env* h, int k) {
    for (int i = 0; i < k - 1 && i < RMJ_MAX_ROLLOUT_STREAMS - 1; i++) {
        if (!h->xstream[i]) HIPCHK(hipStreamCreateWithFlags(&h->xstream[i], hipStreamNonBlocking));
        if (!h->ev_join[i]) HIPCHK(hipEventCreateWithFlags(&h->ev_join[i], hipEventDisableTiming));
    }
    return RMJ_OK;
}
// Does a fused rollout of n_steps run as tickets (k_step4_queue)?  Worth it when the batch is more than one and fewer than eight
// chip-fulls of waves: below, every quad is resident at once and there is no tail; far above, the tail is a small share and the
// chunk hand-overs cost more than it (524 288 games: -2 %).
// steps per ticket: the configured chunk, shorter for a short rollout (its tail is one chunk long: at least 16 chunks per quad)
static uint32_t rollout_chunk(const rmj_env* h, uint32_t n_steps) {
    uint32_t cap = (uint32_t)h->queue_chunk;
    const uint32_t lo = (uint32_t)h->queue_min_chunk, fine = n_steps / 16u < lo ? lo : n_steps / 16u;
    if (cap < (n_steps + 39u) / 40u) cap = (n_steps + 39u) / 40u;   // at most 64 tickets per quad (q_ticket_plan)
    return fine < cap ? fine : cap;
}
static bool rollout_queued(rmj_env* h, uint32_t n_steps, int pol) {
    const uint32_t quads = (h->cfg.n_games + 3u) / 4u;
    if (!(h->quad >= 2 && n_steps >= 2 && h->want_streams >= 2) || h->queue_chunk <= 0 || n_steps < 2u * rollout_chunk(h, n_steps)) return false;
    if (h->max_xcc_id > 7u) return false;   // more XCC ids than queues: no single L2 per queue (see rmj_create)
    pol = pol == 1 ? 1 : 0;
    if (h->q_slots_pol[pol] == 0) {
        int per_cu = 0, cus = 0;
        const bool sanma = h->cfg.game_mode >= 3;
        const hipError_t e = pol == 1 ? (sanma ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, rmj3::k_step4_queue<1>, 64, 0)
                                               : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, rmj4::k_step4_queue<1>, 64, 0))
                                      : (sanma ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, rmj3::k_step4_queue<0>, 64, 0)
                                               : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, rmj4::k_step4_queue<0>, 64, 0));
        if (e != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->cfg.device) != hipSuccess) return false;
        h->q_slots_pol[pol] = (uint32_t)(per_cu > 0 ? per_cu : 1) * (uint32_t)(cus > 0 ? cus : 1);
    }
    const uint32_t slots = h->q_slots_pol[pol];
    if (h->queue_force) return quads >= 64u;   // RMJ_QUEUE_FORCE=1 (tests): any batch with a quad per XCD queue to spare
    return quads > slots && quads < 8u * slots;
}
// device-policy rollout: pol 0 = RandomAgent (rmj_step_random), 1 = the greedy policy (rmj_step_greedy)
#define RMJ_LAUNCH_POL(NS, KERNEL, POL, ...)                                              \
    do {                                                                                  \
        if ((POL) == 1) hipLaunchKernelGGL((NS::KERNEL<1>), __VA_ARGS__);                 \
        else hipLaunchKernelGGL((NS::KERNEL<0>), __VA_ARGS__);                            \
    } while (0)
#define RMJ_LAUNCH_LOOP_POL(NS, POL, ...)                                                 \
    do {                                                                                  \
        if ((POL) == 1) hipLaunchKernelGGL((NS::k_step4<true, 1>), __VA_ARGS__);          \
        else hipLaunchKernelGGL((NS::k_step4<true, 0>), __VA_ARGS__);                     \
    } while (0)
static int step_policy_impl(rmj_handle h, uint64_t policy_seed, uint32_t n_steps, int auto_reset, int pol, uint32_t call_rate) {
    if (!h) return fail(RMJ_ERR_ARG, "null handle");
    HIPCHK(hipSetDevice(h->cfg.device));
    uint32_t flags = STEP_F_RANDOM | (auto_reset ? STEP_F_AUTORESET : 0u);
    if (pol == 1) {
        if (!h->quad) return fail(RMJ_ERR_ARG, "the greedy device policy runs in the four-games-per-wave kernels (RMJ_STEP4=0 selects the one-game kernel)");
        flags |= STEP_F_GREEDY | ((call_rate > 255u ? 255u : call_rate) << 8);
    }
    const uint32_t n = h->cfg.n_games;
    const bool sanma = h->cfg.game_mode >= 3;
    if (h->quad >= 2 && n_steps >= 2 && h->want_streams >= 2) {   // (rmj_set_rollout_streams(h, 1): one launch per step, one stream)
        // four games per wave, the whole rollout in ONE launch: every wave steps its own games n_steps times (k_step4<true>)
        const dim3 grid((n + 3u) / 4u);
        if (rollout_queued(h, n_steps, pol)) {
            // ... or, for a long rollout of a batch that does not fill the chip a whole number of times, in (quad, chunk) tickets
            // ticket counters (one line per XCD) + the quads' ticket counts: one set for this rollout - zeroed once here, re-armed by every
            // k_step4_fixup launch behind its ticket launch, so no memset sits between rollouts and a captured rollout can be replayed (round 6;
            // round 5 alternated two sets from the host, which a graph replay of one launch found exhausted) - and one for the step + encode
            // rollout (which memsets its own); behind them the games' step counts of the running rollout, carried from ticket to ticket
            const size_t set_words = 8 * RMJ_Q_STRIDE + (size_t)grid.x;
            if (!h->d_qheads) {
                HIPCHK(hipMalloc(&h->d_qheads, (3 * set_words + (size_t)grid.x * 4) * sizeof(uint32_t)));
                HIPCHK(hipMemsetAsync(h->d_qheads, 0, 3 * set_words * sizeof(uint32_t), h->stream));
            }
            uint32_t* const heads = h->d_qheads;
            uint32_t* const done = heads + 8 * RMJ_Q_STRIDE;
            uint32_t* const d_qprog = h->d_qheads + 3 * set_words;   // (written before it is read)
            const dim3 gq(grid.x < h->q_slots_pol[pol == 1 ? 1 : 0] ? grid.x : h->q_slots_pol[pol == 1 ? 1 : 0]);
            const dim3 gfix((grid.x + 63u) / 64u);
            const uint32_t chunk = rollout_chunk(h, n_steps);
            if (sanma) {
                RMJ_LAUNCH_POL(rmj3, k_step4_queue, pol, gq, dim3(64), 0, h->stream, (const Env*)h->d_env, policy_seed, flags, n, n_steps, chunk, heads, done, h->queue_skip_xcds, d_qprog, (uint32_t)h->queue_tail);
                RMJ_LAUNCH_POL(rmj3, k_step4_fixup, pol, gfix, dim3(64), 0, h->stream, (const Env*)h->d_env, policy_seed, flags, n, n_steps, done);
            } else {
                RMJ_LAUNCH_POL(rmj4, k_step4_queue, pol, gq, dim3(64), 0, h->stream, (const Env*)h->d_env, policy_seed, flags, n, n_steps, chunk, heads, done, h->queue_skip_xcds, d_qprog, (uint32_t)h->queue_tail);
                RMJ_LAUNCH_POL(rmj4, k_step4_fixup, pol, gfix, dim3(64), 0, h->stream, (const Env*)h->d_env, policy_seed, flags, n, n_steps, done);
            }
            HIPCHK(hipGetLastError());
            return RMJ_OK;
        }
        const uint32_t rows = h->rows_pw;
        const dim3 grid_r((n + rows - 1u) / rows);
        flags |= (rows == 4u ? 0u : rows) << STEP_F_ROWS_SHIFT;
        const HeavyOrder no_order = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0u};
        if (sanma) RMJ_LAUNCH_LOOP_POL(rmj3, pol, grid_r, dim3(64), 0, h->stream, (const Env*)h->d_env, policy_seed, flags, 0u, n, n_steps, (const uint64_t*)nullptr, no_order);
        else RMJ_LAUNCH_LOOP_POL(rmj4, pol, grid_r, dim3(64), 0, h->stream, (const Env*)h->d_env, policy_seed, flags, 0u, n, n_steps, (const uint64_t*)nullptr, no_order);
        HIPCHK(hipGetLastError());
        return RMJ_OK;
    }
    const int k = rollout_streams(h, n_steps);
    if (k >= 2) {
        // games are independent: each part advances n_steps steps on its own stream (header: rmj_step_random)
        if (int rc = ensure_side_streams(h, k)) return rc;
        HIPCHK(hipEventRecord(h->ev_fork, h->stream));
        for (int i = 1; i < k; i++) HIPCHK(hipStreamWaitEvent(h->xstream[i - 1], h->ev_fork, 0));
        for (uint32_t s = 0; s < n_steps; s++)
            for (int i = 0; i < k; i++)
                launch_step_range(h, i ? h->xstream[i - 1] : h->stream, nullptr, policy_seed, flags,
                                  (uint32_t)((uint64_t)n * i / k), (uint32_t)((uint64_t)n * (i + 1) / k));
        for (int i = 1; i < k; i++) {
            HIPCHK(hipEventRecord(h->ev_join[i - 1], h->xstream[i - 1]));
            HIPCHK(hipStreamWaitEvent(h->stream, h->ev_join[i - 1], 0));
        }
    } else {
        for (uint32_t s = 0; s < n_steps; s++) launch_step_range(h, h->stream, nullptr, policy_seed, flags, 0u, n);
    }
    HIPCHK(hipGetLastError());
    return RMJ_OK;
}
int rmj_step_random(rmj_handle h, uint64_t policy_seed, uint32_t n_steps, int auto_reset) {
    return step_policy_impl(h, policy_seed, n_steps, auto_reset, 0, 0u);
}
int rmj_step_greedy(rmj_handle h, uint64_t policy_seed, uint32_t n_steps, int auto_reset, uint32_t call_rate_256) {
    return step_policy_impl(h, policy_seed, n_steps, auto_reset, 1, call_rate_256);
}
// The feature-output rollout of BASELINE configs[4]: every step of the device-policy rollout is followed by Observation.encode()
// of the seats that are to act, written into the resident tensor d_out [n][4][74][W] (only_active as in rmj_encode_device).
// Same results as n_steps x (rmj_step_random(h, seed, 1, auto_reset); rmj_encode_device(h, only_active, d_out)); issued like
// rmj_step_random as up to four parts of the batch on as many streams, each part running step, encode, step, encode ...
int rmj_step_random_encode(rmj_handle h, uint64_t policy_seed, uint32_t n_steps, int auto_reset, int only_active, float* d_out) {
    if (!h || !d_out) return fail(RMJ_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(h->cfg.device));
    const uint32_t flags = STEP_F_RANDOM | (auto_reset ? STEP_F_AUTORESET : 0u);
    const uint32_t n = h->cfg.n_games;
    if (h->enc_fused && h->quad >= 2 && h->want_streams >= 2 && n_steps >= 2 && only_active == 2) {
        // Round 3: ONE launch - every wave steps its four games and writes the rows of the seats that are to act, step after step
        // (k_step4_enc); as (quad, chunk) tickets when the batch is between one and eight chip-fulls of waves (k_step4_queue_enc)
        const bool sanma = h->cfg.game_mode >= 3;
        const dim3 grid((n + 3u) / 4u);
        if (h->q_slots_enc == 0) {
            int per_cu = 0, cus = 0;
            if ((sanma ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, rmj3::k_step4_queue_enc<0>, 64, 0)
                       : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, rmj4::k_step4_queue_enc<0>, 64, 0)) != hipSuccess ||
                hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->cfg.device) != hipSuccess)
                return fail(RMJ_ERR_HIP, "occupancy query failed");
            h->q_slots_enc = (uint32_t)(per_cu > 0 ? per_cu : 1) * (uint32_t)(cus > 0 ? cus : 1);
        }
        const uint32_t chunk = rollout_chunk(h, n_steps);
        const bool queued = h->queue_chunk > 0 && h->max_xcc_id <= 7u && n_steps >= 2u * chunk &&
                            (h->queue_force ? grid.x >= 64u : (grid.x > h->q_slots_enc && grid.x < 8u * h->q_slots_enc));
        if (queued) {
            // ticket counters (one line per XCD) + the quads' chunk counts: one allocation, zeroed by ONE memset in front of every rollout
            const size_t set_words = 8 * RMJ_Q_STRIDE + (size_t)grid.x;   // (three counter sets, the third is this rollout's: step_policy_impl)
            if (!h->d_qheads) {
                HIPCHK(hipMalloc(&h->d_qheads, (3 * set_words + (size_t)grid.x * 4) * sizeof(uint32_t)));
                HIPCHK(hipMemsetAsync(h->d_qheads, 0, 3 * set_words * sizeof(uint32_t), h->stream));
            }
            uint32_t* const e_heads = h->d_qheads + 2 * set_words;
            uint32_t* const e_done = e_heads + 8 * RMJ_Q_STRIDE;
            HIPCHK(hipMemsetAsync(e_heads, 0, set_words * sizeof(uint32_t), h->stream));
            const dim3 gq(grid.x < h->q_slots_enc ? grid.x : h->q_slots_enc);
            if (sanma) {
                hipLaunchKernelGGL((rmj3::k_step4_queue_enc<0>), gq, dim3(64), 0, h->stream, (const Env*)h->d_env, policy_seed, flags, n, n_steps, chunk, e_heads, e_done, h->queue_skip_xcds, d_out);
                hipLaunchKernelGGL((rmj3::k_step4_fixup_enc<0>), grid, dim3(64), 0, h->stream, (const Env*)h->d_env, policy_seed, flags, n, n_steps, (const uint32_t*)e_done, d_out);
            } else {
                hipLaunchKernelGGL((rmj4::k_step4_queue_enc<0>), gq, dim3(64), 0, h->stream, (const Env*)h->d_env, policy_seed, flags, n, n_steps, chunk, e_heads, e_done, h->queue_skip_xcds, d_out);
                hipLaunchKernelGGL((rmj4::k_step4_fixup_enc<0>), grid, dim3(64), 0, h->stream, (const Env*)h->d_env, policy_seed, flags, n, n_steps, (const uint32_t*)e_done, d_out);
            }
        } else if (sanma) {
            hipLaunchKernelGGL((rmj3::k_step4_enc<0>), grid, dim3(64), 0, h->stream, (const Env*)h->d_env, policy_seed, flags, 0u, n, n_steps, d_out);
        } else {
            hipLaunchKernelGGL((rmj4::k_step4_enc<0>), grid, dim3(64), 0, h->stream, (const Env*)h->d_env, policy_seed, flags, 0u, n, n_steps, d_out);
        }
        HIPCHK(hipGetLastError());
        return RMJ_OK;
    }
    // The encoder is bound by its stores, the step by instruction issue: parts of the batch on k streams put the step of one
    // part under the encoder of another (measured, 65 536 3P games: one stream 283 M env.step/s, four parts 383 M with the
    // four-game kernel and 323 M with the one-game kernel).  RMJ_ENC_STREAMS / RMJ_ENC_PARTS_QUAD: experiment knobs.
    int k = h->enc_streams > 0 ? h->enc_streams : h->want_streams;
    const bool parts_quad = h->enc_parts_quad >= 0 ? h->enc_parts_quad != 0 : h->quad != 0;
    if (k > RMJ_MAX_ROLLOUT_STREAMS) k = RMJ_MAX_ROLLOUT_STREAMS;
    if ((int)(n / RMJ_SPLIT_MIN_PART) < k) k = (int)(n / RMJ_SPLIT_MIN_PART);
    if (n_steps < 2 || n < RMJ_SPLIT_MIN_GAMES || k < 2) k = 1;
    if (k >= 2) {
        if (int rc = ensure_side_streams(h, k)) return rc;
        HIPCHK(hipEventRecord(h->ev_fork, h->stream));
        for (int i = 1; i < k; i++) HIPCHK(hipStreamWaitEvent(h->xstream[i - 1], h->ev_fork, 0));
        for (uint32_t s = 0; s < n_steps; s++)
            for (int i = 0; i < k; i++) {
                hipStream_t st = i ? h->xstream[i - 1] : h->stream;
                const uint32_t g0 = (uint32_t)((uint64_t)n * i / k), g1 = (uint32_t)((uint64_t)n * (i + 1) / k);
                launch_step_range(h, st, nullptr, policy_seed, flags, g0, g1, parts_quad);
                launch_encode_base_range(h, st, only_active, d_out, g0, g1);
            }
        for (int i = 1; i < k; i++) {
            HIPCHK(hipEventRecord(h->ev_join[i - 1], h->xstream[i - 1]));
            HIPCHK(hipStreamWaitEvent(h->stream, h->ev_join[i - 1], 0));
        }
    } else {
        for (uint32_t s = 0; s < n_steps; s++) {
            launch_step_range(h, h->stream, nullptr, policy_seed, flags, 0u, n);
            launch_encode_base_range(h, h->stream, only_active, d_out, 0u, n);
        }
    }
    HIPCHK(hipGetLastError());
    return RMJ_OK;
}
int rmj_random_actions(rmj_handle h, uint64_t policy_seed, rmj_action_t* actions) {
    if (!h || !actions) return fail(RMJ_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(h->cfg.device));
    uint32_t n = h->cfg.n_games;
    hipLaunchKernelGGL(k_random_actions, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->d, policy_seed, h->d_actions);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(actions, h->d_actions, (size_t)n * 4 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return RMJ_OK;
}

#define SYNC_FETCH(dst, src, bytes)                                   \
    do {                                                              \
        HIPCHK(hipSetDevice(h->cfg.device));                          \
        HIPCHK(hipStreamSynchronize(h->stream));                      \
        HIPCHK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));    \
    } while (0)

int rmj_get_status(rmj_handle h, uint8_t* active_mask, uint8_t* phase, uint8_t* done) {
    if (!h) return fail(RMJ_ERR_ARG, "null handle");
    std::vector<uint32_t> st(h->cfg.n_games);
    SYNC_FETCH(st.data(), h->d.status, st.size() * 4);
    for (size_t i = 0; i < st.size(); i++) {
        if (active_mask) active_mask[i] = st[i] & 0xFF;
        if (phase) phase[i] = (st[i] >> 8) & 0xFF;
        if (done) done[i] = (st[i] >> 16) & 0xFF;
    }
    return RMJ_OK;
}
int rmj_get_legal(rmj_handle h, rmj_action_t* legal, uint8_t* counts) {
    if (!h) return fail(RMJ_ERR_ARG, "null handle");
    size_t B = h->cfg.n_games;
    if (legal) SYNC_FETCH(legal, h->d.legal, B * 4 * RMJ_MAX_LEGAL * sizeof(uint64_t));
    if (counts) SYNC_FETCH(counts, h->d.nlegal, B * 4);
    return RMJ_OK;
}
// What a host agent loop reads per step, without the 2 KB per game of the full [n][4][64] list slab: one row per seat that is to
// act, in (game, seat) order - index[row] = game * 4 + seat, its list = entries[offsets[row] .. offsets[row + 1]) - gathered on the
// device and brought down through pinned staging memory (~110 B per game instead of 2 055).  n_rows / n_entries report the totals;
// when they exceed the capacities only the first cap_rows rows / cap_entries entries were written (call again with more room).
int rmj_get_legal_compact(rmj_handle h, uint32_t* index, uint32_t* offsets /*[cap_rows + 1]*/, rmj_action_t* entries, uint32_t cap_rows, uint32_t cap_entries,
                          uint32_t* n_rows, uint32_t* n_entries) {
    if (!h || !index || !offsets || !entries || !n_rows || !n_entries) return fail(RMJ_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(h->cfg.device));
    const uint32_t n = h->cfg.n_games, blocks = (n + LC_BLOCK - 1) / LC_BLOCK;
    // device scratch: pre [n][2] | blk [blocks][2] | totals [2] | index [cap_rows] | offs [cap_rows] | entries [cap_entries]
    const size_t o_blk = (size_t)n * 8, o_tot = o_blk + (size_t)blocks * 8, o_idx = o_tot + 16, o_off = o_idx + (size_t)cap_rows * 4;
    const size_t o_ent = (o_off + ((size_t)cap_rows + 1) * 4 + 15) & ~(size_t)15, total = o_ent + (size_t)cap_entries * 8;
    void* sp;
    int rc = scratch_for(h, total, &sp);
    if (rc) return rc;
    uint8_t* base = (uint8_t*)sp;
    uint32_t *pre = (uint32_t*)base, *blk = (uint32_t*)(base + o_blk), *tot = (uint32_t*)(base + o_tot), *d_idx = (uint32_t*)(base + o_idx), *d_off = (uint32_t*)(base + o_off);
    uint64_t* d_ent = (uint64_t*)(base + o_ent);
    hipLaunchKernelGGL(k_lc_count, dim3(blocks), dim3(LC_BLOCK), 0, h->stream, (const uint32_t*)h->d.status, (const uint8_t*)h->d.nlegal, n, pre, blk);
    hipLaunchKernelGGL(k_lc_scan, dim3(1), dim3(64), 0, h->stream, blk, blocks, tot);
    hipLaunchKernelGGL(k_lc_gather, dim3(blocks), dim3(LC_BLOCK), 0, h->stream, (const uint32_t*)h->d.status, (const uint8_t*)h->d.nlegal, (const uint64_t*)h->d.legal, n,
                       (const uint32_t*)pre, (const uint32_t*)blk, cap_rows, cap_entries, d_idx, d_off, d_ent);
    HIPCHK(hipGetLastError());
    const size_t need_pin = 16 + (size_t)cap_rows * 8 + 4 + (size_t)cap_entries * 8;
    h->stage_valid = false;   // (the pinned staging is shared with rmj_drain_format's size call)
    if (need_pin > h->pin_bytes) {
        if (h->h_pin) hipHostFree(h->h_pin);
        h->h_pin = nullptr; h->pin_bytes = 0;
        HIPCHK(hipHostMalloc(&h->h_pin, need_pin, hipHostMallocDefault));
        h->pin_bytes = need_pin;
    }
    uint8_t* pin = (uint8_t*)h->h_pin;
    HIPCHK(hipMemcpyAsync(pin, tot, 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    const uint32_t rows = ((uint32_t*)pin)[0], ents = ((uint32_t*)pin)[1];
    *n_rows = rows; *n_entries = ents;
    const uint32_t wr = rows < cap_rows ? rows : cap_rows, we = ents < cap_entries ? ents : cap_entries;
    uint8_t *p_idx = pin + 16, *p_off = p_idx + (size_t)cap_rows * 4, *p_ent = p_off + ((size_t)cap_rows + 1) * 4;
    if (wr) {
        HIPCHK(hipMemcpyAsync(p_idx, d_idx, (size_t)wr * 4, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipMemcpyAsync(p_off, d_off, ((size_t)wr + 1) * 4, hipMemcpyDeviceToHost, h->stream));
    }
    if (we) HIPCHK(hipMemcpyAsync(p_ent, d_ent, (size_t)we * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    memcpy(index, p_idx, (size_t)wr * 4);
    if (wr) memcpy(offsets, p_off, ((size_t)wr + 1) * 4);   // offsets[wr] = the end of the last row written
    else offsets[0] = 0u;
    memcpy(entries, p_ent, (size_t)we * 8);
    return RMJ_OK;
}
int rmj_get_mask(rmj_handle h, uint8_t* mask) {
    if (!h || !mask) return fail(RMJ_ERR_ARG, "null argument");
    SYNC_FETCH(mask, h->d.mask, (size_t)h->cfg.n_games * 4 * 82);
    return RMJ_OK;
}
int rmj_get_waits(rmj_handle h, uint64_t* waits) {
    if (!h || !waits) return fail(RMJ_ERR_ARG, "null argument");
    SYNC_FETCH(waits, h->d.waits, (size_t)h->cfg.n_games * 4 * sizeof(uint64_t));
    return RMJ_OK;
}
static int fetch_scores(rmj_handle h, std::vector<int32_t>& sc, std::vector<uint32_t>& evc) {
    uint32_t n = h->cfg.n_games;
    int32_t* d_sc;
    uint32_t* d_ev;
    HIPCHK(hipSetDevice(h->cfg.device));
    void* sp;
    int rcs = scratch_for(h, (size_t)n * 20, &sp);
    if (rcs) return rcs;
    d_sc = (int32_t*)sp;
    d_ev = (uint32_t*)((char*)sp + (size_t)n * 16);
    hipLaunchKernelGGL(k_gather_scores, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->d.core, n, d_sc, d_ev);
    sc.resize((size_t)n * 4);
    evc.resize(n);
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(sc.data(), d_sc, (size_t)n * 16, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(evc.data(), d_ev, (size_t)n * 4, hipMemcpyDeviceToHost));
    return RMJ_OK;
}
int rmj_get_scores(rmj_handle h, int32_t* scores) {
    if (!h || !scores) return fail(RMJ_ERR_ARG, "null argument");
    std::vector<int32_t> sc;
    std::vector<uint32_t> ev;
    int rc = fetch_scores(h, sc, ev);
    if (rc) return rc;
    memcpy(scores, sc.data(), sc.size() * 4);
    return RMJ_OK;
}
// state.wall.salt / state.wall.wall_digest of the reference (state/wall.rs:15-16): 17 / 65 bytes per game, NUL-terminated; both empty for a
// wall without them (no RMJ_RULE_REFERENCE_RNG, or after a start_kyoku event: event_handler.rs:81-82)
int rmj_get_wall_digests(rmj_handle h, uint32_t first, uint32_t n, char* salts /*[n][17]*/, char* digests /*[n][65]*/) {
    if (!h || !salts || !digests) return fail(RMJ_ERR_ARG, "null argument");
    if (first > h->cfg.n_games || n > h->cfg.n_games - first) return fail(RMJ_ERR_RANGE, "game range out of bounds");
    if (n == 0) return RMJ_OK;
    HIPCHK(hipSetDevice(h->cfg.device));
    void* sp;
    int rcs = scratch_for(h, (size_t)n * 44, &sp);
    if (rcs) return rcs;
    hipLaunchKernelGGL(k_wall_digest, dim3((n + 63) / 64), dim3(64), 0, h->stream, h->d.core, h->d.wall, h->d.wall_dg, first, n,
                       h->cfg.game_mode >= 3 ? 108 : 136, (uint32_t*)sp);
    HIPCHK(hipGetLastError());
    std::vector<uint32_t> host((size_t)n * 11);
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(host.data(), sp, (size_t)n * 44, hipMemcpyDeviceToHost));
    static const char* HX = "0123456789abcdef";
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t* o = host.data() + (size_t)i * 11;
        char* s = salts + (size_t)i * 17;
        char* d = digests + (size_t)i * 65;
        if (!o[0]) { s[0] = 0; d[0] = 0; continue; }
        const uint64_t salt = (uint64_t)o[1] | ((uint64_t)o[2] << 32);
        for (int k = 0; k < 16; k++) s[k] = HX[(salt >> (60 - 4 * k)) & 15];   // format!("{:016x}")
        s[16] = 0;
        for (int k = 0; k < 64; k++) d[k] = HX[(o[3 + (k >> 3)] >> (28 - 4 * (k & 7))) & 15];   // format!("{:x}", hasher.finalize())
        d[64] = 0;
    }
    return RMJ_OK;
}
int rmj_get_wall_digest(rmj_handle h, uint32_t game, char* salt /*[17]*/, char* digest /*[65]*/) {
    if (!h) return fail(RMJ_ERR_ARG, "null argument");
    if (game >= h->cfg.n_games) return fail(RMJ_ERR_RANGE, "game index out of range");
    return rmj_get_wall_digests(h, game, 1, salt, digest);
}
int rmj_get_ranks(rmj_handle h, uint8_t* ranks) {  // env.rs:673-689
    if (!h || !ranks) return fail(RMJ_ERR_ARG, "null argument");
    std::vector<int32_t> sc;
    std::vector<uint32_t> ev;
    int rc = fetch_scores(h, sc, ev);
    if (rc) return rc;
    const int np = h->cfg.game_mode >= 3 ? 3 : 4;
    for (uint32_t g = 0; g < h->cfg.n_games; g++)
        for (int a = 0; a < 4; a++) {
            int r = 1;
            for (int b = 0; b < np; b++)
                if (sc[g * 4 + b] > sc[g * 4 + a] || (sc[g * 4 + b] == sc[g * 4 + a] && b < a)) r++;
            ranks[g * 4 + a] = a < np ? (uint8_t)r : 0;
        }
    return RMJ_OK;
}
int rmj_get_event_counts(rmj_handle h, uint32_t* counts) {
    if (!h || !counts) return fail(RMJ_ERR_ARG, "null argument");
    std::vector<int32_t> sc;
    std::vector<uint32_t> ev;
    int rc = fetch_scores(h, sc, ev);
    if (rc) return rc;
    memcpy(counts, ev.data(), ev.size() * 4);
    return RMJ_OK;
}
int rmj_get_step_counts(rmj_handle h, uint64_t* steps) {
    if (!h || !steps) return fail(RMJ_ERR_ARG, "null argument");
    uint32_t n = h->cfg.n_games;
    uint64_t* d;
    HIPCHK(hipSetDevice(h->cfg.device));
    void* sp;
    int rcs = scratch_for(h, (size_t)n * 8, &sp);
    if (rcs) return rcs;
    d = (uint64_t*)sp;
    hipLaunchKernelGGL(k_gather_steps, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->d.core, n, d);
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(steps, d, (size_t)n * 8, hipMemcpyDeviceToHost));
    return RMJ_OK;
}
int rmj_total_steps(rmj_handle h, uint64_t* total) {
    if (!h || !total) return fail(RMJ_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(hipMemsetAsync(h->d_counter, 0, 8, h->stream));
    hipLaunchKernelGGL(k_sum_steps, dim3(512), dim3(256), 0, h->stream, h->d.core, h->cfg.n_games, h->d_counter);
    unsigned long long v = 0;
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(&v, h->d_counter, 8, hipMemcpyDeviceToHost));
    *total = v;
    return RMJ_OK;
}

int rmj_get_win_results(rmj_handle h, uint32_t game, RmjWinResult* out, uint8_t* seat_mask) {
    if (!h || !out || !seat_mask) return fail(RMJ_ERR_ARG, "null argument");
    if (game >= h->cfg.n_games) return fail(RMJ_ERR_RANGE, "game index out of range");
    GState st;
    SYNC_FETCH(&st, h->d.core + game, sizeof(GState));
    HIPCHK(hipMemcpy(out, h->d.win + (size_t)game * 4, 4 * sizeof(RmjWinResult), hipMemcpyDeviceToHost));
    *seat_mask = st.win_mask;
    for (int p = 0; p < 4; p++)
        if (!((st.win_mask >> p) & 1u)) memset(&out[p], 0, sizeof(RmjWinResult));
    return RMJ_OK;
}
int rmj_get_events(rmj_handle h, uint32_t game, uint32_t first, uint32_t max_events, RmjEvent* out, uint32_t* n_out) {
    if (!h || !out || !n_out) return fail(RMJ_ERR_ARG, "null argument");
    if (game >= h->cfg.n_games) return fail(RMJ_ERR_RANGE, "game index out of range");
    GState st;
    SYNC_FETCH(&st, h->d.core + game, sizeof(GState));
    // `first` counts from the current game's first record (GameState.mjai_log: a reset starts it again); the ring runs on stream positions
    const uint32_t total = st.ev_count - st.ev_base;
    const uint32_t lo = total > h->ring ? total - h->ring : 0;
    if (first < lo) return fail(RMJ_ERR_RANGE, "requested events already overwritten in the ring (create with a larger event_ring)");
    uint32_t n = 0;
    std::vector<RmjEvent> ring(h->ring);
    HIPCHK(hipMemcpy(ring.data(), h->d.events + (size_t)game * h->ring, (size_t)h->ring * sizeof(RmjEvent), hipMemcpyDeviceToHost));
    for (uint32_t i = first; i < total && n < max_events; i++) out[n++] = ring[(st.ev_base + i) & (h->ring - 1)];
    *n_out = n;
    return RMJ_OK;
}

// ---- state peek / poke (conversions: rmj_host.h) -------------------------------------------
int rmj_peek_state(rmj_handle h, uint32_t game, RmjStateView* out) {
    if (!h || !out) return fail(RMJ_ERR_ARG, "null argument");
    if (game >= h->cfg.n_games) return fail(RMJ_ERR_RANGE, "game index out of range");
    GState st;
    uint8_t W[RMJ_WALL_STRIDE];
    SYNC_FETCH(&st, h->d.core + game, sizeof(GState));
    HIPCHK(hipMemcpy(W, h->d.wall + (size_t)game * RMJ_WALL_STRIDE, RMJ_WALL_STRIDE, hipMemcpyDeviceToHost));
    rmjh::to_view(st, W, out);
    return RMJ_OK;
}

int rmj_poke_state(rmj_handle h, uint32_t game, const RmjStateView* v) {
    if (!h || !v) return fail(RMJ_ERR_ARG, "null argument");
    if (game >= h->cfg.n_games) return fail(RMJ_ERR_RANGE, "game index out of range");
    GState S;
    uint8_t W[RMJ_WALL_STRIDE];
    SYNC_FETCH(&S, h->d.core + game, sizeof(GState));
    HIPCHK(hipMemcpy(W, h->d.wall + (size_t)game * RMJ_WALL_STRIDE, RMJ_WALL_STRIDE, hipMemcpyDeviceToHost));
    if (S.wall_meta == 1) {   // the view may carry another wall: salt and digest stay those of the shuffled one (state/wall.rs:69-80)
        void* sp;
        int rcs = scratch_for(h, 44, &sp);
        if (rcs) return rcs;
        hipLaunchKernelGGL(k_wall_digest, dim3(1), dim3(64), 0, h->stream, h->d.core, h->d.wall, h->d.wall_dg, game, 1u,
                           h->cfg.game_mode >= 3 ? 108 : 136, (uint32_t*)sp);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(h->d.wall_dg + (size_t)game * 8, (const uint32_t*)sp + 3, 32, hipMemcpyDeviceToDevice, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        S.wall_meta = 2;
    }
    if (const char* err = rmjh::from_view(S, W, v)) return fail(RMJ_ERR_ARG, err);
    HIPCHK(hipMemcpy(h->d.core + game, &S, sizeof(GState), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->d.wall + (size_t)game * RMJ_WALL_STRIDE, W, RMJ_WALL_STRIDE, hipMemcpyHostToDevice));
    if (h->cfg.game_mode >= 3) hipLaunchKernelGGL(rmj3::k_refresh, dim3(1), dim3(64), 0, h->stream, (const Env*)h->d_env, game);
    else hipLaunchKernelGGL(rmj4::k_refresh, dim3(1), dim3(64), 0, h->stream, (const Env*)h->d_env, game);
    if (h->d_track) hipLaunchKernelGGL(k_track_mark, dim3(1), dim3(64), 0, h->stream, (uint8_t*)h->d_track + (size_t)h->cfg.n_games * 37, (const uint8_t*)nullptr, game, 1u);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    return RMJ_OK;
}

// ---- MJAI formatting (state/mod.rs:2094-2148; parser.rs:301-334; formatter: rmj_host.h) -----
int rmj_format_event(const RmjEvent* ev, uint32_t n_avail, int seat, char* buf, uint32_t cap) {
    if (!ev || !buf || n_avail == 0 || cap == 0) return RMJ_ERR_ARG;
    rmjh::Out o{buf, buf + (cap - 1), 0};
    const int used = rmjh::format_event(o, ev, n_avail, seat);
    if (used < 0) return used;
    if (o.need + 1 > cap) return RMJ_ERR_RANGE;
    *o.p = 0;
    return used;
}
int rmj_format_events(const RmjEvent* ev, const uint32_t* offsets, uint32_t n_games, int seat, char* buf, uint64_t cap, uint64_t* text_offsets,
                      uint64_t* needed) {
    if (!ev || !offsets || !text_offsets || !needed) return RMJ_ERR_ARG;
    *needed = rmjh::format_events(ev, offsets, n_games, seat, buf, cap, text_offsets, 0);
    return (buf && *needed <= cap) ? RMJ_OK : RMJ_ERR_RANGE;
}

// ---- per-round rewards for a trainer on the same GPU -------------------------------------------
// What riichienv-ml's PPO worker derives on the host between steps (trainers/_ppo_worker.py:100-116, 240-266, 283-291): when a
// round has ended, the seats' score deltas over that round and the round's opening facts (the GRP features chang / ju / ben /
// liqibang); when the game has ended, its final scores (rank rewards).  A tracker per handle remembers where every game's current
// round began; one small launch after a step compares: the wall's hand index moves with every deal (state/wall.rs:36-40), is_done
// with the end of the game.  ended: 0 = the round goes on, 1 = a round ended and the next one was dealt, 2 = the round and the game
// ended; a finished game that was restarted (auto-reset, rmj_reset) re-opens silently.
struct RoundTrack {            // device arrays of the tracker (rmj_env::d_track)
    uint32_t* hand_index;      // [n] hand index when the game's current round was dealt
    uint8_t* was_done;         // [n]
    int32_t* start_scores;     // [n][4]
    int32_t* start_meta;       // [n][4] round_wind, oya, honba, riichi_sticks at the deal
    uint8_t* mark;             // [n] set by rmj_reset / rmj_poke_state for the games they touch: the tracker takes the new state as its baseline
};

__global__ void k_round_track(const GState* __restrict__ core, uint32_t n, RoundTrack T, int baseline, uint8_t* __restrict__ ended, int32_t* __restrict__ delta,
                              int32_t* __restrict__ meta, uint8_t* __restrict__ kyoku_idx) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    const GState& S = core[g];
    const uint32_t hi = S.hand_index;
    const bool done = S.is_done != 0;
    int32_t sc[4];
    for (int p = 0; p < 4; p++) sc[p] = S.p[p].score;
    uint8_t e = 0;
    bool rebase = baseline != 0;
    const bool marked = T.mark[g] != 0;   // rmj_reset / rmj_poke_state touched the game since the last call: no round of THIS game ended
    if (marked) { T.mark[g] = 0; rebase = true; }
    if (!baseline && !marked) {
        const bool wd = T.was_done[g] != 0;
        if (wd && !done) rebase = true;                                  // restarted: a new game opens
        else if (done && !wd) e = 2;                                     // the round that ended the game
        else if (!done && hi != T.hand_index[g]) { e = 1; rebase = true; }
    }
    if (ended) ended[g] = e;
    if (kyoku_idx) kyoku_idx[g] = S.kyoku_idx;
    for (int p = 0; p < 4; p++) {
        if (delta) delta[(size_t)g * 4 + p] = e ? sc[p] - T.start_scores[(size_t)g * 4 + p] : 0;
        if (meta) meta[(size_t)g * 4 + p] = e ? T.start_meta[(size_t)g * 4 + p] : 0;
    }
    if (rebase) {
        T.hand_index[g] = hi;
        for (int p = 0; p < 4; p++) T.start_scores[(size_t)g * 4 + p] = sc[p];
        T.start_meta[(size_t)g * 4 + 0] = S.round_wind; T.start_meta[(size_t)g * 4 + 1] = S.oya;
        T.start_meta[(size_t)g * 4 + 2] = S.honba; T.start_meta[(size_t)g * 4 + 3] = (int32_t)S.riichi_sticks;
    }
    T.was_done[g] = done ? 1 : 0;
}
static int round_track_impl(rmj_env* h, int baseline, uint8_t* d_ended, int32_t* d_delta, int32_t* d_meta, uint8_t* d_kyoku_idx);

// ---- bulk drain of the event rings ----------------------------------------------------------
// RiichiEnv.mjai_log / per-seat logs of EVERY game (riichienv-python/src/env.rs:729-739, state/mod.rs:2094-2148): the records each
// game slot wrote since the caller's cursor, gathered on the device into one dense buffer (two-level scan of the counts, one wave per
// game copies its window of the ring) and brought down with one copy.  Cursors are positions in the slot's record stream
// (GState::ev_count never goes back: a restart moves ev_base), so a window may hold the end of one game and the start of the next.
// A slot whose ring was lapped since its cursor lost its oldest records: the window starts at the oldest record still there; the loss is
// booked per slot (RmjEventViews.lost, cumulative) by the drain that hands the window over, not by peeks or failed calls.
__global__ __launch_bounds__(LC_BLOCK) void k_ev_count(const GState* __restrict__ core, uint32_t n, uint32_t ring, const uint32_t* __restrict__ cursor,
                                                       uint32_t* __restrict__ first, uint32_t* __restrict__ pre, uint32_t* __restrict__ blk) {
    __shared__ uint32_t sc[LC_BLOCK];
    const uint32_t g = blockIdx.x * LC_BLOCK + threadIdx.x;
    uint32_t c = 0;
    if (g < n) {
        const uint32_t total = core[g].ev_count;
        uint32_t behind = total - cursor[g];           // wrap-safe distance; a cursor "ahead" of the stream (not this slot's) reads as nothing new
        if (behind > 0x80000000u) behind = 0u;
        const uint32_t take = behind > ring ? ring : behind;
        first[g] = total - take;
        c = take;
    }
    sc[threadIdx.x] = c;
    __syncthreads();
    for (int off = 1; off < LC_BLOCK; off <<= 1) {
        uint32_t a = 0;
        if ((int)threadIdx.x >= off) a = sc[threadIdx.x - off];
        __syncthreads();
        sc[threadIdx.x] += a;
        __syncthreads();
    }
    if (g < n) pre[g] = sc[threadIdx.x] - c;
    if (threadIdx.x == LC_BLOCK - 1) blk[blockIdx.x] = sc[threadIdx.x];
}
__global__ void k_ev_scan(uint32_t* blk, uint32_t blocks, uint32_t* total) {
    if (blockIdx.x || threadIdx.x) return;
    uint32_t r = 0;
    for (uint32_t b = 0; b < blocks; b++) { const uint32_t c = blk[b]; blk[b] = r; r += c; }
    total[0] = r;
}
// one wave per game: lane = (record, half) - 16 bytes per lane, 32 records per pass.  newcur[g] = the position behind the window
// (the window = [first[g], ev_count): the size call of rmj_drain_format keeps what it gathered staged, so no second gather needs a stop position)
__global__ __launch_bounds__(256) void k_ev_gather(const GState* __restrict__ core, const RmjEvent* __restrict__ events, uint32_t n, uint32_t ring,
                                                   const uint32_t* __restrict__ first, const uint32_t* __restrict__ pre, const uint32_t* __restrict__ blk,
                                                   uint32_t cap, RmjEvent* __restrict__ out, uint32_t* __restrict__ offs, uint32_t* __restrict__ newcur) {
    const uint32_t g = blockIdx.x * 4u + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (g >= n) return;
    const uint32_t total = core[g].ev_count, lo = first[g], base = blk[g / LC_BLOCK] + pre[g], cnt = total - lo;
    const uint4* src = reinterpret_cast<const uint4*>(events + (size_t)g * ring);
    uint4* dst = reinterpret_cast<uint4*>(out);
    for (uint32_t k = (uint32_t)(lane >> 1); k < cnt; k += 32u) {
        const uint32_t o = base + k;
        if (o < cap) dst[2 * (size_t)o + (lane & 1)] = src[2 * (size_t)((lo + k) & (ring - 1u)) + (lane & 1)];
    }
    if (lane == 0) {
        offs[g] = base;
        if (g == n - 1u) offs[n] = base + cnt;
        newcur[g] = total;
    }
}
// the drain is handed over: what its windows skipped is lost
__global__ void k_ev_book(const uint32_t* __restrict__ cursor, const uint32_t* __restrict__ first, uint32_t n, uint32_t* __restrict__ lost) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    const uint32_t skipped = first[g] - cursor[g];
    if (skipped && skipped <= 0x80000000u) lost[g] += skipped;
}
// the device part of a drain: the handle's scratch holds [cursor | first | pre | blk | total | offsets | new cursor | records]
struct DrainPlan { uint32_t *d_cur, *d_first, *d_pre, *d_blk, *d_tot, *d_off, *d_new; RmjEvent* d_ev; uint32_t cap; };
static int drain_device(rmj_env* h, const uint32_t* cursor, uint32_t cap_events, DrainPlan* P, uint32_t* n_events) {
    const uint32_t n = h->cfg.n_games, blocks = (n + LC_BLOCK - 1) / LC_BLOCK;
    if (!h->d_ev_lost) {
        HIPCHK(hipMalloc(&h->d_ev_lost, (size_t)n * 4));
        HIPCHK(hipMemsetAsync(h->d_ev_lost, 0, (size_t)n * 4, h->stream));
    }
    const size_t o_first = (size_t)n * 4, o_pre = o_first + (size_t)n * 4, o_blk = o_pre + (size_t)n * 4, o_tot = o_blk + (size_t)blocks * 4;
    const size_t o_off = o_tot + 16, o_new = o_off + ((size_t)n + 1) * 4, o_ev = (o_new + (size_t)n * 4 + 31) & ~(size_t)31;
    // the records: a first pass sizes them (the scan total), the buffer is sized by the caller's cap or, when it passes 0, by the total
    void* sp;
    int rc = scratch_for(h, o_ev + (size_t)cap_events * sizeof(RmjEvent), &sp);
    if (rc) return rc;
    uint8_t* base = (uint8_t*)sp;
    P->d_cur = (uint32_t*)base; P->d_first = (uint32_t*)(base + o_first); P->d_pre = (uint32_t*)(base + o_pre); P->d_blk = (uint32_t*)(base + o_blk);
    P->d_tot = (uint32_t*)(base + o_tot); P->d_off = (uint32_t*)(base + o_off); P->d_new = (uint32_t*)(base + o_new); P->d_ev = (RmjEvent*)(base + o_ev);
    P->cap = cap_events;
    HIPCHK(hipMemcpyAsync(P->d_cur, cursor, (size_t)n * 4, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_ev_count, dim3(blocks), dim3(LC_BLOCK), 0, h->stream, (const GState*)h->d.core, n, h->ring, (const uint32_t*)P->d_cur, P->d_first, P->d_pre, P->d_blk);
    hipLaunchKernelGGL(k_ev_scan, dim3(1), dim3(64), 0, h->stream, P->d_blk, blocks, P->d_tot);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(n_events, P->d_tot, 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return RMJ_OK;
}
static void drain_gather(rmj_env* h, const DrainPlan& P) {
    const uint32_t n = h->cfg.n_games;
    hipLaunchKernelGGL(k_ev_gather, dim3((n + 3u) / 4u), dim3(256), 0, h->stream, (const GState*)h->d.core, (const RmjEvent*)h->d.events, n, h->ring,
                       (const uint32_t*)P.d_first, (const uint32_t*)P.d_pre, (const uint32_t*)P.d_blk, P.cap, P.d_ev, P.d_off, P.d_new);
}
static void drain_book(rmj_env* h, const DrainPlan& P) {
    const uint32_t n = h->cfg.n_games;
    hipLaunchKernelGGL(k_ev_book, dim3((n + 255u) / 256u), dim3(256), 0, h->stream, (const uint32_t*)P.d_cur, (const uint32_t*)P.d_first, n, h->d_ev_lost);
}
static int pin_for(rmj_env* h, size_t bytes) {
    if (bytes > h->pin_bytes) {
        if (h->h_pin) hipHostFree(h->h_pin);
        h->h_pin = nullptr; h->pin_bytes = 0;
        h->stage_valid = false;
        HIPCHK(hipHostMalloc(&h->h_pin, bytes, hipHostMallocDefault));
        h->pin_bytes = bytes;
    }
    return RMJ_OK;
}
int rmj_get_log_positions(rmj_handle h, uint32_t* base, uint32_t* pos) {
    if (!h) return fail(RMJ_ERR_ARG, "null handle");
    HIPCHK(hipSetDevice(h->cfg.device));
    const uint32_t n = h->cfg.n_games;
    void* sp;
    int rc = scratch_for(h, (size_t)n * 8, &sp);
    if (rc) return rc;
    uint32_t* d = (uint32_t*)sp;
    hipLaunchKernelGGL(k_log_positions, dim3((n + 255u) / 256u), dim3(256), 0, h->stream, (const GState*)h->d.core, n, d, d + n);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    if (base) HIPCHK(hipMemcpy(base, d, (size_t)n * 4, hipMemcpyDeviceToHost));
    if (pos) HIPCHK(hipMemcpy(pos, d + n, (size_t)n * 4, hipMemcpyDeviceToHost));
    return RMJ_OK;
}
int rmj_drain_events(rmj_handle h, uint32_t* cursor, RmjEvent* out, uint32_t cap_events, uint32_t* offsets, uint32_t* n_events, uint32_t flags) {
    if (!h || !cursor || !offsets || !n_events || (!out && cap_events)) return fail(RMJ_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(h->cfg.device));
    const uint32_t n = h->cfg.n_games;
    h->stage_valid = false;
    DrainPlan P;
    int rc = drain_device(h, cursor, cap_events, &P, n_events);
    if (rc) return rc;
    if (*n_events > cap_events) return fail(RMJ_ERR_RANGE, "rmj_drain_events: more records than cap_events (n_events holds the number; nothing was drained)");
    drain_gather(h, P);
    if (!(flags & RMJ_DRAIN_PEEK)) drain_book(h, P);
    HIPCHK(hipGetLastError());
    rc = pin_for(h, ((size_t)n * 2 + 1) * 4 + (size_t)*n_events * sizeof(RmjEvent));
    if (rc) return rc;
    uint8_t* pin = (uint8_t*)h->h_pin;
    HIPCHK(hipMemcpyAsync(pin, P.d_off, ((size_t)n + 1) * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(pin + ((size_t)n + 1) * 4, P.d_new, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream));
    if (*n_events) HIPCHK(hipMemcpyAsync(out, P.d_ev, (size_t)*n_events * sizeof(RmjEvent), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    memcpy(offsets, pin, ((size_t)n + 1) * 4);
    if (!(flags & RMJ_DRAIN_PEEK)) memcpy(cursor, pin + ((size_t)n + 1) * 4, (size_t)n * 4);
    return RMJ_OK;
}
// drain + format in one call: the records go to pinned staging owned by the handle and are formatted from there by a pool of host
// threads (one log per slot, events separated by '\n').  ms (optional, [3]): device gather, copy to the host, formatting.
// A size call (buf = NULL) leaves its gathered records staged; the call that follows with the same cursors, seat and flags formats that
// staging instead of draining again (the drain is then "as of the size call": what was logged since stays for the next drain).
int rmj_drain_format(rmj_handle h, uint32_t* cursor, int seat, char* buf, uint64_t cap, uint64_t* text_offsets, uint64_t* needed, uint32_t* n_events,
                     double* ms, uint32_t flags) {
    if (!h || !cursor || !text_offsets || !needed || !n_events) return fail(RMJ_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(h->cfg.device));
    const uint32_t n = h->cfg.n_games;
    const size_t o_ev = (((size_t)n * 3 + 1) * 4 + 31) & ~(size_t)31;   // pinned: [offsets n + 1 | new cursors n | first n | pad | records]
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto t0 = now();
    const bool reuse = buf && h->stage_valid && h->stage_seat == seat && h->stage_cursor.size() == n &&
                       memcmp(h->stage_cursor.data(), cursor, (size_t)n * 4) == 0;
    if (!reuse) {
        h->stage_valid = false;
        DrainPlan P;
        int rc = drain_device(h, cursor, 0, &P, n_events);   // size pass with no record buffer ...
        if (rc) return rc;
        rc = drain_device(h, cursor, *n_events, &P, n_events);   // ... then the gather into a buffer of exactly that size (nothing ran in between)
        if (rc) return rc;
        drain_gather(h, P);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(h->stream));
        auto t1 = now();
        rc = pin_for(h, o_ev + (size_t)*n_events * sizeof(RmjEvent));
        if (rc) return rc;
        uint8_t* pin = (uint8_t*)h->h_pin;
        HIPCHK(hipMemcpyAsync(pin, P.d_off, ((size_t)n + 1) * 4, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipMemcpyAsync(pin + ((size_t)n + 1) * 4, P.d_new, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipMemcpyAsync(pin + ((size_t)n * 2 + 1) * 4, P.d_first, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream));
        if (*n_events) HIPCHK(hipMemcpyAsync(pin + o_ev, P.d_ev, (size_t)*n_events * sizeof(RmjEvent), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        auto t2 = now();
        h->stage_ms[0] = std::chrono::duration<double, std::milli>(t1 - t0).count();
        h->stage_ms[1] = std::chrono::duration<double, std::milli>(t2 - t1).count();
        h->stage_cursor.assign(cursor, cursor + n);
        h->stage_seat = seat;
        h->stage_events = *n_events;
        h->stage_valid = true;
    }
    uint8_t* pin = (uint8_t*)h->h_pin;
    *n_events = h->stage_events;
    auto t2 = now();
    *needed = rmjh::format_events((const RmjEvent*)(pin + o_ev), (const uint32_t*)pin, n, seat, buf, cap, text_offsets, 0);
    auto t3 = now();
    if (ms) {
        ms[0] = h->stage_ms[0];
        ms[1] = h->stage_ms[1];
        ms[2] = std::chrono::duration<double, std::milli>(t3 - t2).count();
    }
    if (!buf || *needed > cap)   // nothing was handed over: cursors and loss counters stand, the staging waits for the call with a buffer
        return fail(RMJ_ERR_RANGE, "rmj_drain_format: text buffer too small (needed holds the size; cursors unchanged)");
    h->stage_valid = false;
    if (!(flags & RMJ_DRAIN_PEEK)) {
        // book what the windows skipped (first - cursor) and move the cursors behind the windows
        const uint32_t* firsts = (const uint32_t*)(pin + ((size_t)n * 2 + 1) * 4);
        bool any = false;
        for (uint32_t g = 0; g < n && !any; g++) any = firsts[g] != cursor[g];
        if (any) {
            void* sp;
            int rc = scratch_for(h, (size_t)n * 8, &sp);
            if (rc) return rc;
            uint32_t* d = (uint32_t*)sp;
            HIPCHK(hipMemcpyAsync(d, cursor, (size_t)n * 4, hipMemcpyHostToDevice, h->stream));
            HIPCHK(hipMemcpyAsync(d + n, firsts, (size_t)n * 4, hipMemcpyHostToDevice, h->stream));
            hipLaunchKernelGGL(k_ev_book, dim3((n + 255u) / 256u), dim3(256), 0, h->stream, (const uint32_t*)d, (const uint32_t*)(d + n), n, h->d_ev_lost);
            HIPCHK(hipGetLastError());
            HIPCHK(hipStreamSynchronize(h->stream));
        }
        memcpy(cursor, pin + ((size_t)n + 1) * 4, (size_t)n * 4);
    }
    return RMJ_OK;
}
static int round_track_impl(rmj_env* h, int baseline, uint8_t* d_ended, int32_t* d_delta, int32_t* d_meta, uint8_t* d_kyoku_idx) {
    const uint32_t n = h->cfg.n_games;
    if (!h->d_track) {
        HIPCHK(hipMalloc(&h->d_track, (size_t)n * (4 + 16 + 16 + 4)));
        HIPCHK(hipMemsetAsync(h->d_track, 0, (size_t)n * (4 + 16 + 16 + 4), h->stream));
        baseline = 1;
    }
    RoundTrack T;
    uint8_t* b = (uint8_t*)h->d_track;
    T.hand_index = (uint32_t*)b; T.start_scores = (int32_t*)(b + (size_t)n * 4); T.start_meta = (int32_t*)(b + (size_t)n * 20); T.was_done = b + (size_t)n * 36; T.mark = b + (size_t)n * 37;
    hipLaunchKernelGGL(k_round_track, dim3((n + 255u) / 256u), dim3(256), 0, h->stream, (const GState*)h->d.core, n, T, baseline, d_ended, d_delta, d_meta, d_kyoku_idx);
    HIPCHK(hipGetLastError());
    return RMJ_OK;
}
int rmj_round_track_device(rmj_handle h, uint8_t* d_ended, int32_t* d_delta, int32_t* d_meta, uint8_t* d_kyoku_idx) {
    if (!h) return fail(RMJ_ERR_ARG, "null handle");
    HIPCHK(hipSetDevice(h->cfg.device));
    return round_track_impl(h, 0, d_ended, d_delta, d_meta, d_kyoku_idx);
}
int rmj_round_track_reset(rmj_handle h) {
    if (!h) return fail(RMJ_ERR_ARG, "null handle");
    HIPCHK(hipSetDevice(h->cfg.device));
    return round_track_impl(h, 1, nullptr, nullptr, nullptr, nullptr);
}
int rmj_get_events_lost(rmj_handle h, uint32_t* lost) {
    if (!h || !lost) return fail(RMJ_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(h->cfg.device));
    if (!h->d_ev_lost) { memset(lost, 0, (size_t)h->cfg.n_games * 4); return RMJ_OK; }
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(lost, h->d_ev_lost, (size_t)h->cfg.n_games * 4, hipMemcpyDeviceToHost));
    return RMJ_OK;
}
int rmj_event_views(rmj_handle h, RmjEventViews* out) {
    if (!h || !out) return fail(RMJ_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(h->cfg.device));
    if (!h->d_ev_lost) {
        HIPCHK(hipMalloc(&h->d_ev_lost, (size_t)h->cfg.n_games * 4));
        HIPCHK(hipMemsetAsync(h->d_ev_lost, 0, (size_t)h->cfg.n_games * 4, h->stream));
    }
    out->n_games = h->cfg.n_games;
    out->ring = h->ring;
    out->events = h->d.events;
    out->ev_count = &h->d.core[0].ev_count;
    out->ev_count_stride = (uint32_t)sizeof(GState);
    out->lost = h->d_ev_lost;
    out->ev_base = &h->d.core[0].ev_base;
    return RMJ_OK;
}

// ---- batched hand math ---------------------------------------------------------------------
int rmj_eval_hands(int device, const RmjHandCase* cases, uint32_t n, RmjHandResult* out) {
    DevTmp tmp;
    if (!cases || !out) return fail(RMJ_ERR_ARG, "null argument");
    int rc = ensure_device(device);
    if (rc) return rc;
    if (n == 0) return RMJ_OK;
    RmjHandCase* d_in;
    RmjHandResult* d_out;
    HIPCHK(tmp.alloc(&d_in, (size_t)n * sizeof(RmjHandCase)));
    HIPCHK(tmp.alloc(&d_out, (size_t)n * sizeof(RmjHandResult)));
    HIPCHK(hipMemcpy(d_in, cases, (size_t)n * sizeof(RmjHandCase), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_eval_hands, dim3((n + 15u) / 16u), dim3(256), 0, 0, d_in, n, d_out);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out, d_out, (size_t)n * sizeof(RmjHandResult), hipMemcpyDeviceToHost));
    return RMJ_OK;
}
int rmj_agari_counts(int device, const uint8_t* counts, uint32_t n, uint8_t* is_agari_out, uint8_t* is_tenpai, uint64_t* waits) {
    DevTmp tmp;
    if (!counts || !is_agari_out || !is_tenpai || !waits) return fail(RMJ_ERR_ARG, "null argument");
    int rc = ensure_device(device);
    if (rc) return rc;
    if (n == 0) return RMJ_OK;
    uint8_t *d_c, *d_a, *d_t;
    uint64_t* d_w;
    HIPCHK(tmp.alloc(&d_c, (size_t)n * 34));
    HIPCHK(tmp.alloc(&d_a, n));
    HIPCHK(tmp.alloc(&d_t, n));
    HIPCHK(tmp.alloc(&d_w, (size_t)n * 8));
    HIPCHK(hipMemcpy(d_c, counts, (size_t)n * 34, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_agari_counts, dim3((n + 15) / 16), dim3(256), 0, 0, d_c, n, d_a, d_t, d_w);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(is_agari_out, d_a, n, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(is_tenpai, d_t, n, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(waits, d_w, (size_t)n * 8, hipMemcpyDeviceToHost));
    return RMJ_OK;
}
int rmj_calculate_score(int device, const uint8_t* han, const uint8_t* fu, const uint8_t* is_oya, const uint8_t* is_tsumo,
                        const uint32_t* honba, const uint8_t* num_players, uint32_t n, uint32_t* out) {
    DevTmp tmp;
    if (!han || !fu || !is_oya || !is_tsumo || !honba || !num_players || !out) return fail(RMJ_ERR_ARG, "null argument");
    int rc = ensure_device(device);
    if (rc) return rc;
    if (n == 0) return RMJ_OK;
    uint8_t *d_h, *d_f, *d_o, *d_t, *d_n;
    uint32_t *d_hb, *d_out;
    if ((rc = tmp.upload(han, n, &d_h)) || (rc = tmp.upload(fu, n, &d_f)) || (rc = tmp.upload(is_oya, n, &d_o)) || (rc = tmp.upload(is_tsumo, n, &d_t)) ||
        (rc = tmp.upload(num_players, n, &d_n)) || (rc = tmp.upload(honba, n, &d_hb)))
        return rc;
    HIPCHK(tmp.alloc(&d_out, (size_t)n * 16));
    hipLaunchKernelGGL(k_score, dim3((n + 255) / 256), dim3(256), 0, 0, d_h, d_f, d_o, d_t, d_hb, d_n, n, d_out);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out, d_out, (size_t)n * 16, hipMemcpyDeviceToHost));
    return RMJ_OK;
}

// ---- feature encoder (row A14) --------------------------------------------------------------------
// Observation.encode() of the games [g0, g1) on stream `st`.  The whole 74-channel tensor of a seat is staged at once: windows
// of 37 / 16 channels (more resident waves, the seat's work repeated per window) measured 0.195 / 0.35 ms against 0.19 ms
// for 65 536 4P games - the kernel is bound by its own instruction stream and the store epilogue, not by occupancy.
static void launch_encode_base_range(rmj_env* h, hipStream_t st, int only_active, float* d_out, uint32_t g0, uint32_t g1) {
    const dim3 grid(g1 - g0), block(64);
    if (h->cfg.game_mode >= 3) hipLaunchKernelGGL((k_encode_base<true, false>), grid, block, 0, st, h->d, only_active, d_out, g0, (const uint32_t*)nullptr, (int32_t*)nullptr, 0u, (const uint32_t*)nullptr, (uint32_t*)nullptr);
    else hipLaunchKernelGGL((k_encode_base<false, false>), grid, block, 0, st, h->d, only_active, d_out, g0, (const uint32_t*)nullptr, (int32_t*)nullptr, 0u, (const uint32_t*)nullptr, (uint32_t*)nullptr);
}
static int launch_encode(rmj_handle h, int only_active, float* d_out, bool ext) {
    if (!h || !d_out) return fail(RMJ_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(h->cfg.device));
    const dim3 grid(h->cfg.n_games * 4), block(64);
    const float* decay = h->d_decay;
    const bool sanma = h->cfg.game_mode >= 3;
    if (sanma && ext) hipLaunchKernelGGL((k_encode_ext<true>), grid, block, 0, h->stream, h->d, only_active, decay, d_out);
    else if (ext) hipLaunchKernelGGL((k_encode_ext<false>), grid, block, 0, h->stream, h->d, only_active, decay, d_out);
    else launch_encode_base_range(h, h->stream, only_active, d_out, 0u, h->cfg.n_games);
    HIPCHK(hipGetLastError());
    return RMJ_OK;
}
int rmj_encode_device(rmj_handle h, int only_active, float* d_out) { return launch_encode(h, only_active, d_out, false); }
static void launch_encode_compact(rmj_env* h, float* d_out, int32_t* d_index, uint32_t capacity, uint32_t* d_count) {
    const uint32_t n = h->cfg.n_games, nb = (n + OBS_SCAN_BLOCK - 1) / OBS_SCAN_BLOCK;
    uint32_t* totals = h->d_obs_offs + n;
    hipLaunchKernelGGL(k_obs_offsets, dim3(nb), dim3(OBS_SCAN_BLOCK), 0, h->stream, (const uint32_t*)h->d.status, n, h->d_obs_offs, totals);
    if (h->cfg.game_mode >= 3) hipLaunchKernelGGL((k_encode_base<true, true>), dim3(n), dim3(64), 0, h->stream, h->d, 2, d_out, 0u, (const uint32_t*)h->d_obs_offs, d_index, capacity, (const uint32_t*)totals, d_count);
    else hipLaunchKernelGGL((k_encode_base<false, true>), dim3(n), dim3(64), 0, h->stream, h->d, 2, d_out, 0u, (const uint32_t*)h->d_obs_offs, d_index, capacity, (const uint32_t*)totals, d_count);
}
int rmj_encode_compact_device(rmj_handle h, float* d_out, int32_t* d_index, uint32_t capacity, uint32_t* d_count) {
    if (!h || !d_out || !d_index || !d_count) return fail(RMJ_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(h->cfg.device));
    launch_encode_compact(h, d_out, d_index, capacity, d_count);
    HIPCHK(hipGetLastError());
    return RMJ_OK;
}
int rmj_step_random_encode_compact(rmj_handle h, uint64_t policy_seed, uint32_t n_steps, int auto_reset, float* d_out, int32_t* d_index,
                                   uint32_t capacity, uint32_t* d_count) {
    if (!h || !d_out || !d_index || !d_count) return fail(RMJ_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(h->cfg.device));
    const uint32_t flags = STEP_F_RANDOM | (auto_reset ? STEP_F_AUTORESET : 0u);
    for (uint32_t s = 0; s < n_steps; s++) {
        launch_step_range(h, h->stream, nullptr, policy_seed, flags, 0u, h->cfg.n_games);
        launch_encode_compact(h, d_out, d_index, capacity, d_count);
    }
    HIPCHK(hipGetLastError());
    return RMJ_OK;
}
int rmj_bench_encode_compact(rmj_handle h, float* d_out, int32_t* d_index, uint32_t capacity, uint32_t* d_count, uint32_t reps, double* avg_ms) {
    DevTmp tmp;
    if (!h || !d_out || !d_index || !d_count || !avg_ms || reps == 0) return fail(RMJ_ERR_ARG, "bad argument");
    HIPCHK(hipSetDevice(h->cfg.device));
    hipEvent_t e0, e1;
    HIPCHK(tmp.event(&e0));
    HIPCHK(tmp.event(&e1));
    launch_encode_compact(h, d_out, d_index, capacity, d_count);  // warm-up
    HIPCHK(hipEventRecord(e0, h->stream));
    for (uint32_t i = 0; i < reps; i++) launch_encode_compact(h, d_out, d_index, capacity, d_count);
    HIPCHK(hipEventRecord(e1, h->stream));
    HIPCHK(hipEventSynchronize(e1));
    HIPCHK(hipGetLastError());
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    *avg_ms = (double)ms / reps;
    return RMJ_OK;
}
int rmj_encode(rmj_handle h, int only_active, float* out) {
    if (!h || !out) return fail(RMJ_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(h->cfg.device));
    size_t bytes = (size_t)h->cfg.n_games * 4 * (size_t)h->d.enc_stride * sizeof(float);
    void* sp;
    int rc = scratch_for(h, bytes, &sp);
    if (rc) return rc;
    float* d = (float*)sp;
    if ((rc = rmj_encode_device(h, only_active, d))) return rc;
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(out, d, bytes, hipMemcpyDeviceToHost));
    return RMJ_OK;
}

static size_t aux_floats(const rmj_env* h, int which) {
    const size_t np = h->cfg.game_mode >= 3 ? 3 : 4, w = h->cfg.game_mode >= 3 ? ENC_W3 : ENC_W4;
    return which == 0 ? np * 7 * w : (which == 1 ? np * 21 * 2 : np * 21);
}
int rmj_encode_aux_device(rmj_handle h, int which, float* d_out) {
    if (!h || !d_out) return fail(RMJ_ERR_ARG, "null argument");
    if (which < 0 || which > 2) return fail(RMJ_ERR_ARG, "unknown auxiliary encoder");
    HIPCHK(hipSetDevice(h->cfg.device));
    if (h->cfg.game_mode >= 3) hipLaunchKernelGGL((k_encode_aux<true>), dim3(h->cfg.n_games), dim3(64), 0, h->stream, h->d, which, d_out);
    else hipLaunchKernelGGL((k_encode_aux<false>), dim3(h->cfg.n_games), dim3(64), 0, h->stream, h->d, which, d_out);
    HIPCHK(hipGetLastError());
    return RMJ_OK;
}
int rmj_encode_aux(rmj_handle h, int which, float* out) {
    if (!h || !out) return fail(RMJ_ERR_ARG, "null argument");
    if (which < 0 || which > 2) return fail(RMJ_ERR_ARG, "unknown auxiliary encoder");
    HIPCHK(hipSetDevice(h->cfg.device));
    const size_t bytes = (size_t)h->cfg.n_games * aux_floats(h, which) * sizeof(float);
    void* sp;
    int rc = scratch_for(h, bytes, &sp);
    if (rc) return rc;
    if ((rc = rmj_encode_aux_device(h, which, (float*)sp))) return rc;
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(out, sp, bytes, hipMemcpyDeviceToHost));
    return RMJ_OK;
}

// sequence features (row N3, 4P only like the reference): see rmj_seq.hip.h
int rmj_encode_seq_device(rmj_handle h, int game_style, const RmjSeqBuffers* d) {
    if (!h || !d || !d->sparse || !d->n_sparse || !d->numeric || !d->progression || !d->n_progression || !d->candidates || !d->n_candidates)
        return fail(RMJ_ERR_ARG, "null argument");
    if (h->cfg.game_mode >= 3) return fail(RMJ_ERR_ARG, "sequence features exist for 4-player games only (observation/sequence_features.rs)");
    if (h->cfg.skip_mjai_logging) return fail(RMJ_ERR_ARG, "sequence features read the event log: create the handle with logging on");
    HIPCHK(hipSetDevice(h->cfg.device));
    hipLaunchKernelGGL(k_encode_seq, dim3(h->cfg.n_games), dim3(64), 0, h->stream, h->d, game_style, *d);
    HIPCHK(hipGetLastError());
    return RMJ_OK;
}
int rmj_encode_seq(rmj_handle h, int game_style, const RmjSeqBuffers* out) {
    if (!h || !out) return fail(RMJ_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(h->cfg.device));
    const size_t n = h->cfg.n_games;
    const size_t sz[7] = {n * 4 * RMJ_SEQ_SPARSE * 2, n * 4, n * 4 * 12 * 4, n * RMJ_SEQ_PROG * 5 * 2, n * 2, n * 4 * RMJ_SEQ_CAND * 4 * 2, n * 4};
    size_t off[8] = {0};
    for (int i = 0; i < 7; i++) off[i + 1] = off[i] + ((sz[i] + 255) & ~(size_t)255);
    void* sp;
    int rc = scratch_for(h, off[7], &sp);
    if (rc) return rc;
    char* b = (char*)sp;
    RmjSeqBuffers d{(uint16_t*)(b + off[0]), (uint8_t*)(b + off[1]), (float*)(b + off[2]), (uint16_t*)(b + off[3]), (uint16_t*)(b + off[4]),
                    (uint16_t*)(b + off[5]), (uint8_t*)(b + off[6])};
    if ((rc = rmj_encode_seq_device(h, game_style, &d))) return rc;
    HIPCHK(hipStreamSynchronize(h->stream));
    void* dst[7] = {out->sparse, out->n_sparse, out->numeric, out->progression, out->n_progression, out->candidates, out->n_candidates};
    for (int i = 0; i < 7; i++) {
        if (!dst[i]) return fail(RMJ_ERR_ARG, "null output array");
        HIPCHK(hipMemcpy(dst[i], b + off[i], sz[i], hipMemcpyDeviceToHost));
    }
    return RMJ_OK;
}

int rmj_encode_seq_delta_device(rmj_handle h, int game_style, const RmjSeqDeltaBuffers* d) {
    if (!h || !d || !d->sparse || !d->n_sparse || !d->numeric || !d->progression || !d->n_progression || !d->candidates || !d->n_candidates)
        return fail(RMJ_ERR_ARG, "null argument");
    if (h->cfg.game_mode >= 3) return fail(RMJ_ERR_ARG, "sequence features exist for 4-player games only (observation/sequence_features.rs)");
    if (h->cfg.skip_mjai_logging) return fail(RMJ_ERR_ARG, "sequence features read the event log: create the handle with logging on");
    HIPCHK(hipSetDevice(h->cfg.device));
    hipLaunchKernelGGL(k_encode_seq_delta, dim3(h->cfg.n_games), dim3(64), 0, h->stream, h->d, game_style, *d);
    HIPCHK(hipGetLastError());
    return RMJ_OK;
}
int rmj_encode_seq_delta(rmj_handle h, int game_style, const RmjSeqDeltaBuffers* out) {
    if (!h || !out) return fail(RMJ_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(h->cfg.device));
    const size_t n = h->cfg.n_games;
    const size_t sz[7] = {n * 4 * RMJ_SEQ_SPARSE * 2, n * 4, n * 4 * 12 * 4, n * 4 * RMJ_SEQ_DELTA_PROG * 5 * 2, n * 4 * 2,
                          n * 4 * RMJ_SEQ_CAND * 4 * 2, n * 4};
    size_t off[8] = {0};
    for (int i = 0; i < 7; i++) off[i + 1] = off[i] + ((sz[i] + 255) & ~(size_t)255);
    void* sp;
    int rc = scratch_for(h, off[7], &sp);
    if (rc) return rc;
    char* b = (char*)sp;
    RmjSeqDeltaBuffers d{(uint16_t*)(b + off[0]), (uint8_t*)(b + off[1]), (float*)(b + off[2]), (uint16_t*)(b + off[3]), (uint16_t*)(b + off[4]),
                         (uint16_t*)(b + off[5]), (uint8_t*)(b + off[6])};
    if ((rc = rmj_encode_seq_delta_device(h, game_style, &d))) return rc;
    HIPCHK(hipStreamSynchronize(h->stream));
    void* dst[7] = {out->sparse, out->n_sparse, out->numeric, out->progression, out->n_progression, out->candidates, out->n_candidates};
    for (int i = 0; i < 7; i++) {
        if (!dst[i]) return fail(RMJ_ERR_ARG, "null output array");
        HIPCHK(hipMemcpy(dst[i], b + off[i], sz[i], hipMemcpyDeviceToHost));
    }
    return RMJ_OK;
}

int rmj_encode_extended_device(rmj_handle h, int only_active, float* d_out) { return launch_encode(h, only_active, d_out, true); }
int rmj_encode_extended(rmj_handle h, int only_active, float* out) {
    if (!h || !out) return fail(RMJ_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(h->cfg.device));
    size_t bytes = (size_t)h->cfg.n_games * 4 * ENC_EXT_CH * (h->cfg.game_mode >= 3 ? ENC_W3 : ENC_W4) * sizeof(float);
    void* sp;
    int rc = scratch_for(h, bytes, &sp);
    if (rc) return rc;
    float* d = (float*)sp;
    if ((rc = rmj_encode_extended_device(h, only_active, d))) return rc;
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(out, d, bytes, hipMemcpyDeviceToHost));
    return RMJ_OK;
}

// ---- shanten (row A7) -------------------------------------------------------------------------------
static int shanten_tables_for(int device, ShantenTables* out) {
    static ShantenTables cache[64];
    static bool have[64] = {false};
    static std::mutex mu;   // handles are created from several host threads (MultiGpuVecEnv: one per shard, shards may share a device)
    if (device < 0 || device >= 64) return fail(RMJ_ERR_ARG, "device ordinal");
    std::lock_guard<std::mutex> lock(mu);
    if (!have[device]) {
        const ShantenHostTables& H = shanten_host_tables();
        uint64_t *ds, *dh;
        uint32_t *r9, *r7;
        HIPCHK(hipMalloc(&ds, H.suit.size() * 8));
        HIPCHK(hipMalloc(&dh, H.honor.size() * 8));
        HIPCHK(hipMalloc(&r9, H.rank9.size() * 4));
        HIPCHK(hipMalloc(&r7, H.rank7.size() * 4));
        HIPCHK(hipMemcpy(ds, H.suit.data(), H.suit.size() * 8, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(dh, H.honor.data(), H.honor.size() * 8, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(r9, H.rank9.data(), H.rank9.size() * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(r7, H.rank7.data(), H.rank7.size() * 4, hipMemcpyHostToDevice));
        uint32_t* r2;
        uint64_t* v6;
        HIPCHK(hipMalloc(&r2, H.r2.size() * 4));
        HIPCHK(hipMalloc(&v6, H.v6.size() * 8));
        HIPCHK(hipMemcpy(r2, H.r2.data(), H.r2.size() * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(v6, H.v6.data(), H.v6.size() * 8, hipMemcpyHostToDevice));
        cache[device].suit = ds; cache[device].honor = dh; cache[device].rank9 = r9; cache[device].rank7 = r7;
        cache[device].r2 = r2; cache[device].v6 = v6;
        have[device] = true;
    }
    *out = cache[device];
    return RMJ_OK;
}
int rmj_shanten(int device, const uint8_t* counts, uint32_t n, int sanma, int8_t* out) {
    DevTmp tmp;
    if (!counts || !out) return fail(RMJ_ERR_ARG, "null argument");
    int rc = ensure_device(device);
    if (rc) return rc;
    if (n == 0) return RMJ_OK;
    ShantenTables T;
    if ((rc = shanten_tables_for(device, &T))) return rc;
    uint8_t* d_c;
    int8_t* d_o;
    HIPCHK(tmp.alloc(&d_c, (size_t)n * 34));
    HIPCHK(tmp.alloc(&d_o, n));
    HIPCHK(hipMemcpy(d_c, counts, (size_t)n * 34, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_shanten, dim3((n + 255) / 256), dim3(256), 0, 0, T, d_c, n, sanma, d_o);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out, d_o, n, hipMemcpyDeviceToHost));
    return RMJ_OK;
}

static int run_ukeire(int device, const uint8_t* counts, const uint8_t* visible, uint32_t n, int sanma, int mode, uint32_t* out) {
    DevTmp tmp;
    if (!counts || !out || (mode == 1 && !visible)) return fail(RMJ_ERR_ARG, "null argument");
    int rc = ensure_device(device);
    if (rc) return rc;
    if (n == 0) return RMJ_OK;
    ShantenTables T;
    if ((rc = shanten_tables_for(device, &T))) return rc;
    uint8_t *d_c = nullptr, *d_v = nullptr;
    uint32_t* d_o = nullptr;
    HIPCHK(tmp.alloc(&d_c, (size_t)n * 34));
    HIPCHK(tmp.alloc(&d_o, (size_t)n * 4));
    HIPCHK(hipMemcpy(d_c, counts, (size_t)n * 34, hipMemcpyHostToDevice));
    if (mode == 1) {
        HIPCHK(tmp.alloc(&d_v, (size_t)n * 34));
        HIPCHK(hipMemcpy(d_v, visible, (size_t)n * 34, hipMemcpyHostToDevice));
    }
    hipLaunchKernelGGL(k_ukeire, dim3((n + 3) / 4), dim3(256), 0, 0, T, (const uint8_t*)d_c, (const uint8_t*)d_v, n, sanma, mode, d_o);
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out, d_o, (size_t)n * 4, hipMemcpyDeviceToHost));
    return RMJ_OK;
}
int rmj_effective_tiles(int device, const uint8_t* counts, uint32_t n, int sanma, uint32_t* out) {
    return run_ukeire(device, counts, nullptr, n, sanma, 0, out);
}
int rmj_best_ukeire(int device, const uint8_t* counts, const uint8_t* visible, uint32_t n, int sanma, uint32_t* out) {
    return run_ukeire(device, counts, visible, n, sanma, 1, out);
}

// ---- MJAI event ingestion (row N1) --------------------------------------------------------------
int rmj_apply_events(rmj_handle h, const RmjEvent* events) {
    if (!h || !events) return fail(RMJ_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(h->cfg.device));
    const size_t bytes = (size_t)h->cfg.n_games * 3 * sizeof(RmjEvent);
    void* sp;
    int rcs = scratch_for(h, bytes, &sp);
    if (rcs) return rcs;
    RmjEvent* d_ev = (RmjEvent*)sp;
    HIPCHK(hipMemcpyAsync(d_ev, events, bytes, hipMemcpyHostToDevice, h->stream));
    if (h->cfg.game_mode >= 3) hipLaunchKernelGGL(rmj3::k_apply_event, game_grid(h->cfg.n_games), dim3(256), 0, h->stream, (const Env*)h->d_env, (const RmjEvent*)d_ev);
    else hipLaunchKernelGGL(rmj4::k_apply_event, game_grid(h->cfg.n_games), dim3(256), 0, h->stream, (const Env*)h->d_env, (const RmjEvent*)d_ev);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    return RMJ_OK;
}

// ---- measurement -----------------------------------------------------------------------------
static int bench_rollout_impl(rmj_handle h, uint64_t policy_seed, uint32_t warmup, uint32_t steps, RmjBenchResult* out, bool count, int pol = 0, uint32_t rate = 0u) {
    if (!h || !out) return fail(RMJ_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(h->cfg.device));
    int rc = warmup ? step_policy_impl(h, policy_seed, warmup, 1, pol, rate) : RMJ_OK;
    if (rc) return rc;
    uint64_t before = 0, after = 0, full0 = 0, full1 = 0;
    if (count && ((rc = rmj_total_steps(h, &before)) || (rc = rmj_total_full_path(h, &full0)))) return rc;
    for (int i = 0; i < 2; i++)
        if (!h->ev_time[i]) HIPCHK(hipEventCreate(&h->ev_time[i]));
    hipEvent_t e0 = h->ev_time[0], e1 = h->ev_time[1];
    HIPCHK(hipEventRecord(e0, h->stream));
    if ((rc = step_policy_impl(h, policy_seed, steps, 1, pol, rate))) return rc;
    HIPCHK(hipEventRecord(e1, h->stream));
    // (polling, not hipEventSynchronize: a blocked host thread is woken 10-20 us after the event completes - 2 % of the driver's 20-step window, which ends with this wait)
    for (;;) {
        const hipError_t qe = hipEventQuery(e1);
        if (qe == hipSuccess) break;
        if (qe != hipErrorNotReady) return fail(RMJ_ERR_HIP, std::string("hipEventQuery: ") + hipGetErrorString(qe));
    }
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    if (count && ((rc = rmj_total_steps(h, &after)) || (rc = rmj_total_full_path(h, &full1)))) return rc;
    out->total_ms = ms;
    const uint32_t fl = (uint32_t)rollout_streams(h, steps);
    const bool fused = h->quad >= 2 && steps >= 2 && h->want_streams >= 2;
    out->launches = fused ? 1u : steps * fl;
    out->step_kernel_ms = steps ? ms / steps : 0.0;  // each stream runs `steps` launches back to back during `ms`
    out->env_steps = after - before;
    out->launches_in_flight = fl;
    out->full_path_steps = full1 - full0;
    out->queued = rollout_queued(h, steps, pol) ? 1u : 0u;
    out->reserved = 0u;
    return RMJ_OK;
}
int rmj_bench_rollout(rmj_handle h, uint64_t policy_seed, uint32_t warmup, uint32_t steps, RmjBenchResult* out) {
    return bench_rollout_impl(h, policy_seed, warmup, steps, out, true);
}
// The timed region alone: HIP events on the handle's stream around rmj_step_random(h, policy_seed, steps, auto_reset = 1), nothing
// else issued or synchronised (the step / full-path counters of rmj_bench_rollout cost four small launches and four host round
// trips - a fifth of a 20-step rollout).  env_steps / full_path_steps are left 0: read rmj_total_steps / rmj_total_full_path
// outside the region.
int rmj_time_rollout(rmj_handle h, uint64_t policy_seed, uint32_t steps, RmjBenchResult* out) {
    return bench_rollout_impl(h, policy_seed, 0u, steps, out, false);
}
// the same around rmj_step_random_encode(h, policy_seed, steps, 1, 2, d_out) (BASELINE configs[4]); queued: the one-launch rollout ran as tickets
int rmj_time_rollout_encode(rmj_handle h, uint64_t policy_seed, uint32_t steps, float* d_out, RmjBenchResult* out) {
    DevTmp tmp;
    if (!h || !out || !d_out) return fail(RMJ_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(h->cfg.device));
    hipEvent_t e0, e1;
    HIPCHK(tmp.event(&e0));
    HIPCHK(tmp.event(&e1));
    HIPCHK(hipEventRecord(e0, h->stream));
    int rc = rmj_step_random_encode(h, policy_seed, steps, 1, 2, d_out);
    if (rc) return rc;
    HIPCHK(hipEventRecord(e1, h->stream));
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    memset(out, 0, sizeof(*out));
    const bool fused = h->enc_fused && h->quad >= 2 && h->want_streams >= 2 && steps >= 2;
    const uint32_t quads = (h->cfg.n_games + 3u) / 4u, chunk = rollout_chunk(h, steps);
    out->total_ms = ms;
    out->step_kernel_ms = steps ? ms / steps : 0.0;
    out->launches = fused ? 1u : 2u * steps;
    out->launches_in_flight = 1u;
    out->queued = (fused && h->queue_chunk > 0 && h->max_xcc_id <= 7u && steps >= 2u * chunk && h->q_slots_enc &&
                   (h->queue_force ? quads >= 64u : (quads > h->q_slots_enc && quads < 8u * h->q_slots_enc))) ? 1u : 0u;
    return RMJ_OK;
}
// the same around rmj_step_greedy(h, policy_seed, steps, 1, call_rate_256)
int rmj_time_rollout_greedy(rmj_handle h, uint64_t policy_seed, uint32_t steps, uint32_t call_rate_256, RmjBenchResult* out) {
    return bench_rollout_impl(h, policy_seed, 0u, steps, out, false, 1, call_rate_256);
}
int rmj_set_rollout_streams(rmj_handle h, int k) {
    if (!h || k < 1 || k > RMJ_MAX_ROLLOUT_STREAMS) return fail(RMJ_ERR_ARG, "rollout streams must be 1..8");
    h->want_streams = k;
    return RMJ_OK;
}
// device memory / synchronisation for a harness that has no other way to HIP (bench.py runs one GPU without torch)
int rmj_bench_device_alloc(int device, uint64_t bytes, void** out) {
    if (!out) return fail(RMJ_ERR_ARG, "null argument");
    int rc = ensure_device(device);
    if (rc) return rc;
    HIPCHK(hipMalloc(out, (size_t)bytes));
    HIPCHK(hipMemset(*out, 0, (size_t)bytes));
    return RMJ_OK;
}
int rmj_bench_device_free(int device, void* p) {
    int rc = ensure_device(device);
    if (rc) return rc;
    HIPCHK(hipFree(p));
    return RMJ_OK;
}
int rmj_bench_device_sync(int device) {
    int rc = ensure_device(device);
    if (rc) return rc;
    HIPCHK(hipDeviceSynchronize());
    return RMJ_OK;
}
int rmj_total_full_path(rmj_handle h, uint64_t* total) {
    if (!h || !total) return fail(RMJ_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(hipMemsetAsync(h->d_counter, 0, 8, h->stream));
    hipLaunchKernelGGL(k_sum_full, dim3(256), dim3(256), 0, h->stream, h->d.core, h->cfg.n_games, h->d_counter);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    unsigned long long v = 0;
    HIPCHK(hipMemcpy(&v, h->d_counter, 8, hipMemcpyDeviceToHost));
    *total = v;
    return RMJ_OK;
}
int rmj_random_actions_device(rmj_handle h, uint64_t policy_seed, rmj_action_t* d_actions) {
    if (!h || !d_actions) return fail(RMJ_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(h->cfg.device));
    const uint32_t n = h->cfg.n_games;
    hipLaunchKernelGGL(k_random_actions, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->d, policy_seed, (uint64_t*)d_actions);
    HIPCHK(hipGetLastError());
    return RMJ_OK;
}
// The unfused counterpart of rmj_bench_rollout: every step is one policy launch (k_random_actions writes packed actions
// to a device buffer) followed by one step launch that VALIDATES those actions against the stored legal lists like
// GameState::step does for an external agent (state/mod.rs:339-402); finished games restart.  One stream, whole batch.
int rmj_bench_rollout_validated(rmj_handle h, uint64_t policy_seed, uint32_t warmup, uint32_t steps, RmjBenchResult* out) {
    DevTmp tmp;
    if (!h || !out) return fail(RMJ_ERR_ARG, "null argument");
    HIPCHK(hipSetDevice(h->cfg.device));
    const uint32_t n = h->cfg.n_games;
    uint64_t before = 0, after = 0, full0 = 0, full1 = 0;
    hipEvent_t e0, e1;
    HIPCHK(tmp.event(&e0));
    HIPCHK(tmp.event(&e1));
    int rc;
    for (uint32_t s = 0; s < warmup + steps; s++) {
        if (s == warmup) {
            if ((rc = rmj_total_steps(h, &before)) || (rc = rmj_total_full_path(h, &full0))) return rc;
            HIPCHK(hipEventRecord(e0, h->stream));
        }
        hipLaunchKernelGGL(k_random_actions, dim3((n + 255) / 256), dim3(256), 0, h->stream, h->d, policy_seed, h->d_actions);
        launch_step_range(h, h->stream, h->d_actions, 0ull, STEP_F_AUTORESET, 0u, n);
    }
    HIPCHK(hipEventRecord(e1, h->stream));
    HIPCHK(hipEventSynchronize(e1));
    HIPCHK(hipGetLastError());
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    if ((rc = rmj_total_steps(h, &after)) || (rc = rmj_total_full_path(h, &full1))) return rc;
    out->total_ms = ms;
    out->launches = steps;
    out->step_kernel_ms = steps ? ms / steps : 0.0;  // policy launch + step launch
    out->env_steps = after - before;
    out->launches_in_flight = 1;
    out->full_path_steps = full1 - full0;
    out->queued = 0u;
    out->reserved = 0u;
    return RMJ_OK;
}
// Average duration of one encoder launch over `reps` back-to-back launches (HIP events on the handle's stream): the
// roofline figure of BASELINE's feature-output configuration.  d_out like rmj_encode_device / rmj_encode_extended_device.
int rmj_bench_encode(rmj_handle h, int extended, int only_active, float* d_out, uint32_t reps, double* avg_ms) {
    DevTmp tmp;
    if (!h || !d_out || !avg_ms || reps == 0) return fail(RMJ_ERR_ARG, "bad argument");
    HIPCHK(hipSetDevice(h->cfg.device));
    hipEvent_t e0, e1;
    HIPCHK(tmp.event(&e0));
    HIPCHK(tmp.event(&e1));
    int rc;
    if ((rc = launch_encode(h, only_active, d_out, extended != 0))) return rc;  // warm-up
    HIPCHK(hipEventRecord(e0, h->stream));
    for (uint32_t i = 0; i < reps; i++)
        if ((rc = launch_encode(h, only_active, d_out, extended != 0))) return rc;
    HIPCHK(hipEventRecord(e1, h->stream));
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    *avg_ms = (double)ms / reps;
    return RMJ_OK;
}
// Kernel-gate benchmark (the groups of riichienv-core/benches/agari_bench.rs:142-376): average duration of
// ONE launch of a hand-math kernel over `n` device-resident inputs, HIP events around `reps` back-to-back launches after a warm-up
// launch; nothing is copied back.  which: 0 = k_eval_hands (a = RmjHandCase[n]), 1 = k_agari_counts (a = counts[n][34]),
// 2 = k_shanten, 3 = k_ukeire effective tiles, 4 = k_ukeire best ukeire (b = visible[n][34]), 5 = k_score (a = han, fu, oya, tsumo,
// num_players as five byte arrays of n one after the other; b = honba u32[n]).
int rmj_bench_hand_kernel(int device, int which, const void* a, const void* b, uint32_t n, int sanma, uint32_t reps, double* avg_ms) {
    DevTmp tmp;
    if (!a || !avg_ms || n == 0 || reps == 0 || which < 0 || which > 5 || ((which == 4 || which == 5) && !b)) return fail(RMJ_ERR_ARG, "bad argument");
    int rc = ensure_device(device);
    if (rc) return rc;
    ShantenTables T;
    if ((rc = shanten_tables_for(device, &T))) return rc;
    const size_t in_a = which == 0 ? (size_t)n * sizeof(RmjHandCase) : (which == 5 ? (size_t)n * 5 : (size_t)n * 34);
    const size_t in_b = which == 4 ? (size_t)n * 34 : (which == 5 ? (size_t)n * 4 : 0);
    uint8_t *d_a = nullptr, *d_b = nullptr, *d_o = nullptr;
    HIPCHK(tmp.alloc(&d_a, in_a));
    HIPCHK(hipMemcpy(d_a, a, in_a, hipMemcpyHostToDevice));
    if (in_b) {
        HIPCHK(tmp.alloc(&d_b, in_b));
        HIPCHK(hipMemcpy(d_b, b, in_b, hipMemcpyHostToDevice));
    }
    HIPCHK(tmp.alloc(&d_o, (size_t)n * (which == 0 ? sizeof(RmjHandResult) : 16)));
    hipEvent_t e0, e1;
    HIPCHK(tmp.event(&e0));
    HIPCHK(tmp.event(&e1));
    auto launch = [&]() {
        switch (which) {
            case 0: hipLaunchKernelGGL(k_eval_hands, dim3((n + 15u) / 16u), dim3(256), 0, 0, (const RmjHandCase*)d_a, n, (RmjHandResult*)d_o); break;
            case 1: hipLaunchKernelGGL(k_agari_counts, dim3((n + 15) / 16), dim3(256), 0, 0, (const uint8_t*)d_a, n, d_o, d_o + n, (uint64_t*)(d_o + 8 * (size_t)n)); break;
            case 2: hipLaunchKernelGGL(k_shanten, dim3((n + 255) / 256), dim3(256), 0, 0, T, (const uint8_t*)d_a, n, sanma, (int8_t*)d_o); break;
            case 3: hipLaunchKernelGGL(k_ukeire, dim3((n + 3) / 4), dim3(256), 0, 0, T, (const uint8_t*)d_a, (const uint8_t*)nullptr, n, sanma, 0, (uint32_t*)d_o); break;
            case 4: hipLaunchKernelGGL(k_ukeire, dim3((n + 3) / 4), dim3(256), 0, 0, T, (const uint8_t*)d_a, (const uint8_t*)d_b, n, sanma, 1, (uint32_t*)d_o); break;
            default: hipLaunchKernelGGL(k_score, dim3((n + 255) / 256), dim3(256), 0, 0, (const uint8_t*)d_a, (const uint8_t*)d_a + n, (const uint8_t*)d_a + 2 * (size_t)n,
                                        (const uint8_t*)d_a + 3 * (size_t)n, (const uint32_t*)d_b, (const uint8_t*)d_a + 4 * (size_t)n, n, (uint32_t*)d_o); break;
        }
    };
    launch();
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipEventRecord(e0, 0));
    for (uint32_t i = 0; i < reps; i++) launch();
    HIPCHK(hipEventRecord(e1, 0));
    HIPCHK(hipEventSynchronize(e1));
    HIPCHK(hipGetLastError());
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    *avg_ms = (double)ms / reps;
    return RMJ_OK;
}
// Observation outputs of ONE game (sampled parity checks at batch sizes where fetching every game's lists is wasteful)
int rmj_peek_outputs(rmj_handle h, uint32_t game, rmj_action_t* legal /*[4][64]*/, uint8_t* counts /*[4]*/, uint8_t* mask /*[4][82]*/,
                     uint64_t* waits /*[4]*/, uint32_t* status) {
    if (!h || !legal || !counts || !mask || !waits || !status) return fail(RMJ_ERR_ARG, "null argument");
    if (game >= h->cfg.n_games) return fail(RMJ_ERR_RANGE, "game index out of range");
    HIPCHK(hipSetDevice(h->cfg.device));
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(legal, h->d.legal + (size_t)game * 4 * RMJ_MAX_LEGAL, 4 * RMJ_MAX_LEGAL * sizeof(uint64_t), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(counts, h->d.nlegal + (size_t)game * 4, 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(mask, h->d.mask + (size_t)game * 4 * 82, 4 * 82, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(waits, h->d.waits + (size_t)game * 4, 4 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(status, h->d.status + game, 4, hipMemcpyDeviceToHost));
    return RMJ_OK;
}

#ifdef RMJ_CUTS
// instruction accounting build only (scripts/valu_sections.py): waves end at PROF mark `cut` (-1: run to the end)
extern "C" int rmj_prof_set_cut(int cut, int cut2, int cut3) {
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(rmj::g_cut3), &cut3, sizeof(cut3)));
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(rmj::g_cut), &cut, sizeof(cut)));
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(rmj::g_cut2), &cut2, sizeof(cut2)));
    return RMJ_OK;
}
#endif

#if defined(RMJ_CUTS) || defined(RMJ_CENSUS)
extern "C" int rmj_prof_bail_census(uint32_t* out32, int reset) {
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpyFromSymbol(out32, HIP_SYMBOL(rmj::g_bail_reason), 32 * sizeof(uint32_t)));
    if (reset) {
        uint32_t z[32] = {0};
        HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(rmj::g_bail_reason), z, sizeof(z)));
    }
    return RMJ_OK;
}
#endif


#ifdef RMJ_RE_PROF
extern "C" int rmj_debug_re_prof(unsigned long long* out8, int reset) {
    unsigned long long z[24] = {0};
    if (out8) HIPCHK(hipMemcpyFromSymbol(out8, HIP_SYMBOL(rmj::g_re_prof), sizeof(z)));
    if (reset) HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(rmj::g_re_prof), z, sizeof(z)));
    return RMJ_OK;
}
#endif
#ifdef RMJ_DEBUG_HWID
// debugging build only: HW ids / times of the waves of the last k_step4_act_enc launch (4 u64 per block)
extern "C" int rmj_debug_hwid_fetch(uint64_t* out, uint32_t n_blocks) {
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(rmj::g_dbg_hwid), (size_t)(n_blocks < RMJ_DEBUG_HWID ? n_blocks : RMJ_DEBUG_HWID) * 32));
    return RMJ_OK;
}
#endif

#ifdef RMJ_TL4
// timeline build only (scripts/timeline4.py): rows of the waves of the last launch of k_step4<false> (allocates on first call)
int rmj_tl4_fetch(uint64_t* out, uint32_t n_waves) {
    static unsigned long long* buf = nullptr;
    static uint32_t cap = 0;
    HIPCHK(hipDeviceSynchronize());
    if (!buf) {
        cap = n_waves;
        HIPCHK(hipMalloc(&buf, (size_t)cap * RMJ_TL4_ROW * 8));
        HIPCHK(hipMemset(buf, 0, (size_t)cap * RMJ_TL4_ROW * 8));
        HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(rmj::g_tl4), &buf, sizeof(buf)));
        return RMJ_OK;
    }
    HIPCHK(hipMemcpy(out, buf, (size_t)(n_waves < cap ? n_waves : cap) * RMJ_TL4_ROW * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemset(buf, 0, (size_t)cap * RMJ_TL4_ROW * 8));   // blocks that leave at once (heavy-first order) write nothing
    HIPCHK(hipDeviceSynchronize());
    return RMJ_OK;
}
#endif

#ifdef RMJ_QTL
// ticket timeline build only (scripts/timeline_queue.py): rows of the waves of the last k_step4_queue launch (allocates on first call)
int rmj_qtl_fetch(uint64_t* out, uint32_t n_waves) {
    static unsigned long long* buf = nullptr;
    static uint32_t cap = 0;
    HIPCHK(hipDeviceSynchronize());
    if (!buf) {
        cap = n_waves;
        HIPCHK(hipMalloc(&buf, (size_t)cap * RMJ_QTL_ROW * 8));
        HIPCHK(hipMemset(buf, 0, (size_t)cap * RMJ_QTL_ROW * 8));
        HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(rmj::g_qtl), &buf, sizeof(buf)));
        return RMJ_OK;
    }
    HIPCHK(hipMemcpy(out, buf, (size_t)(n_waves < cap ? n_waves : cap) * RMJ_QTL_ROW * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemset(buf, 0, (size_t)cap * RMJ_QTL_ROW * 8));
    HIPCHK(hipDeviceSynchronize());
    return RMJ_OK;
}
#endif

#ifdef RMJ_PROFILE
// profiling build only (scripts/prof_sections.py): per-section wave cycles / visit counts of k_step, summed over games
int rmj_prof_fetch(uint32_t n_games, uint64_t* cyc, uint64_t* cnt, int reset) {
    static uint32_t* buf = nullptr;
    HIPCHK(hipDeviceSynchronize());
    if (!buf) {
        HIPCHK(hipMalloc(&buf, (size_t)n_games * 64 * 4));
        HIPCHK(hipMemset(buf, 0, (size_t)n_games * 64 * 4));
        HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(rmj::g_prof_buf), &buf, sizeof(buf)));
    }
    std::vector<uint32_t> h((size_t)n_games * 64);
    HIPCHK(hipMemcpy(h.data(), buf, h.size() * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < 32; i++) { cyc[i] = 0; cnt[i] = 0; }
    for (size_t g = 0; g < n_games; g++)
        for (int i = 0; i < 32; i++) { cyc[i] += h[g * 64 + i]; cnt[i] += h[g * 64 + 32 + i]; }
    if (reset) HIPCHK(hipMemset(buf, 0, (size_t)n_games * 64 * 4));
    return RMJ_OK;
}
#endif

}  // extern "C"
