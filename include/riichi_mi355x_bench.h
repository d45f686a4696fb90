/*
 * riichi_mi355x_bench.h - measurement and test hooks of libriichi_mi355x.so.
 *
 * NOT part of the drop-in boundary (include/riichi_mi355x.h): these entry points time rollouts and kernels for bench.py, the
 * profiles under profiles/ and the test suite.  A host that replaces riichienv-core's step path never needs them.
 *
 * Environment variables read by rmj_create (latched per handle; experiments and tests only):
 *   RMJ_STEP4=0|1|2            0: one game per wavefront (k_step), 1: four games per wavefront with one launch per step,
 *                              2 (default): and device-policy rollouts as one launch
 *   RMJ_STEP_STREAMS=k         parts of a per-step rollout (rmj_set_rollout_streams)
 *   RMJ_QUEUE_CHUNK=n          steps per ticket of the fused rollout (0: no tickets), RMJ_QUEUE_MIN_CHUNK, RMJ_QUEUE_FORCE=1 (tickets
 *                              for any batch of >= 64 quads: tests)
 *   RMJ_QUEUE_TEST_SKIP_XCDS=mask   TEST HOOK: waves on these XCDs leave the ticket kernel at once, as if the dispatcher had given
 *                              that XCD no block - exercises the fix-up launch (tests/test_gpu_fullsize.py)
 *   RMJ_ENC_STREAMS, RMJ_ENC_PARTS_QUAD, RMJ_ENC_FUSED   schedules of the step + encode rollout
 *   RMJ_EXTRA_LDS=bytes        occupancy experiments of the per-step kernel
 */
#ifndef RIICHI_MI355X_BENCH_H
#define RIICHI_MI355X_BENCH_H

#include "riichi_mi355x.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct RmjBenchResult {
    double total_ms;      /* HIP-event time over the timed region (stream of the handle) */
    double step_kernel_ms;/* average duration of one step of all games: total_ms / steps (a fused rollout is one launch of
                             `steps` steps; with per-step launches on several streams, the launches of one stream run back
                             to back over the timed region) */
    uint64_t env_steps;   /* sum over games of step calls that advanced the game */
    uint32_t launches;    /* step-kernel launches in the timed region (1 for a fused rollout) */
    uint32_t launches_in_flight; /* streams the rollout ran on (parts of the batch, rmj_step_random); 1 = one stream */
    uint64_t full_path_steps; /* game-steps of the timed region that left the fast path of the step kernel (round ends,
                                 yaku evaluation, kans, riichi, restarts of finished games) */
    uint32_t queued;      /* 1: the fused rollout ran as (quad, chunk) tickets (kernel k_step4_queue), 0: k_step4<true> or per step */
    uint32_t reserved;
} RmjBenchResult;
int rmj_bench_rollout(rmj_handle h, uint64_t policy_seed, uint32_t warmup, uint32_t steps, RmjBenchResult* out);
/* The timed region alone: HIP events around rmj_step_random(h, policy_seed, steps, 1) on the handle's stream and nothing else
 * (no counter launches, no host round trips besides the final event wait); env_steps / full_path_steps stay 0 - read
 * rmj_total_steps / rmj_total_full_path before and after.  What bench.py times. */
int rmj_time_rollout(rmj_handle h, uint64_t policy_seed, uint32_t steps, RmjBenchResult* out);
/* the same around rmj_step_random_encode(h, policy_seed, steps, 1, 2, d_out): launches = 1 when the rollout ran as one launch */
int rmj_time_rollout_encode(rmj_handle h, uint64_t policy_seed, uint32_t steps, float* d_out, RmjBenchResult* out);
/* the same around rmj_step_greedy(h, policy_seed, steps, 1, call_rate_256) */
int rmj_time_rollout_greedy(rmj_handle h, uint64_t policy_seed, uint32_t steps, uint32_t call_rate_256, RmjBenchResult* out);
/* The unfused counterpart: per step one policy launch (packed actions into a device buffer) and one step launch that
 * validates them against the stored legal lists like GameState::step does for an external agent (state/mod.rs:339-402);
 * finished games restart; one stream, the whole batch per launch.  step_kernel_ms = policy + step launch. */
int rmj_bench_rollout_validated(rmj_handle h, uint64_t policy_seed, uint32_t warmup, uint32_t steps, RmjBenchResult* out);
/* Kernel-gate benchmark of the hand-math kernels (the groups of the reference's
 * riichienv-core/benches/agari_bench.rs:142-376): average duration (ms) of ONE launch over n device-resident inputs (uploaded
 * once, nothing copied back), HIP events around `reps` launches.  which: 0 = rmj_eval_hands (a = RmjHandCase[n]),
 * 1 = rmj_agari_counts (a = counts[n][34]), 2 = rmj_shanten, 3 = rmj_effective_tiles, 4 = rmj_best_ukeire (b = visible[n][34]),
 * 5 = rmj_calculate_score (a = han | fu | is_oya | is_tsumo | num_players, five byte arrays of n; b = honba u32[n]). */
int rmj_bench_hand_kernel(int device, int which, const void* a, const void* b, uint32_t n, int sanma, uint32_t reps, double* avg_ms);
/* Average duration (ms) of one encoder launch over `reps` back-to-back launches, HIP events on the handle's stream;
 * extended = 0: rmj_encode_device, 1: rmj_encode_extended_device (same d_out / only_active meaning). */
int rmj_bench_encode(rmj_handle h, int extended, int only_active, float* d_out, uint32_t reps, double* avg_ms);
/* the same for rmj_encode_compact_device (slot scan + encoder launch) */
int rmj_bench_encode_compact(rmj_handle h, float* d_out, int32_t* d_index, uint32_t capacity, uint32_t* d_count, uint32_t reps, double* avg_ms);
/* Device memory and a device-wide synchronisation for a harness that has no other way to HIP: bench.py runs its one-GPU legs without
 * torch (zero-filled allocation; hipDeviceSynchronize). */
int rmj_bench_device_alloc(int device, uint64_t bytes, void** out);
int rmj_bench_device_free(int device, void* p);
int rmj_bench_device_sync(int device);
/* Sum over games of the steps that took the full path of the step kernel since the handle was created. */
int rmj_total_full_path(rmj_handle h, uint64_t* total);


#ifdef __cplusplus
}
#endif
#endif /* RIICHI_MI355X_BENCH_H */
