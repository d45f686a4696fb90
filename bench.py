#!/usr/bin/env python3
"""bench.py — env.step()/s of the MI355X-native batched Riichi step path.

One "step" = one batched env.step over every game of the shard (device-side RandomAgent policy,
auto-reset of finished games), i.e. one launch of the step kernel.  Workload at N=1:
BASELINE.json configs[2] — 65 536 parallel 4p-red-half games, RandomAgent, MJAI logging on.
N>1: one process per GPU (torch.distributed.run), games sharded by global index, no collective
on the data path (weak scaling: 65 536 games per GPU).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_STEP_4P = 1688          # algorithmic bytes per env.step per 4P game
B_STEP_3P = 1280          # 3P: 2*512 + 12 + 180 + 64
HBM_PEAK = 8.0e12         # B/s, /opt/skills/guides/MI355X_MICROARCH.md (HBM3E peak, spec)


def cpu_baseline(game_mode, rule_bits, policy_seed, target_s=15.0):
    """Oracle (CPU restatement of riichienv-core, kind="port") on the host cores: bounded sample."""
    from oracle import oracle

    threads = os.cpu_count() or 1
    n_games = threads * 8
    # calibrate with a short run, then size the sample for ~target_s seconds
    steps, secs = oracle.bench_rollout(game_mode, rule_bits, False, n_games, 0, policy_seed, 200, threads)
    rate = steps / max(secs, 1e-9)
    per_game = int(max(200, min(200000, rate * target_s / n_games)))
    steps, secs = oracle.bench_rollout(game_mode, rule_bits, False, n_games, 0, policy_seed, per_game, threads)
    # one thread, a few seconds: the per-core rate beside the all-cores one
    s1, t1 = oracle.bench_rollout(game_mode, rule_bits, False, 8, 0, policy_seed, 200, 1)
    per1 = int(max(200, min(200000, (s1 / max(t1, 1e-9)) * 3.0 / 8)))
    s1, t1 = oracle.bench_rollout(game_mode, rule_bits, False, 8, 0, policy_seed, per1, 1)
    return {"value": steps / secs, "unit": "env.step/s", "cores": threads, "kind": "port",
            "sample": f"{n_games} games x {per_game} steps, {threads} threads, {secs:.1f}s, "
                      "oracle/ C++ restatement with MJAI logging on (Rust toolchain unavailable)",
            "single_thread": {"value": s1 / t1, "sample": f"8 games x {per1} steps, {t1:.1f}s"}}


def pmc_traffic(games, mode):
    """HBM bytes per k_step launch from the committed rocprofv3 PMC passes (profiles/r01_pmc_k_step.json: FETCH_SIZE and
    WRITE_SIZE collected in separate --pmc runs of this same command, FETCH doubled per the gfx950 note of the
    microarch guide).  bench.py cannot run the profiler itself; null when no matching profile is committed."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_k_step.json")
    try:
        with open(path) as f:
            d = json.load(f)
        if d.get("games_per_launch") == games and mode == 2:  # games = games of ONE launch
            return d["hbm_traffic"]["bytes_per_launch"], "profiles/r01_pmc_k_step.json"
    except (OSError, KeyError, ValueError):
        pass
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--games", type=int, default=65536, help="games per GPU")
    ap.add_argument("--mode", type=int, default=2, help="0/1/2 = 4p-red-single/east/half, 3/4/5 = 3p-red-single/east/half")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--encode", action="store_true",
                    help="also produce the feature tensor of the acting seats every step (BASELINE configs[4]: sanma with "
                         "feature-encoding tensor output); reported in config, the step kernel's roofline is unchanged")
    args = ap.parse_args()

    import torch

    from riichienv_amd import abi, shard, vecenv

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist

        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")

    policy_seed = 0xC0FFEE
    env = vecenv.VecRiichiEnv(args.games, game_mode=args.mode, seed=0, rule_bits=abi.RULE_TENHOU, device=local_rank,
                              game_offset=shard.shard_offset(rank, args.games), event_ring=64)
    env.reset()
    env.step_random(policy_seed, args.warmup, auto_reset=True)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    r_enc = None
    if args.encode:
        # k_step duration for the roofline object: single launches over all games (a policy that needs the features is a
        # barrier between steps, so this mode cannot keep several launches in flight); extra warm-up, outside the timed region
        rs = [env.bench_rollout(policy_seed, 0, 1) for _ in range(20)]
        r_enc = rs[0]
        r_enc.step_kernel_ms = sum(x.step_kernel_ms for x in rs) / len(rs)
        obs = torch.zeros((args.games, 4, 74, 27 if args.mode >= 3 else 34), dtype=torch.float32, device=f"cuda:{local_rank}")
    barrier()
    t0 = time.perf_counter()
    if args.encode:
        import ctypes as C

        before = env.total_steps()
        for _ in range(args.steps):   # one step launch + one encode launch per step, same stream, no host sync in between
            env.step_random(policy_seed, 1, auto_reset=True)
            vecenv._chk(env.L.rmj_encode_device(env.h, 2, C.c_void_p(obs.data_ptr())))
        r = r_enc
        r.env_steps = env.total_steps() - before   # synchronises the stream
        r.launches_in_flight = 1
    else:
        r = env.bench_rollout(policy_seed, 0, args.steps)   # exactly K steps of every game, HIP events on the handle's stream
    barrier()
    t1 = time.perf_counter()
    wall = t1 - t0
    steps_local = float(r.env_steps)
    wall, steps_total = shard.reduce_measurement(dist, wall, steps_local, device="cuda")

    if rank == 0:
        kernel_s = r.step_kernel_ms * 1e-3
        b_step = B_STEP_3P if args.mode >= 3 else B_STEP_4P
        # a device rollout runs as `in_flight` concurrent launches (halves of the batch on two streams, rmj_step_random):
        # bytes and duration are per launch, the bandwidth the chip delivers is in_flight launches' worth
        in_flight = max(1, int(r.launches_in_flight))
        games_per_launch = args.games // in_flight
        traffic, traffic_src = pmc_traffic(games_per_launch, args.mode)
        achieved = in_flight * b_step * games_per_launch / kernel_s
        out = {
            "metric": "env.step()/s (whole node) at 65 536 parallel 4p games; bit-exact MJAI parity",
            "value": steps_total / wall, "unit": "env.step/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": wall * 1e3 / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{args.games} parallel {['4p-red-single','4p-red-east','4p-red-half','3p-red-single','3p-red-east','3p-red-half'][args.mode]} "
                                   "games per GPU, device RandomAgent, auto-reset, MJAI logging on",
                       "games_per_gpu": args.games, "sharding": "by game index, no collectives",
                       "feature_tensor_output": bool(args.encode)},
            "roofline": {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK, "traffic": traffic, "traffic_unit": "bytes/launch",
                         "traffic_source": traffic_src, "kernel": "k_step",
                         "kernel_ms": r.step_kernel_ms, "bytes_per_launch": b_step * games_per_launch,
                         "games_per_launch": games_per_launch, "launches_in_flight": in_flight},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.mode, abi.RULE_TENHOU, policy_seed)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
