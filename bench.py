#!/usr/bin/env python3
"""bench.py — env.step()/s of the MI355X-native batched Riichi step path.

One "step" = one batched env.step over every game of the shard (device-side RandomAgent policy,
auto-reset of finished games), i.e. one launch of the step kernel per part of the batch.  Workload at N=1:
BASELINE.json configs[2] — 65 536 parallel 4p-red-half games, RandomAgent, MJAI logging on.
N>1: one process per GPU, games sharded by global index, no collective on the data path (weak scaling: 65 536 games
per GPU).  `python bench.py --gpus N` without a torchrun environment starts the N rank processes itself
(torch.distributed.run as a child process, before this process has imported torch or touched HIP) and relays rank 0's
JSON line; under `python -m torch.distributed.run ... bench.py --gpus N` it is one of the ranks.
A one-GPU run imports no torch: device memory and synchronisation go through the library's own bench hooks.  The default
one-GPU line also carries a steady-state leg of the same environment (`long_rollout`), the other single-GPU configurations of
BASELINE.json (`configs`: [1], one shard of [3], [4] with the feature tensor) and a lossless log leg (`log_drain`); a multi-rank
line carries `ranks_seen` / `per_rank_value` (one all_gather) and rank 0's `cpu_baseline`.
"""
import argparse
import glob
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_STEP_4P = 1688          # algorithmic bytes per env.step per 4P game
B_STEP_3P = 1280          # 3P: 2*512 + 12 + 180 + 64
B_OBS_4P = 74 * 34 * 4    # Observation.encode(): 10 064 B per acting seat (docs/FEATURE_ENCODING.md:8-82)
B_OBS_3P = 74 * 27 * 4    # 7 992 B
HBM_PEAK = 8.0e12         # B/s, /opt/skills/guides/MI355X_MICROARCH.md (HBM3E peak, spec)
MODES = ['4p-red-single', '4p-red-east', '4p-red-half', '3p-red-single', '3p-red-east', '3p-red-half']
STEADY_MIN = 200          # steady-state window of >= 200 batched steps after the warm-up has reached round ends
PREROLL = 6000            # untimed steps of every game BEFORE the warm-up, whatever --warmup says: the first ~60 steps of a game
                          # cannot end a round and games started together end their rounds in bursts (after 600 steps a 20-step
                          # window still saw 4 % of the game-steps in the full path, after 2 000 steps 2.3 %, the long-run share
                          # is 1.7 %); after 6 000 steps (~50 rounds of RandomAgent play, 0.22 s at 65 536 games) a 20-step window
                          # holds the long-run share (measured: 0.0169 at 6 000, 0.0167 at 8 000, 0.0165 at 12 000; DESIGN.md
                          # section 11.1) - the steady state the metric is defined on.  Reported as "preroll_steps".


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--games", type=int, default=65536, help="games per GPU")
    ap.add_argument("--mode", type=int, default=2, help="0/1/2 = 4p-red-single/east/half, 3/4/5 = 3p-red-single/east/half")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the single-stream / validated-actions side measurements")
    ap.add_argument("--policy", choices=["random", "greedy"], default="random",
                    help="device policy of the rollout: the RandomAgent of BASELINE's configs, or the greedy policy that plays to win "
                         "(rmj_step_greedy: shanten-greedy discards, every win / riichi / kan / kita taken, calls at --call-rate)")
    ap.add_argument("--call-rate", type=int, default=64, help="greedy policy: pon / chi taken with probability call_rate / 256")
    ap.add_argument("--oversubscribe", action="store_true",
                    help="allow more ranks than GPUs: rank r works on GPU r %% visible GPUs and the ranks meet over gloo (RCCL refuses two ranks "
                         "on one device) - exercises the world > 1 path on a 1-GPU box; never a measurement")
    ap.add_argument("--preroll", type=int, default=PREROLL,
                    help="untimed steps before the warm-up that bring every game to steady state (0: time the opening phase)")
    ap.add_argument("--reference-rng", action="store_true",
                    help="deal every wall through the reference's own seed -> wall chain (RMJ_RULE_REFERENCE_RNG: PCG32 seed expansion, "
                         "ChaCha12, rand's shuffle, salt) instead of the build's counter-based shuffle")
    ap.add_argument("--in-process", type=int, default=0, metavar="K",
                    help="ONE process, K shards of --games games each through MultiGpuVecEnv (a host thread per shard, shard i on device i mod the "
                         "visible devices): the in-process counterpart of --gpus K; prints its own JSON line")
    ap.add_argument("--no-configs", action="store_true", help="skip the legs of the other BASELINE.json configurations (configs[1], [3], [4])")
    ap.add_argument("--padded-rows", action="store_true", help="--encode: rows padded to a multiple of 256 B instead of the dense [games][4][74][W] tensor")
    ap.add_argument("--encode", action="store_true",
                    help="also produce the feature tensor of the acting seats every step (BASELINE configs[4]: sanma with "
                         "feature-encoding tensor output): one step launch + one encode launch per step")
    return ap.parse_args(argv)


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launcher_command(args, port, script=None):
    """Command + environment that start `args.gpus` rank processes of this script on one node (one process per GPU).
    Pure function of its arguments: tests/test_bench_contract.py checks it without a GPU."""
    script = script or os.path.abspath(__file__)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), script,
           "--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup", str(args.warmup),
           "--games", str(args.games), "--mode", str(args.mode)]
    if args.preroll != PREROLL:
        cmd += ["--preroll", str(args.preroll)]
    if args.policy != "random":
        cmd += ["--policy", args.policy, "--call-rate", str(args.call_rate)]
    if args.no_cpu_baseline:
        cmd.append("--no-cpu-baseline")
    if args.no_extras:
        cmd.append("--no-extras")
    if args.encode:
        cmd.append("--encode")
    if args.padded_rows:
        cmd.append("--padded-rows")
    if getattr(args, "reference_rng", False):
        cmd.append("--reference-rng")
    if getattr(args, "oversubscribe", False):
        cmd.append("--oversubscribe")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this pool (RCCL barrier between the ranks)
    env["MASTER_ADDR"] = "127.0.0.1"
    return cmd, env


def launch_ranks(args):
    """--gpus N > 1 outside torchrun: start the ranks as children (this process never initialises the GPU, and nothing is
    exec'ed over a process that has), relay rank 0's JSON line, exit with the launcher's code."""
    cmd, env = launcher_command(args, free_port())
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in p.stdout.splitlines():
        s = ln.strip()
        if s.startswith("{") and '"metric"' in s:
            line = s
        elif s:
            print(s, file=sys.stderr)
    if p.returncode != 0 or line is None:
        print(f"bench.py: the {args.gpus}-rank launch failed (exit code {p.returncode})", file=sys.stderr)
        return p.returncode or 1
    print(line)
    return 0


_WORKER = """
import sys
sys.path.insert(0, sys.argv[1])
from oracle import oracle
mode, rule, pseed, first, games, per = (int(x) for x in sys.argv[2:8])
s, t = oracle.bench_rollout(mode, rule, False, games, first, pseed, per, 1)
print(s, t)
"""


def usable_cores():
    """CPUs this process may actually use: the smaller of the scheduler affinity and the cgroup CPU quota (cpu.max / cfs_quota_us).
    os.cpu_count() reports the host's hardware threads (256 on the GPU box) even when the container is limited to a few of them -
    round 2's "9.5 x one thread on 256 threads" was that limit, not allocator contention: 256 single-threaded worker PROCESSES
    reach the same total."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                       # cgroup v2: "<quota> <period>" or "max <period>"
            q, per = f.read().split()[:2]
            if q != "max":
                quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:   # cgroup v1
                q, per = float(f.read()), float(g.read())
                if q > 0:
                    quota = q / per
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = max(1, min(n, int(quota + 0.5)))
    return n


def cpu_baseline(game_mode, rule_bits, policy_seed, target_s=15.0):
    """Oracle (CPU restatement of riichienv-core, kind="port") on the host cores: bounded sample.  Must run BEFORE this process
    touches the GPU (it starts worker processes).  The all-cores figure comes from one single-threaded worker PROCESS per usable
    core - `cores` = usable_cores(), the container's CPU allowance, not the host's hardware threads (`hardware_threads`).  Reported
    beside it: the same workload as threads of one process (`threads_value`) and parallel_efficiency = value / (cores x
    single-thread rate)."""
    from oracle import oracle

    cores = usable_cores()
    # one thread, ~3 s: the per-core rate
    s1, t1 = oracle.bench_rollout(game_mode, rule_bits, False, 8, 0, policy_seed, 200, 1)
    per1 = int(max(200, min(200000, (s1 / max(t1, 1e-9)) * 3.0 / 8)))
    s1, t1 = oracle.bench_rollout(game_mode, rule_bits, False, 8, 0, policy_seed, per1, 1)
    rate1 = s1 / t1
    # all cores: `cores` processes x 8 games, sized for ~target_s seconds at 0.8 x the single-thread rate
    per_game = int(max(200, min(200000, 0.8 * rate1 * target_s / 8)))
    t0 = time.perf_counter()
    procs = [subprocess.Popen([sys.executable, "-c", _WORKER, ROOT, str(game_mode), str(rule_bits), str(policy_seed), str(8 * i), "8",
                               str(per_game)], stdout=subprocess.PIPE, text=True) for i in range(cores)]
    outs = [p.communicate()[0].split() for p in procs]
    wall = time.perf_counter() - t0
    if any(p.returncode != 0 or len(o) != 2 for p, o in zip(procs, outs)):
        raise RuntimeError("cpu_baseline: an oracle worker process failed")
    steps = sum(int(o[0]) for o in outs)
    secs = max(float(o[1]) for o in outs)
    # the same as threads of ONE process, ~5 s (what round 2 reported as the baseline)
    th_games = cores * 8
    per_th = max(100, per_game // 32)
    st, tt = oracle.bench_rollout(game_mode, rule_bits, False, th_games, 0, policy_seed, per_th, cores)
    return {"value": steps / secs, "unit": "env.step/s", "cores": cores, "hardware_threads": os.cpu_count() or 1, "kind": "port",
            "sample": f"{cores} worker processes x 8 games x {per_game} steps (one thread each), slowest worker {secs:.1f}s "
                      f"({wall:.1f}s with process start), oracle/ C++ restatement with MJAI logging on (Rust toolchain unavailable)",
            "parallel_efficiency": steps / secs / (cores * rate1),
            "single_thread": {"value": rate1, "sample": f"8 games x {per1} steps, {t1:.1f}s"},
            "threads_value": {"value": st / tt, "sample": f"{th_games} games x {per_th} steps as {cores} threads of one process, {tt:.1f}s "
                                                        "(the oracle allocates containers per step)"}}


def pmc_traffic(kernel, games_per_launch, mode, ran_as=None):
    """HBM bytes per launch of `kernel` from the newest committed rocprofv3 PMC summary that matches the launch shape
    (profiles/r*_pmc_<kernel>.json: FETCH_SIZE and WRITE_SIZE collected in separate --pmc runs of this same command,
    FETCH doubled per the gfx950 note of the microarch guide) and, with `ran_as`, the kernel the timed launch ran as (the
    summary's "kernel" field).  bench.py cannot run the profiler itself; null when no matching profile is committed."""
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_{kernel}.json")), reverse=True):
        try:
            with open(path) as f:
                d = json.load(f)
            if ran_as is not None and ran_as not in str(d.get("kernel", "")):
                continue
            if d.get("games_per_launch") == games_per_launch and d.get("mode", 2) == mode:
                t = d["hbm_traffic"]
                return t.get("bytes_per_step", t["bytes_per_launch"]), os.path.relpath(path, ROOT)
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def fused_kernel_name(r):
    """the kernel a multi-step device rollout ran as (rmj_step_random): long rollouts of batches between one and eight chip-fulls of
    waves are handed out as (quad, chunk) tickets to a chip-sized grid (k_step4_queue); otherwise every wave keeps its quad for the
    rollout (k_step4<true>).  The library reports which (RmjBenchResult.queued)."""
    return "k_step4_queue" if int(r.queued) else "k_step4<true>"


def workload_name(args):
    pol = "device RandomAgent" if args.policy == "random" else f"greedy device policy (calls {args.call_rate}/256)"
    s = f"{args.games} parallel {MODES[args.mode]} games per GPU, {pol}, auto-reset, MJAI logging on"
    if args.encode:
        s += ", Observation.encode() of every acting seat written to a resident tensor after every step" + \
             (" (rows padded to a multiple of 256 B)" if args.padded_rows else "")
    return s


def metric_name(args):
    """BASELINE.json's metric for the configuration it is quoted on; a descriptive one for every other workload."""
    if args.mode < 3 and args.games == 65536 and not args.encode and args.policy == "random":
        return "env.step()/s (whole node) at 65 536 parallel 4p games; bit-exact MJAI parity"
    seats = "3p" if args.mode >= 3 else "4p"
    extra = " with feature-encoding tensor output" if args.encode else ""
    if args.policy != "random":
        extra += f" under the {args.policy} device policy"
    return f"env.step()/s (whole node) at {args.games} parallel {seats} games per GPU{extra}; bit-exact MJAI parity"


class Dev:
    """HIP through the library's own bench hooks (include/riichi_mi355x_bench.h): a one-GPU run needs no torch."""

    def __init__(self, lib, device):
        import ctypes as C

        self.C, self.L, self.device, self.bufs = C, lib, int(device), []

    def alloc(self, nbytes):
        p = self.C.c_void_p()
        rc = self.L.rmj_bench_device_alloc(self.device, int(nbytes), self.C.byref(p))
        if rc:
            raise RuntimeError("rmj_bench_device_alloc failed: " + self.L.rmj_last_error().decode())
        self.bufs.append(p)
        return p.value

    def free_all(self):
        for p in self.bufs:
            self.L.rmj_bench_device_free(self.device, p)
        self.bufs = []

    def sync(self):
        if self.L.rmj_bench_device_sync(self.device):
            raise RuntimeError("rmj_bench_device_sync failed: " + self.L.rmj_last_error().decode())


def acting_seats(env):
    act, _, dn = env.status()
    return int(sum(bin(int(a)).count("1") for a, d in zip(act, dn) if not d))


def rollout_roofline(r, games, steps, sanma, encode=False, acting=0, greedy=False, enc_step_ms=None):
    """The roofline object of a timed device rollout (RmjBenchResult `r`): algorithmic bytes per launch (B_step x games x steps of the
    launch, + one Observation.encode() per acting seat and step with --encode) / the launch's duration from HIP events on the
    handle's stream / the HBM peak."""
    b_step = B_STEP_3P if sanma else B_STEP_4P
    steps_per_launch = 1
    if encode and int(r.launches) == 1 and steps > 1:
        in_flight, kernel_ms, steps_per_launch = 1, r.total_ms, steps          # step + encode rollout as ONE launch
    elif encode:
        in_flight, kernel_ms = 1, enc_step_ms
    elif int(r.launches) == 1 and steps > 1:
        in_flight, kernel_ms, steps_per_launch = 1, r.total_ms, steps          # every wave steps its games `steps` times in one launch
    else:
        in_flight, kernel_ms = max(1, int(r.launches_in_flight)), r.step_kernel_ms   # per-step launches on `in_flight` streams
    games_per_launch = games // in_flight
    kernel_name = fused_kernel_name(r) if steps_per_launch > 1 else "k_step4<false>"
    if encode and steps_per_launch > 1:
        kernel_name = "k_step4_queue_enc" if int(r.queued) else "k_step4_enc"
    if greedy:
        kernel_name += " (greedy)"
    bytes_per_launch = b_step * games_per_launch * steps_per_launch
    if encode and steps_per_launch > 1:
        bytes_per_launch += (B_OBS_3P if sanma else B_OBS_4P) * acting * steps_per_launch
    achieved = in_flight * bytes_per_launch / (kernel_ms * 1e-3)
    return {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": achieved / HBM_PEAK,
            "kernel": kernel_name, "kernel_ms": kernel_ms, "bytes_per_launch": bytes_per_launch, "bytes_per_game_step": b_step,
            "games_per_launch": games_per_launch, "steps_per_launch": steps_per_launch, "launches_in_flight": in_flight}


def config_leg(vecenv, abi, shard, dev, name, games, mode, steps, policy_seed, device, rank, encode=False, preroll=PREROLL, rule_extra=0):
    """One BASELINE.json configuration as a steady-state leg of its own: a fresh environment, `preroll` untimed steps, then `steps`
    (>= 300) timed steps of the device RandomAgent rollout between device-wide synchronisations."""
    sanma = mode >= 3
    env = vecenv.VecRiichiEnv(games, game_mode=mode, seed=0, rule_bits=abi.RULE_TENHOU | rule_extra, device=device,
                              game_offset=shard.shard_offset(rank, games), event_ring=64)
    env.reset()
    env.step_random(policy_seed, preroll, auto_reset=True)
    obs = None
    if encode:
        stride = 74 * (27 if sanma else 34)
        env.set_encode_row_stride(stride)
        obs = dev.alloc(games * 4 * stride * 4)
        env.time_rollout_encode(policy_seed, 5, obs)
    s0, f0 = env.total_steps(), env.total_full_path()
    dev.sync()
    t0 = time.perf_counter()
    r = env.time_rollout_encode(policy_seed, steps, obs) if encode else env.time_rollout(policy_seed, steps)
    dev.sync()
    t1 = time.perf_counter()
    made, full = env.total_steps() - s0, env.total_full_path() - f0
    acting = acting_seats(env) if encode else 0
    out = {"config": name, "workload": f"{games} parallel {MODES[mode]} games, device RandomAgent, auto-reset, MJAI logging on" +
                                       (", Observation.encode() of every acting seat after every step" if encode else ""),
           "value": made / (t1 - t0), "unit": "env.step/s", "steps": steps, "ms_per_step": (t1 - t0) * 1e3 / steps,
           "steady_state": bool(preroll >= 300 and steps >= STEADY_MIN), "preroll_steps": preroll, "full_path_frac": full / max(made, 1),
           "roofline": rollout_roofline(r, games, steps, sanma, encode=encode, acting=acting)}
    out["kernel_ms"], out["frac"] = out["roofline"]["kernel_ms"], out["roofline"]["frac"]
    if games <= 8192 and not encode:
        # what a search / MCTS caller feels at this batch size: ONE step of all games as its own launch, host call to host-visible completion
        for _ in range(20):
            env.step_random(policy_seed, 1, auto_reset=True)
        dev.sync()
        reps = 300
        ta = time.perf_counter()
        for _ in range(reps):
            env.step_random(policy_seed, 1, auto_reset=True)
            env.sync()
        tb = time.perf_counter()
        out["single_step_latency_us"] = (tb - ta) * 1e6 / reps
        out["single_step_latency_what"] = (f"rmj_step_random(n_steps=1) + rmj_sync, {games} games, mean of {reps} calls: launch, the step of every game "
                                           "(observations, masks and lists published), completion seen by the host")
    if encode:
        b_obs = B_OBS_3P if sanma else B_OBS_4P
        enc_ms = env.bench_encode(obs, 50, extended=False, only_active=2)
        out["roofline_encode"] = {"bound": "hbm", "achieved": b_obs * acting / (enc_ms * 1e-3) / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                                  "frac": b_obs * acting / (enc_ms * 1e-3) / HBM_PEAK, "kernel": "k_encode_base", "kernel_ms": enc_ms,
                                  "bytes_per_launch": b_obs * acting, "acting_seats": acting, "bytes_per_observation": b_obs}
    env.close()
    dev.free_all()
    return out


def external_policy_steps(envs, id_bufs, seed, k):
    """k iterations of what an external (neural) policy drives: per iteration and environment one sampler launch (rmj_sample_ids_device: the masked draw a
    policy's logits would feed; uniform here) and one step launch under its ids (rmj_step_ids_device = Observation.find_action + RiichiEnv.step, finished games
    restart).  Everything is asynchronous on each environment's own stream; with two environments - the two HALVES of a batch - the step of one overlaps the
    sampler / the launch ramp and tail of the other, which is how a trainer that alternates two groups hides them (riichienv-ml's actors: _ppo_worker.py:38)."""
    import ctypes as C

    for it in range(k):
        for e, buf in zip(envs, id_bufs):
            rc = e.L.rmj_sample_ids_device(e.h, None, 0, (seed + it) & 0xFFFFFFFFFFFFFFFF, buf)
            if rc == 0:
                rc = e.L.rmj_step_ids_device(e.h, buf, 1)
            if rc:
                raise RuntimeError(f"external_policy_steps: rc {rc}")


def external_policy_leg(vecenv, abi, shard, dev, games, mode, steps, device, rank, preroll=PREROLL, seed=0x5EED):
    """The per-step path of `env.step(actions)` under an external policy (env.rs:857-872), twice: the whole batch on one stream, and the batch as two halves
    (environments of games / 2, game offsets 0 and games / 2: the same games) on two streams, alternating."""
    import ctypes as C

    out = {}
    for name, parts in (("one_stream", 1), ("alternating_halves", 2)):
        per = games // parts
        envs = [vecenv.VecRiichiEnv(per, game_mode=mode, seed=0, rule_bits=abi.RULE_TENHOU, device=device,
                                    game_offset=shard.shard_offset(rank, games) + i * per, event_ring=64) for i in range(parts)]
        e0 = envs[0]
        e0.L.rmj_sample_ids_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p]
        e0.L.rmj_step_ids_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        bufs = [C.c_void_p(dev.alloc(per * 4 * 4)) for _ in envs]
        for e in envs:
            e.reset()
            e.step_random(0xC0FFEE, preroll, auto_reset=True)
        external_policy_steps(envs, bufs, seed, 20)
        for e in envs:
            e.sync()
        s0 = sum(e.total_steps() for e in envs)
        dev.sync()
        t0 = time.perf_counter()
        external_policy_steps(envs, bufs, seed + 1000, steps)
        for e in envs:
            e.sync()
        dev.sync()
        t1 = time.perf_counter()
        made = sum(e.total_steps() for e in envs) - s0
        out[name] = {"value": made / (t1 - t0), "unit": "env.step/s", "ms_per_step": (t1 - t0) * 1e3 / steps, "steps": steps, "streams": parts,
                     "launches_per_step": 2 * parts}
        for e in envs:
            e.close()
        dev.free_all()
    out["what"] = ("rmj_sample_ids_device (uniform) + rmj_step_ids_device per iteration - the launches an external policy causes; one_stream: the whole batch, "
                   "alternating_halves: two environments of half the games on two streams, issued alternately (the step of one half overlaps the sampler and the "
                   "ramp / tail of the other)")
    return out


def log_leg(vecenv, abi, shard, games, mode, policy_seed, device, rank, rounds=3, chunk=100, ring=512):
    """Lossless logs end to end: `rounds` x (a `chunk`-step auto-reset rollout + a drain of every slot's records, formatted to MJAI
    text on the host's threads) - the ring holds a chunk, the drains' cursors survive the restarts in between (stream positions), and
    the loss counters must stay 0."""
    import numpy as np

    env = vecenv.VecRiichiEnv(games, game_mode=mode, seed=0, rule_bits=abi.RULE_TENHOU, device=device,
                              game_offset=shard.shard_offset(rank, games), event_ring=ring)
    env.reset()
    env._log_cursor()
    env.step_random(policy_seed, chunk, auto_reset=True)      # untimed first round: sizes the text buffer, warms the pinned staging
    buf0, toffs0 = env.drain_logs(raw=True)
    text = np.empty(int(toffs0[-1]) * 2 + (1 << 20), np.uint8)
    del buf0
    s0 = env.total_steps()
    ev = tb = 0
    gather = copy = fmt = roll_s = drain_s = 0.0
    t0 = time.perf_counter()
    for _ in range(rounds):
        ta = time.perf_counter()
        env.step_random(policy_seed, chunk, auto_reset=True)
        env.sync()
        tb_ = time.perf_counter()
        tms = []
        _, toffs = env.drain_logs(raw=True, timings=tms, out=text)
        tc = time.perf_counter()
        roll_s += tb_ - ta
        drain_s += tc - tb_
        ev += int(env.last_drain_events)
        tb += int(toffs[-1])
        gather += tms[0]; copy += tms[1]; fmt += tms[2]
    wall = time.perf_counter() - t0
    made = env.total_steps() - s0
    lost = int(env.events_lost().sum())
    env.close()
    if lost != 0:
        raise RuntimeError(f"bench.py log_drain: {lost} records lost although the ring ({ring}) holds a {chunk}-step chunk")
    return {"games": games, "rollout_steps": rounds * chunk, "drain_every_steps": chunk, "event_ring": ring, "events": ev, "text_bytes": tb,
            "lost_events": lost, "wall_s": wall, "rollout_s": roll_s, "drain_s": drain_s, "gather_ms": gather, "copy_ms": copy, "format_ms": fmt,
            "events_per_s": ev / max(drain_s, 1e-9), "format_events_per_s": ev / max(fmt * 1e-3, 1e-9),
            "end_to_end_env_steps_per_s": made / max(wall, 1e-9),
            "what": f"{rounds} x ({chunk}-step auto-reset rollout + rmj_drain_format of every slot: device gather, one pinned copy, C formatter "
                    "on the host's threads into a reused text buffer); every record logged in the region is retrieved and formatted, "
                    "restarts included; end_to_end = env.steps of the region / (rollout + drain wall time)"}


def in_process_leg(args):
    """`--in-process K`: the batch sharded inside ONE process (riichienv_amd.multi_gpu.MultiGpuVecEnv: a handle and a host thread per shard, no
    collective) - W untimed warm-up steps, then exactly `--steps` steps of the device RandomAgent on every shard at once between two
    synchronisations of all shards.  Weak scaling like the ranks of --gpus: every shard owns --games games, global game indices key seeds and policy."""
    from riichienv_amd import abi, vecenv
    from riichienv_amd.multi_gpu import MultiGpuVecEnv

    k = args.in_process
    have = vecenv.load_lib().rmj_device_count()
    if have < 1:
        print("bench.py: no GPU visible (the product path has no CPU fallback)", file=sys.stderr)
        return 2
    devices = [i % have for i in range(k)]
    policy_seed = 0xC0FFEE
    env = MultiGpuVecEnv(args.games * k, devices=devices, game_mode=args.mode, seed=0, rule_bits=abi.RULE_TENHOU, event_ring=64)
    env.reset()
    if args.preroll > 0:
        env.step_random(policy_seed, args.preroll, auto_reset=True)
    env.step_random(policy_seed, max(args.warmup, 1), auto_reset=True)
    env._map(lambda i, e: e.sync())
    s0 = env._map(lambda i, e: e.total_steps())
    walls = [0.0] * k

    def timed(i, e):
        t0 = time.perf_counter()
        e.step_random(policy_seed, args.steps, auto_reset=True)
        e.sync()
        walls[i] = time.perf_counter() - t0
    t0 = time.perf_counter()
    env._map(timed)
    wall = time.perf_counter() - t0
    made = [b - a for a, b in zip(s0, env._map(lambda i, e: e.total_steps()))]
    env.close()
    line = {"metric": metric_name(args), "value": sum(made) / wall, "unit": "env.step/s", "n_gpus": len(set(devices)), "shards": k, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": wall * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
            "data": "synthetic", "host": "in-process: MultiGpuVecEnv, one host thread and one C-ABI handle per shard, no collective",
            "devices": devices, "per_shard_value": [m / max(w, 1e-12) for m, w in zip(made, walls)], "per_shard_wall_s": walls,
            "config": {"workload": f"{k} x {args.games} parallel {MODES[args.mode]} games, device RandomAgent, auto-reset, MJAI logging on", "games_per_shard": args.games,
                       "parallelism": f"{k} in-process shards on {len(set(devices))} device(s)"},
            "note": "shards that share a device are time-sliced by it: on one GPU this line shows that the path runs, not a scaling figure"}
    print(json.dumps(line))
    return 0


def main(argv=None):
    args = parse_args(argv)
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.in_process > 0:
        return in_process_leg(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: start it as `python bench.py --gpus {args.gpus}` or under "
              f"torch.distributed.run with --nproc-per-node {args.gpus}", file=sys.stderr)
        return 2

    # stdout carries exactly ONE line, the JSON of rank 0: everything libraries print on file descriptor 1 meanwhile (RCCL
    # announces its library path there) goes to stderr; the descriptor comes back for the line itself
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    # The CPU baseline starts worker processes: it runs FIRST on rank 0 (at any world size: the other ranks wait for it at the
    # rendezvous), before this process imports torch or touches HIP (a process that has initialised the GPU must not fork +
    # exec on this pool); it is outside the timed region either way.
    cpu_line = None
    if rank == 0 and not args.no_cpu_baseline:
        cpu_line = cpu_baseline(args.mode, 64 | 128, 0xC0FFEE)   # abi.RULE_TENHOU (the GPU run's rule set), the GPU run's policy seed

    from riichienv_amd import abi, shard, vecenv

    lib = vecenv.load_lib()
    use_dist = world > 1 or bool(os.environ.get("RMJ_BENCH_FORCE_DIST"))   # (the variable: the RCCL path with one rank on a 1-GPU box)
    torch = None
    if use_dist:
        import torch   # the ranks meet over torch.distributed (RCCL); a one-GPU run has no use for torch
    have = lib.rmj_device_count()
    shared_gpus = args.oversubscribe and 0 < have < world
    if shared_gpus:
        local_rank = local_rank % have
    if have < world and not shared_gpus:
        print(f"bench.py: {world} ranks need {world} GPUs on this node, {have} visible (the product path has no CPU fallback)",
              file=sys.stderr)
        return 3
    dist = None
    if use_dist:
        import torch.distributed as dist

        torch.cuda.set_device(local_rank)
        if shared_gpus:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    dev = Dev(lib, local_rank)

    policy_seed = 0xC0FFEE
    sanma = args.mode >= 3
    rule = abi.RULE_TENHOU | (abi.RULE_REFERENCE_RNG if args.reference_rng else 0)
    env = vecenv.VecRiichiEnv(args.games, game_mode=args.mode, seed=0, rule_bits=rule, device=local_rank,
                              game_offset=shard.shard_offset(rank, args.games), event_ring=64)
    env.reset()
    greedy = args.policy == "greedy"
    if greedy and args.encode:
        print("bench.py: --policy greedy has no --encode leg", file=sys.stderr)
        return 2

    def roll(k):
        if greedy:
            env.step_greedy(policy_seed, k, auto_reset=True, call_rate_256=args.call_rate)
        else:
            env.step_random(policy_seed, k, auto_reset=True)

    if args.preroll > 0:
        roll(args.preroll)   # to steady state (see PREROLL), untimed
    if args.encode:
        roll(args.warmup)
    elif args.warmup > 0:
        # the W untimed warm-up steps go through the entry point of the timed region (the same rollout, HIP events around it): its first call resolves the
        # event functions of the HIP runtime and builds the ctypes call frame - ~40 us of host time that a 20-step window (0.78 ms) would otherwise carry
        # (scripts/r06_window_order.py: first window 1.66-1.69 G, with this warm-up 1.74 G like every later one)
        if greedy:
            env.time_rollout_greedy(policy_seed, args.warmup, args.call_rate)
        else:
            env.time_rollout(policy_seed, args.warmup)

    def barrier():
        if dist is not None:
            dist.barrier()
        if torch is not None:
            torch.cuda.synchronize()
        dev.sync()           # hipDeviceSynchronize through the library (all a one-GPU run needs)

    obs = None
    if args.encode:
        # --padded-rows: every row padded to a multiple of 256 B (rmj_set_encode_row_stride; [games][4][2 048] floats in 3P, the first
        # 74 x 27 of a row are the tensor); the default is the dense [games][4][74][W] tensor
        stride = env.padded_row_stride() if args.padded_rows else 74 * (27 if sanma else 34)
        env.set_encode_row_stride(stride)
        obs = dev.alloc(args.games * 4 * stride * 4)
    full0 = env.total_full_path()
    before = env.total_steps()
    barrier()
    t0 = time.perf_counter()
    if args.encode:
        # every step of every game is followed by encode() of its acting seats into the resident tensor: ONE launch in which
        # every wave steps its four games and writes their rows (k_step4_enc / k_step4_queue_enc; RMJ_ENC_FUSED=0: four parts on
        # four streams, step + encode launches per part); HIP events on the handle's stream around it
        r = env.time_rollout_encode(policy_seed, args.steps, obs)
    else:
        # exactly K steps of every game and nothing else inside the region: HIP events on the handle's stream around the rollout
        # (returns when the second event has completed); the step counters are read outside
        r = env.time_rollout_greedy(policy_seed, args.steps, args.call_rate) if greedy else env.time_rollout(policy_seed, args.steps)
    barrier()
    t1 = time.perf_counter()
    steps_local = float(env.total_steps() - before)
    full_steps = env.total_full_path() - full0
    gm = shard.gather_measurement(dist, rank, t1 - t0, steps_local, device="cpu" if (shared_gpus or torch is None) else "cuda")
    wall, steps_total = gm["wall"], gm["steps"]

    # ---- side measurements, outside the timed region (rank 0 of a 1-GPU run only)
    extras = {}
    r_enc_step = None
    if args.encode:
        # k_step / k_encode launch durations for the roofline objects: single launches over all games (a policy that needs
        # the features is a barrier between steps, so this mode cannot keep several launches in flight)
        rs = [env.bench_rollout(policy_seed, 0, 1) for _ in range(20)]
        r_enc_step = sum(x.step_kernel_ms for x in rs) / len(rs)
        extras["encode"] = (acting_seats(env), env.bench_encode(obs, 50, extended=False, only_active=2))
    side = rank == 0 and world == 1 and not args.no_extras and not args.encode and not greedy
    if side:
        # the policy that plays to win (the RandomAgent wins once in ~250 rounds, a trainer's policy does not):
        # a second environment, pre-rolled under that policy, 300 timed steps
        genv = vecenv.VecRiichiEnv(args.games, game_mode=args.mode, seed=0, rule_bits=abi.RULE_TENHOU, device=local_rank,
                                   game_offset=shard.shard_offset(rank, args.games), event_ring=64)
        genv.reset()
        genv.step_greedy(policy_seed, 1500, auto_reset=True, call_rate_256=args.call_rate)
        gs0, gf0 = genv.total_steps(), genv.total_full_path()
        gr = genv.time_rollout_greedy(policy_seed, 300, args.call_rate)
        gs1, gf1 = genv.total_steps(), genv.total_full_path()
        extras["greedy_policy"] = {"value": (gs1 - gs0) / (gr.total_ms * 1e-3), "ms_per_step": gr.total_ms / 300, "steps": 300,
                                   "full_path_frac": (gf1 - gf0) / max(1, gs1 - gs0), "call_rate_256": args.call_rate,
                                   "kernel": fused_kernel_name(gr) + " (greedy policy instantiation)",
                                   "what": "the same rollout under rmj_step_greedy: shanten-greedy discards, every win / riichi / kan / kita "
                                           "taken, pon / chi at the call rate - rounds end with wins, not exhaustive draws"}
        genv.close()
        k = max(min(args.steps, 300), 20)
        env.set_rollout_streams(1)
        r1 = env.bench_rollout(policy_seed, 0, k)
        extras["single_stream"] = {"value": r1.env_steps / (r1.total_ms * 1e-3), "ms_per_step": r1.total_ms / k, "steps": k,
                                   "what": "the same rollout with every step as its own launch over all games on one stream "
                                           "(what a policy that is a barrier between steps gets)"}
        rv = env.bench_rollout_validated(policy_seed, 0, k)
        extras["validated_actions"] = {"value": rv.env_steps / (rv.total_ms * 1e-3), "ms_per_step": rv.total_ms / k, "steps": k,
                                       "what": "one policy launch writing packed actions + one step launch that validates them "
                                               "against the stored legal lists (state/mod.rs:339-402), one stream"}
        env.set_rollout_streams(4)
        if not args.no_configs:
            extras["external_policy"] = external_policy_leg(vecenv, abi, shard, dev, args.games, args.mode, max(min(args.steps, 300), 100), local_rank, rank)
        extras["log_drain"] = log_leg(vecenv, abi, shard, args.games, args.mode, policy_seed, local_rank, rank)
        # the steady-state figure of the SAME environment: >= 300 steps as one launch between device-wide synchronisations (a short
        # timed region pays the launch's ramp-up and tail, ~0.1 ms, once per K steps)
        kl = max(1000, args.steps)
        ls0, lf0 = env.total_steps(), env.total_full_path()
        dev.sync()
        tl0 = time.perf_counter()
        rl = env.time_rollout(policy_seed, kl)
        dev.sync()
        tl1 = time.perf_counter()
        lmade = env.total_steps() - ls0
        extras["long_rollout"] = {"value": lmade / (tl1 - tl0), "ms_per_step": (tl1 - tl0) * 1e3 / kl, "steps": kl, "steady_state": True,
                                  "kernel": fused_kernel_name(rl), "full_path_frac": (env.total_full_path() - lf0) / max(lmade, 1),
                                  "roofline": rollout_roofline(rl, args.games, kl, sanma),
                                  "what": f"the same rollout over {kl} steps as one launch, wall clock between device synchronisations"}
        # the same workload with every wall dealt through the REFERENCE's seed -> wall chain (RMJ_RULE_REFERENCE_RNG: what compat.RiichiEnv(seed=...)
        # and VecRiichiEnv(seeds=...) use by default; a base-seeded throughput run like this one keeps the build's own shuffle): a fresh
        # environment, pre-rolled, 1 000 timed steps
        if not args.reference_rng and not args.no_configs:
            leg = config_leg(vecenv, abi, shard, dev, "reference_rng", args.games, args.mode, 1000, policy_seed, local_rank, rank,
                             rule_extra=abi.RULE_REFERENCE_RNG)
            leg["what"] = ("the headline workload with RMJ_RULE_REFERENCE_RNG: every round start deals through StdRng::seed_from_u64 (PCG32), ChaCha12, "
                           "rand's chunked Fisher-Yates, salt and (on demand) the SHA-256 digest, state/wall.rs:36-67; compare with long_rollout")
            extras["reference_rng"] = leg
        # every other single-GPU configuration of BASELINE.json, each a steady-state leg of its own (>= 300 timed steps)
        if args.games == 65536 and args.mode == 2 and not args.no_configs:
            extras["configs"] = [
                config_leg(vecenv, abi, shard, dev, "configs[1]", 4096, 0, 2000, policy_seed, local_rank, rank),
                config_leg(vecenv, abi, shard, dev, "configs[3] (one shard of the eight: 524 288 games on this GPU)", 524288, 2, 300,
                           policy_seed, local_rank, rank),
                config_leg(vecenv, abi, shard, dev, "configs[4]", 65536, 5, 300, policy_seed, local_rank, rank, encode=True)]

    if rank == 0:
        acting_now = extras["encode"][0] if "encode" in extras else 0
        roof = rollout_roofline(r, args.games, args.steps, sanma, encode=args.encode, acting=acting_now, greedy=greedy, enc_step_ms=r_enc_step)
        # (the committed counter summary is of the fused rollout kernel, per step of all games; the per-step launches of the
        #  feature rollout have no counter profile of their own)
        traffic, traffic_src = pmc_traffic("k_step4_enc" if args.encode else "k_step4", roof["games_per_launch"], args.mode, ran_as=roof["kernel"])
        if traffic is not None:
            traffic *= roof["steps_per_launch"]
        roof.update({"traffic": traffic, "traffic_unit": "bytes/launch", "traffic_source": traffic_src})
        out = {
            "metric": metric_name(args),
            "value": steps_total / wall, "unit": "env.step/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": wall * 1e3 / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": workload_name(args), "games_per_gpu": args.games,
                       "sharding": "by game index, no collectives", "feature_tensor_output": bool(args.encode),
                       "parity": "bit-exact vs the oracle on identical walls; seed -> wall: " +
                                 ("the reference's StdRng / shuffle / salt chain (RMJ_RULE_REFERENCE_RNG)" if args.reference_rng else
                                  "the build's own shuffle (base-seeded throughput run; explicit per-game seeds and the drop-in shim deal the reference's chain, measured in the `reference_rng` leg: DESIGN.md §6)")},
            # proof that N processes took part: the rank ids the all_gather of the measurement returned, and each rank's own rate
            "ranks_seen": gm["ranks_seen"], "per_rank_value": gm["per_rank_value"], "per_rank_wall_s": gm["per_rank_wall_s"],
            "host_runtime": "torch.distributed (RCCL)" if dist is not None and not shared_gpus else ("torch.distributed (gloo)" if dist is not None else "none (C-ABI only)"),
            **({"oversubscribed": True, "note": "ranks share GPUs (--oversubscribe): a functional run of the world > 1 path, not a measurement"} if shared_gpus else {}),
            # the metric is defined over a window of >= 200 batched steps behind a warm-up that has reached round ends.  A shorter
            # timed window (the driver's --steps 20) of games that ARE in that state is marked separately: its rate includes the launch's
            # ramp-up and tail once per window, the steady-state figure of the same run is `long_rollout`
            "steady_state": bool(args.preroll + args.warmup >= 300 and args.steps >= STEADY_MIN),
            "window_ok": bool(args.preroll + args.warmup >= 300),
            "preroll_steps": args.preroll,
            "full_path_frac": full_steps / max(steps_local, 1.0),
            "roofline": roof,
        }
        if "encode" in extras:
            acting, enc_ms = extras.pop("encode")
            b_obs = B_OBS_3P if sanma else B_OBS_4P
            tr, tr_src = pmc_traffic("k_encode_3p" if sanma else "k_encode_4p", args.games, args.mode)
            out["roofline_encode"] = {"bound": "hbm", "achieved": b_obs * acting / (enc_ms * 1e-3) / 1e9, "peak": HBM_PEAK / 1e9,
                                      "unit": "GB/s", "frac": b_obs * acting / (enc_ms * 1e-3) / HBM_PEAK, "traffic": tr,
                                      "traffic_unit": "bytes/launch", "traffic_source": tr_src, "kernel": "k_encode_base",
                                      "kernel_ms": enc_ms, "bytes_per_launch": b_obs * acting, "acting_seats": acting,
                                      "bytes_per_observation": b_obs}
        out.update(extras)
        if cpu_line is not None:
            out["cpu_baseline"] = cpu_line
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    env.close()
    dev.free_all()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
