"""The committed bench line (profiles/r*_bench_n1.json, written by bench.py on the GPU box) carries every field of the
bench contract: the metric of BASELINE.json, the roofline object and the CPU baseline; and `bench.py --gpus N` builds
the N-rank launch the contract describes (checked without a GPU: the launcher is a pure function)."""
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _newest_line():
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_n1.json")))
    with open(paths[-1]) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def test_committed_bench_line_has_the_contract_fields():
    line = _newest_line()
    with open(os.path.join(ROOT, "BASELINE.json")) as f:
        base = json.load(f)
    assert line["metric"] == base["metric"] and line["unit"] == "env.step/s" and line["higher_is_better"] is True
    for k in ("value", "n_gpus", "steps", "warmup", "ms_per_step", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in line, k
    assert line["n_gpus"] == 1 and line["scaling"] == "weak" and line["vs_baseline"] is None and "workload" in line["config"]
    r = line["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    # achieved = launches in flight x algorithmic bytes per launch / average launch duration
    assert abs(r["achieved"] - r["launches_in_flight"] * r["bytes_per_launch"] / (r["kernel_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    assert r["bytes_per_launch"] == 1688 * r["games_per_launch"] * r.get("steps_per_launch", 1)
    assert r["traffic"] is None or r["traffic"] > 0
    c = line["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "env.step/s" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    # round 5: the driver's own command (--steps 20 --warmup 5) carries every single-GPU configuration of BASELINE.json as a steady-state leg,
    # the steady-state figure of the timed environment, a lossless log leg, and the proof of its ranks
    assert line["steps"] == 20 and line["warmup"] == 5 and line["ranks_seen"] == [0] and len(line["per_rank_value"]) == 1
    assert line["host_runtime"] == "none (C-ABI only)"
    lr = line["long_rollout"]
    assert lr["steady_state"] is True and lr["steps"] >= 300 and abs(lr["roofline"]["frac"] - lr["roofline"]["achieved"] / 8000.0) < 1e-9
    names = [c["config"].split(" ")[0] for c in line["configs"]]
    assert names == ["configs[1]", "configs[3]", "configs[4]"]
    for c in line["configs"]:
        assert c["steady_state"] is True and c["steps"] >= 300 and c["kernel_ms"] > 0 and 0 < c["frac"] < 1 and c["unit"] == "env.step/s"
        assert abs(c["frac"] - c["roofline"]["achieved"] / c["roofline"]["peak"]) < 1e-9
    assert "roofline_encode" in line["configs"][2] and "3p" in line["configs"][2]["workload"]
    assert 0 < line["configs"][0]["single_step_latency_us"] < 1e4   # the small batch: what one step as its own launch costs a caller
    ld = line["log_drain"]
    assert ld["lost_events"] == 0 and ld["events"] > 0 and ld["end_to_end_env_steps_per_s"] > 0 and ld["format_events_per_s"] > 0


def test_gpus_flag_builds_the_n_rank_launch():
    import bench

    args = bench.parse_args(["--gpus", "8", "--steps", "300", "--warmup", "250", "--no-cpu-baseline"])
    cmd, env = bench.launcher_command(args, 29511, script="/x/bench.py")
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    tail = cmd[cmd.index("/x/bench.py") + 1:]
    # the ranks get the same workload flags, and --gpus so that every rank can check it against WORLD_SIZE
    assert tail[:10] == ["--gpus", "8", "--steps", "300", "--warmup", "250", "--games", "65536", "--mode", "2"]
    assert "--no-cpu-baseline" in tail and "--encode" not in tail
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and env["MASTER_ADDR"] == "127.0.0.1"
    # weak scaling: rank r owns the global games [r * games, (r + 1) * games)
    from riichienv_amd import shard

    assert [shard.shard_offset(r, args.games) for r in range(8)] == [r * 65536 for r in range(8)]


def test_gpus_flag_must_match_world_size():
    """Under a launcher the flag and WORLD_SIZE must agree; the check fires before torch or the GPU are touched."""
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env, capture_output=True, text=True)
    assert p.returncode == 2 and "WORLD_SIZE=2" in p.stderr and p.stdout.strip() == ""


def test_metric_names():
    import bench

    with open(os.path.join(ROOT, "BASELINE.json")) as f:
        base = json.load(f)
    assert bench.metric_name(bench.parse_args([])) == base["metric"]
    assert bench.metric_name(bench.parse_args(["--gpus", "8"])) == base["metric"]
    m5 = bench.metric_name(bench.parse_args(["--mode", "5", "--encode"]))
    assert "3p" in m5 and "feature-encoding" in m5 and "4p" not in m5


def test_in_process_flag_parses():
    import bench

    args = bench.parse_args(["--in-process", "8", "--games", "65536"])
    assert args.in_process == 8 and args.gpus == 1 and bench.parse_args([]).in_process == 0
