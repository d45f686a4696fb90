"""The N-rank leg of bench.py on a 1-GPU box: the exact launch the driver uses for its scaling runs
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...`)
with N = 1 and RMJ_BENCH_FORCE_DIST=1, so that RCCL's init_process_group, the barriers and the all_reduce of
shard.reduce_measurement really run on the GPU.  The launcher is started as a CHILD of pytest and this file touches no GPU
itself (it sorts in front of the in-process GPU tests): nothing is exec'ed over a process that has initialised HIP."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _one_json_line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_bench_under_torchrun_with_rccl_world_of_one():
    env = dict(os.environ, RMJ_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--games", "4096", "--steps", "40",
           "--warmup", "40", "--no-cpu-baseline", "--no-extras"]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    line = _one_json_line(p.stdout)
    assert line["n_gpus"] == 1 and line["steps"] == 40 and line["warmup"] == 40 and line["unit"] == "env.step/s"
    assert line["scaling"] == "weak" and line["value"] > 1e6 and line["window_ok"] is True and line["steady_state"] is False and line["preroll_steps"] >= 300
    # 4 096 games x 40 steps, nearly every game advances every step
    assert 0.9 * 4096 * 40 <= line["value"] * line["ms_per_step"] * 1e-3 * 40 <= 4096 * 40 * 1.0001
    assert line["roofline"]["frac"] > 0 and "cpu_baseline" not in line
    assert line["ranks_seen"] == [0] and len(line["per_rank_value"]) == 1 and line["host_runtime"].startswith("torch.distributed (RCCL)")


def test_bench_default_launch_with_the_drivers_flags():
    """`python bench.py --steps 20 --warmup 5` (the driver's end-of-round command at N = 1): a steady-state line of the ticket
    kernel - every game is pre-rolled through several round ends before the warm-up, and the 20 timed steps run as k_step4_queue."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--no-cpu-baseline"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    line = _one_json_line(p.stdout)
    with open(os.path.join(ROOT, "BASELINE.json")) as f:
        assert line["metric"] == json.load(f)["metric"]
    assert line["steps"] == 20 and line["warmup"] == 5 and line["steady_state"] is False and line["window_ok"] is True
    assert 0.0005 < line["full_path_frac"] < 0.02, line["full_path_frac"]          # the lean tier of the fused rollout still leaves for wait probes and kans; round ends no longer do
    r = line["roofline"]
    assert r["kernel"] == "k_step4_queue" and r["steps_per_launch"] == 20 and r["games_per_launch"] == 65536
    assert r["traffic_source"] is None or "k_step4" in r["traffic_source"]
    assert line["value"] > 5e8
    # a one-GPU run needs no torch: the harness reaches HIP through the library's own hooks
    assert line["host_runtime"] == "none (C-ABI only)" and line["ranks_seen"] == [0]
    # the steady-state leg of the same environment, the other single-GPU configurations of BASELINE.json, the lossless log leg
    lr = line["long_rollout"]
    assert lr["steady_state"] is True and lr["steps"] >= 300 and lr["value"] > line["value"] and 0 < lr["roofline"]["frac"] < 1
    cfgs = {c["config"].split(" ")[0]: c for c in line["configs"]}
    assert set(cfgs) == {"configs[1]", "configs[3]", "configs[4]"}
    for c in cfgs.values():
        assert c["steady_state"] is True and c["steps"] >= 300 and c["kernel_ms"] > 0 and 0 < c["frac"] < 1 and c["value"] > 1e8
        ro = c["roofline"]
        assert abs(ro["achieved"] - ro["launches_in_flight"] * ro["bytes_per_launch"] / (ro["kernel_ms"] * 1e-3) / 1e9) < 1e-6 * ro["achieved"]
    assert "4096 parallel 4p-red-single" in cfgs["configs[1]"]["workload"] and "524288 parallel 4p-red-half" in cfgs["configs[3]"]["workload"]
    assert "65536 parallel 3p-red-half" in cfgs["configs[4]"]["workload"] and cfgs["configs[4]"]["roofline_encode"]["frac"] > 0
    ld = line["log_drain"]
    assert ld["lost_events"] == 0 and ld["events"] > 1e7 and ld["end_to_end_env_steps_per_s"] > 0 and ld["format_events_per_s"] > 0
