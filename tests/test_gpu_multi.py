"""GPU: the in-process multi-GPU host (riichienv_amd.multi_gpu.MultiGpuVecEnv: one handle and one host thread per device) with two
shards on ONE device against a single handle, and bench.py's world > 1 path with two ranks sharing the GPU (--oversubscribe: gloo,
RCCL refuses two ranks on one device).  No scaling curve has been measured: a multi-GPU node was never available to the build."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from riichienv_amd import vecenv
from riichienv_amd.multi_gpu import MultiGpuVecEnv

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("mode", [2, 5])
def test_two_shards_on_two_threads_equal_one_handle(mode):
    n, seed, pseed = 512, 41 + mode, 99
    one = vecenv.VecRiichiEnv(n, game_mode=mode, seed=seed, event_ring=2048)
    two = MultiGpuVecEnv(n, devices=[0, 0], game_mode=mode, seed=seed, event_ring=2048)
    one.reset()
    two.reset()
    for k in (1, 150, 33):
        one.step_random(pseed, k, auto_reset=True)
        two.step_random(pseed, k, auto_reset=True)
    assert two.total_steps() == one.total_steps()
    for a, b in zip(two.status(), one.status()):
        assert (a == b).all()
    la, ca = two.legal()
    lb, cb = one.legal()
    assert (ca == cb).all()
    for g in range(n):
        for s in range(4):
            assert (la[g, s, : ca[g, s]] == lb[g, s, : cb[g, s]]).all()
    assert (two.mask() == one.mask()).all() and (two.scores() == one.scores()).all() and (two.step_counts() == one.step_counts()).all()
    # host-driven steps through the fan-out
    acts = one.random_actions(pseed)
    assert (two.random_actions(pseed) == acts).all()
    one.step(acts)
    two.step(acts)
    for a, b in zip(two.status(), one.status()):
        assert (a == b).all()
    logs = two.mjai_logs()
    for g in (0, 255, 256, n - 1):
        assert logs[g] == one.mjai_log(g) == two.mjai_log(g), g
    two.close()
    one.close()


def test_bench_with_two_ranks_sharing_the_gpu():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--games", "8192", "--steps", "60", "--warmup", "20",
           "--preroll", "400", "--no-extras", "--oversubscribe"]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, p.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["oversubscribed"] is True and line["scaling"] == "weak"
    # the line proves its ranks: the ids one all_gather returned, each rank's own rate, and rank 0's CPU baseline at world > 1
    assert line["ranks_seen"] == [0, 1] and len(line["per_rank_value"]) == 2 and all(v > 0 for v in line["per_rank_value"])
    assert line["value"] <= sum(line["per_rank_value"]) * 1.0001 and line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["value"] > 0
    # two ranks x 8 192 games x 60 steps, nearly every game advances every step: the SUM over ranks, the MAX of their times
    steps = line["value"] * line["ms_per_step"] * 1e-3 * 60
    assert 0.9 * 2 * 8192 * 60 <= steps <= 2 * 8192 * 60 * 1.0001, steps


def test_bench_in_process_shards():
    """bench.py --in-process 2: the in-process counterpart of --gpus 2 (MultiGpuVecEnv, a host thread and a handle per shard); on a one-GPU
    box both shards share the device."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--in-process", "2", "--games", "8192", "--steps", "60", "--warmup", "20", "--preroll", "400"]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, p.stdout
    line = json.loads(lines[0])
    assert line["shards"] == 2 and line["scaling"] == "weak" and line["n_gpus"] >= 1 and len(line["devices"]) == 2
    assert len(line["per_shard_value"]) == 2 and all(v > 0 for v in line["per_shard_value"]) and "in-process" in line["host"]
    steps = line["value"] * line["ms_per_step"] * 1e-3 * 60
    assert 0.9 * 2 * 8192 * 60 <= steps <= 2 * 8192 * 60 * 1.0001, steps


@pytest.mark.parametrize("mode", [2, 5])
def test_alternating_halves_driver_equals_the_whole_batch(mode):
    """bench.py's `external_policy` leg (round 6): the batch as two environments of half the games on two streams, each iteration one sampler
    launch + one step launch under its ids per half, issued alternately - must leave every game where the same loop over ONE environment of all
    games on one stream leaves it (the sampler's noise and the walls are keyed by the GLOBAL game index)."""
    import ctypes as C

    torch = pytest.importorskip("torch")
    sys.path.insert(0, ROOT)
    import bench

    n, k = 4096, 120
    whole = [vecenv.VecRiichiEnv(n, game_mode=mode, seed=3, event_ring=64)]
    halves = [vecenv.VecRiichiEnv(n // 2, game_mode=mode, seed=3, game_offset=i * (n // 2), event_ring=64) for i in range(2)]
    L = whole[0].L
    L.rmj_sample_ids_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p]
    L.rmj_step_ids_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    for e in whole + halves:
        e.reset()
    ids_w = [torch.full((n, 4), -1, dtype=torch.int32, device="cuda:0")]
    ids_h = [torch.full((n // 2, 4), -1, dtype=torch.int32, device="cuda:0") for _ in range(2)]
    torch.cuda.synchronize()
    bench.external_policy_steps(whole, [C.c_void_p(t.data_ptr()) for t in ids_w], 17, k)
    bench.external_policy_steps(halves, [C.c_void_p(t.data_ptr()) for t in ids_h], 17, k)
    for e in whole + halves:
        e.sync()
    assert (np.concatenate([h.step_counts() for h in halves]) == whole[0].step_counts()).all()
    assert (np.concatenate([h.scores() for h in halves]) == whole[0].scores()).all()
    for a, b in zip(zip(*[h.status() for h in halves]), whole[0].status()):
        assert (np.concatenate(a) == b).all()
    assert (np.concatenate([h.mask() for h in halves]) == whole[0].mask()).all()
    assert torch.equal(torch.cat(ids_h), ids_w[0]) and int(whole[0].total_steps()) > n * k // 2
