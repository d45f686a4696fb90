#!/usr/bin/env python3
"""Trainer-side loop throughput (row N4): encode observations + masked uniform policy in torch + id-based step, all on
one GPU.  Prints env.step/s of the whole loop and of its parts."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from riichienv_amd.torch_env import TorchVecEnv  # noqa: E402


def run(n, ext, shared, fused=False):
    env = TorchVecEnv(n, game_mode=2, seed=0, extended=ext, share_stream=shared)
    gen = torch.Generator(device=env.device)
    gen.manual_seed(0)
    it = [0]

    def policy():
        it[0] += 1
        return env.sample_ids(seed=it[0]) if fused else env.sample_random_ids(gen)

    for _ in range(20):
        env.step(policy())
    torch.cuda.synchronize()
    t_obs = t_pol = t_step = 0.0
    steps0 = env.env.total_steps()
    K = 100
    t0 = time.perf_counter()
    for _ in range(K):
        a = time.perf_counter()
        env.obs(only_active=True)
        b = time.perf_counter()
        ids = policy()
        if not shared:
            torch.cuda.synchronize()      # own stream: the parts are timed one by one
        c = time.perf_counter()
        env.step(ids)
        d = time.perf_counter()
        t_obs += b - a
        t_pol += c - b
        t_step += d - c
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    steps = env.env.total_steps() - steps0
    parts = "" if shared else f" (obs {t_obs / K * 1e3:.2f} ms, torch policy {t_pol / K * 1e3:.2f} ms, step {t_step / K * 1e3:.2f} ms per iteration)"
    print(f"games {n} channels {env.channels} {'fused masked sampler (rmj_sample_ids_device)' if fused else 'torch masked multinomial'}, "
          f"{'shared stream, no host sync' if shared else 'own stream, synchronised parts'}: "
          f"loop {steps / (t1 - t0) / 1e6:.1f} M env.step/s, {(t1 - t0) / K * 1e3:.2f} ms per iteration{parts}")


def run_one_launch(n, pad=False):
    """the loop with step + encode as ONE launch per iteration (TorchVecEnv.step_obs -> rmj_step_ids_encode_device) and the fused sampler;
    pad: every (game, seat) row of the feature tensor padded to a multiple of 256 B (TorchVecEnv(pad_rows=True): a strided view)"""
    env = TorchVecEnv(n, game_mode=2, seed=0, share_stream=True, pad_rows=pad)
    env.obs(only_active=True)
    for k in range(20):
        env.step_obs(env.sample_ids(seed=k + 1))
    torch.cuda.synchronize()
    steps0 = env.env.total_steps()
    K = 100
    t0 = time.perf_counter()
    for k in range(K):
        env.step_obs(env.sample_ids(seed=100 + k))
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    steps = env.env.total_steps() - steps0
    print(f"games {n} channels 74 fused masked sampler + step and encode as one launch (rmj_step_ids_encode_device){', rows padded to 256 B' if pad else ''}, shared stream: "
          f"loop {steps / (t1 - t0) / 1e6:.1f} M env.step/s, {(t1 - t0) / K * 1e3:.2f} ms per iteration")


def run_sample_step_one_launch(n, pad=True):
    """the loop with the policy's draw, the step and the next observations as ONE launch per iteration (TorchVecEnv.step_sample_obs ->
    rmj_step_sample_encode_device): the whole environment side of an iteration between two policy forward passes"""
    env = TorchVecEnv(n, game_mode=2, seed=0, share_stream=True, pad_rows=pad)
    env.obs(only_active=True)
    for k in range(20):
        env.step_sample_obs(None, seed=k + 1)
    torch.cuda.synchronize()
    steps0 = env.env.total_steps()
    K = 100
    t0 = time.perf_counter()
    for k in range(K):
        env.step_sample_obs(None, seed=100 + k)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    steps = env.env.total_steps() - steps0
    print(f"games {n} channels 74 sampler + step + encode as ONE launch (rmj_step_sample_encode_device){', rows padded to 256 B' if pad else ''}, shared stream: "
          f"loop {steps / (t1 - t0) / 1e6:.1f} M env.step/s, {(t1 - t0) / K * 1e3:.2f} ms per iteration")


def run_parts_one_launch(n, parts):
    """shards on streams, step + encode as one launch per shard and iteration"""
    from riichienv_amd.torch_env import ShardedTorchVecEnv

    env = ShardedTorchVecEnv(n, parts=parts, game_mode=2, seed=0)
    it = 0

    def one_round():
        nonlocal it
        it += 1
        env.step_policy_one_launch(lambda e, obs: e.sample_ids(seed=it))

    for _ in range(20):
        one_round()
    torch.cuda.synchronize()
    env.synchronize()
    steps0 = sum(e.env.total_steps() for e in env.shards)
    K = 100
    t0 = time.perf_counter()
    for _ in range(K):
        one_round()
    torch.cuda.synchronize()
    env.synchronize()
    t1 = time.perf_counter()
    steps = sum(e.env.total_steps() for e in env.shards) - steps0
    print(f"games {n} as {parts} shards on {parts} streams, step and encode as one launch per shard, fused masked sampler: "
          f"loop {steps / (t1 - t0) / 1e6:.1f} M env.step/s, {(t1 - t0) / K * 1e3:.2f} ms per iteration")


def run_parts(n, parts, compact=False):
    """The same loop on ShardedTorchVecEnv: the batch as `parts` shards (by game index) on `parts` torch streams - the store-bound
    encoder of one shard runs under the issue-bound step of another (what rmj_step_random_encode does inside the library)."""
    from riichienv_amd.torch_env import ShardedTorchVecEnv

    env = ShardedTorchVecEnv(n, parts=parts, game_mode=2, seed=0)
    it = 0

    def one_round():
        nonlocal it
        it += 1
        env.step_policy(lambda e, *obs: e.sample_ids(seed=it), compact=compact)

    for _ in range(20):
        one_round()
    torch.cuda.synchronize()
    env.synchronize()
    steps0 = sum(e.env.total_steps() for e in env.shards)
    K = 100
    t0 = time.perf_counter()
    for _ in range(K):
        one_round()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    steps = sum(e.env.total_steps() for e in env.shards) - steps0
    print(f"games {n} as {parts} shards on {parts} streams, {'compact batch' if compact else '[n, 4] tensor'}, fused masked sampler: "
          f"loop {steps / (t1 - t0) / 1e6:.1f} M env.step/s, {(t1 - t0) / K * 1e3:.2f} ms per iteration")


def run_graph(n, per_graph=16, with_policy=True):
    """policy (one torch matmul over the observation rows) -> fused sampler -> step + encode, eager against the same iterations captured
    once as a HIP graph on a side stream the environment is bound to (TorchVecEnv.bind_stream) and replayed"""
    env = TorchVecEnv(n, game_mode=2, seed=0, share_stream=True)
    env.obs(only_active=True)
    w = (torch.randn(74 * 34, 82, device="cuda") * 0.05).contiguous()

    def iteration():
        logits = (env._obs.reshape(n * 4, 74 * 34) @ w).view(n, 4, 82) if with_policy else None
        env.step_obs(env.sample_ids(logits=logits, seed=7))

    def timed(fn, reps):
        torch.cuda.synchronize()
        s0 = env.env.total_steps()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        return (env.env.total_steps() - s0) / (t1 - t0), (t1 - t0) / reps

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        env.bind_stream()
        for _ in range(20):
            iteration()
        side.synchronize()
        eager, t_e = timed(iteration, 25 * per_graph)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for _ in range(per_graph):
                iteration()
        for _ in range(3):
            g.replay()
        graph, t_g = timed(g.replay, 25)
    what = "torch matmul policy + fused sampler + step and encode" if with_policy else "fused sampler + step and encode"
    print(f"games {n} {what}, one stream: eager {eager / 1e6:.1f} M env.step/s ({t_e * 1e6:.0f} us per iteration), "
          f"as a HIP graph of {per_graph} iterations {graph / 1e6:.1f} M env.step/s ({t_g / per_graph * 1e6:.0f} us per iteration)", flush=True)


class QNetShape(torch.nn.Module):
    """A policy of the reference's shape (riichienv-ml models/q_network.py:7-30 + backbone.py:47-70: Conv1d(74 -> 64, k 3) + 3 residual blocks of two
    Conv1d(64 -> 64, k 3) over the 34 tile columns, Linear(64 * 34 -> 256), Linear(256 -> 82)), random weights, bf16.  CALLER side, not product: it only
    exists so that the loop is measured with a network in it.  The k = 3 convolutions are written as GEMMs over three shifted copies of the columns
    (hipBLASLt / MFMA; no MIOpen kernel search on a fresh box); BatchNorm in eval mode is an affine map folded into the weights."""

    def __init__(self, width=34, c_in=74, c=64, blocks=3, fc=256, actions=82):
        super().__init__()
        mk = lambda i, o: torch.nn.Parameter(torch.randn(3 * i, o) * (1.0 / (3 * i)) ** 0.5)   # noqa: E731
        self.w_in = mk(c_in, c)
        self.w_res = torch.nn.ParameterList([mk(c, c) for _ in range(2 * blocks)])
        self.fc1 = torch.nn.Linear(c * width, fc)
        self.fc2 = torch.nn.Linear(fc, actions)
        self.width = width

    @staticmethod
    def conv3(x, w):          # x [B, W, C] -> [B, W, C'] : y[:, j] = concat(x[:, j - 1], x[:, j], x[:, j + 1]) @ w  (zero padded)
        z = torch.nn.functional.pad(x, (0, 0, 1, 1))
        return torch.cat([z[:, :-2], z[:, 1:-1], z[:, 2:]], dim=-1) @ w

    def forward(self, obs):   # obs [B, 74, W] float32 (strided rows are fine)
        x = obs.to(torch.bfloat16).transpose(1, 2)            # [B, W, 74]
        x = torch.relu(self.conv3(x, self.w_in))
        for i in range(0, len(self.w_res), 2):
            y = torch.relu(self.conv3(x, self.w_res[i]))
            x = torch.relu(x + self.conv3(y, self.w_res[i + 1]))
        x = torch.relu(self.fc1(x.transpose(1, 2).reshape(x.shape[0], -1)))
        return self.fc2(x).float()


def run_net(n, halves=False):
    """The trainer loop with a REAL network in it: compact observations of the acting seats -> QNetShape (bf16) -> masked sampler ->
    step + encode; one stream, or - halves - two shards on two streams issued alternately (the policy of one half next to the step of the other).
    Reports the loop's rate and how much of an iteration belongs to the environment (HIP events around the library's launches)."""
    from riichienv_amd.torch_env import ShardedTorchVecEnv

    parts = 2 if halves else 1
    env = ShardedTorchVecEnv(n, parts=parts, game_mode=2, seed=0)
    net = QNetShape().to(env.device).to(torch.bfloat16).eval()
    it = [0]
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(parts)]

    def one(e, i, timed=False):
        it[0] += 1
        obs, index, cnt = e.obs_compact(sync_count=False)
        k = min(obs.shape[0], e.n + e.n // 4)                                  # rows behind the device count are stale and ignored by the sampler
        if timed:
            ev[i][0].record()
        with torch.no_grad():
            logits = net(obs[:k])
        if timed:
            ev[i][1].record()
        e.step(e.sample_ids(logits=logits, seed=it[0], index=index[:k], count=cnt))
        if timed:
            ev[i][2].record()

    for _ in range(6):
        env.for_each(lambda e, i: one(e, i))
    torch.cuda.synchronize()
    env.synchronize()
    steps0 = sum(e.env.total_steps() for e in env.shards)
    K = 20
    t0 = time.perf_counter()
    for r in range(K):
        env.for_each(lambda e, i: one(e, i, timed=(r == K - 1)))
    torch.cuda.synchronize()
    env.synchronize()
    t1 = time.perf_counter()
    steps = sum(e.env.total_steps() for e in env.shards) - steps0
    pol = sum(ev[i][0].elapsed_time(ev[i][1]) for i in range(parts)) / parts
    envms = sum(ev[i][1].elapsed_time(ev[i][2]) for i in range(parts)) / parts
    print(f"games {n} {'as two halves on two streams' if halves else 'on one stream'}, policy = QNetShape (74 x 34 -> 3 residual blocks of 64 channels -> 256 -> 82, bf16 GEMMs) on the "
          f"compact batch of the acting seats + fused masked sampler + step: loop {steps / (t1 - t0) / 1e6:.2f} M env.step/s, {(t1 - t0) / K * 1e3:.2f} ms per iteration; "
          f"of one {'half-' if halves else ''}iteration the network takes {pol:.2f} ms, sampler + step {envms:.2f} ms (HIP events)", flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "net":
        n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
        run_net(n)
        if len(sys.argv) > 3 and sys.argv[3] == "halves":      # (two shards on two streams: with a network ~80 x the environment's time there is nothing for it to hide;
            run_net(n, halves=True)                              #  the first attempt on the GPU box did not return within 10 minutes - not investigated, run it under `timeout`)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "graph":
        for n in (1024, 4096, 16384, 65536):
            run_graph(n, with_policy=False)
            run_graph(n, with_policy=True)
        return
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    ext = len(sys.argv) > 2 and sys.argv[2] == "ext"
    run(n, ext, False)
    run(n, ext, True)
    run(n, ext, False, fused=True)
    run(n, ext, True, fused=True)
    if not ext:
        run_one_launch(n)
        run_one_launch(n, pad=True)
        run_sample_step_one_launch(n)
        for parts in (2, 4):
            run_parts(n, parts)
        run_parts(n, 4, compact=True)
        for parts in (2, 4):
            run_parts_one_launch(n, parts)


if __name__ == "__main__":
    main()
