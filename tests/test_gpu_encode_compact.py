"""rmj_encode_compact_device: the Observation.encode() tensors of the acting seats as one dense batch in (game, seat) order
(what a trainer stacks from the reference's `{pid: obs.encode()}` of env.step, env.rs:857-872).  Every row must be byte-equal
to the acting seat's row of the [n_games][4] tensor of rmj_encode_device and - on sampled games - to the oracle's encode();
the index must list exactly the acting seats of the unfinished games in order."""
import ctypes as C

import numpy as np
import pytest

from riichienv_amd.shard import game_seed

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode,n", [(2, 3001), (5, 4099), (0, 257)])
def test_compact_rows_equal_the_full_tensor_and_the_oracle(mode, n):
    import torch

    from oracle import oracle
    from riichienv_amd import vecenv

    w = 27 if mode >= 3 else 34
    seed, pseed = 77, 0xBEEF
    env = vecenv.VecRiichiEnv(n, game_mode=mode, seed=seed, event_ring=64)
    env.reset()
    games = [oracle.Game(game_mode=mode, seed=game_seed(seed, g)) for g in range(8)]
    for o in games:
        o.reset()
    full = torch.zeros((n, 4, 74, w), dtype=torch.float32, device="cuda:0")
    cap = n + n // 2
    out = torch.full((cap, 74, w), -7.0, dtype=torch.float32, device="cuda:0")
    index = torch.full((cap,), -1, dtype=torch.int32, device="cuda:0")
    count = torch.zeros((1,), dtype=torch.int32, device="cuda:0")
    multi = 0
    for step in range(160):      # no auto-reset: finished games must drop out of the batch (mode 0 games end inside the loop)
        if step % 3 == 0:
            full.zero_()
            out.fill_(-7.0)
            torch.cuda.synchronize()      # torch's stream is not the handle's stream
            vecenv._chk(env.L.rmj_encode_device(env.h, 2, C.c_void_p(full.data_ptr())))
            env.encode_compact_device(out.data_ptr(), index.data_ptr(), cap, count.data_ptr())
            env.L.rmj_sync(env.h)
            act, _, done = env.status()
            want = [g * 4 + s for g in range(n) if not done[g] for s in range(4) if (int(act[g]) >> s) & 1]
            k = int(count.item())
            assert k == len(want) and k <= cap, (step, k, len(want))
            idx = index[:k].cpu().numpy()
            assert idx.tolist() == want
            rows = out[:k]
            ref = full.view(n * 4, 74, w)[torch.from_numpy(idx.astype(np.int64)).to("cuda:0")]
            bad = (rows != ref).flatten(1).any(dim=1).nonzero().flatten().tolist()
            assert not bad, (step, len(bad), bad[:5], (rows[bad[0]] != ref[bad[0]]).nonzero()[:6].tolist(), idx[bad[0]])
            assert float(out[k:].max().item()) == -7.0 and float(out[k:].min().item()) == -7.0     # nothing behind the batch is touched
            multi += int((np.bincount(idx >> 2, minlength=n) > 1).sum())
            host = rows.cpu().numpy()
            for j, gi in enumerate(idx):
                g, s = int(gi) >> 2, int(gi) & 3
                if g < len(games):
                    assert host[j].tobytes() == games[g].encode(s, mode >= 3).tobytes(), (step, g, s)
        for g, o in enumerate(games):
            if not o.status()[2]:
                o.step(o.random_actions(pseed, g))
        env.step_random(pseed, 1, auto_reset=False)
    assert multi > 0 or mode != 2, multi   # 4P half games: states with several claimants (pon + chi of one discard) were part of it


def test_capacity_smaller_than_the_batch_writes_only_capacity_rows():
    import torch

    from riichienv_amd import vecenv

    n = 512
    env = vecenv.VecRiichiEnv(n, game_mode=2, seed=5, event_ring=64)
    env.reset()
    env.step_random(1, 30, auto_reset=True)
    cap = 100
    out = torch.full((cap + 8, 74, 34), -7.0, dtype=torch.float32, device="cuda:0")
    index = torch.full((cap + 8,), -1, dtype=torch.int32, device="cuda:0")
    count = torch.zeros((1,), dtype=torch.int32, device="cuda:0")
    torch.cuda.synchronize()
    env.encode_compact_device(out.data_ptr(), index.data_ptr(), cap, count.data_ptr())
    env.L.rmj_sync(env.h)
    assert int(count.item()) >= n          # the count reports the whole batch
    assert (index[:cap] >= 0).all() and (index[cap:] == -1).all()
    assert float(out[cap:].max().item()) == -7.0 and float(out[:cap].min().item()) >= 0.0


@pytest.mark.parametrize("mode", [2, 5])
def test_padded_row_stride_equals_dense_rows(mode):
    """rmj_set_encode_row_stride: with every (game, seat) row padded to a multiple of 256 B the base encoder - per-step launch, dense
    compact batch, the one-launch step + encode rollout, the one-launch step under action ids - writes the same 74 x W floats at the
    start of every row and never touches the pad."""
    import torch

    from riichienv_amd import vecenv
    from riichienv_amd.torch_env import TorchVecEnv

    n, w = 4099, (27 if mode >= 3 else 34)
    a = vecenv.VecRiichiEnv(n, game_mode=mode, seed=91, event_ring=64)
    b = vecenv.VecRiichiEnv(n, game_mode=mode, seed=91, event_ring=64)
    stride = b.padded_row_stride()
    assert stride % 64 == 0 and 74 * w <= stride < 74 * w + 64
    b.set_encode_row_stride(stride)
    for e in (a, b):
        e.reset()
    oa = torch.zeros((n, 4, 74 * w), dtype=torch.float32, device="cuda:0")
    ob = torch.full((n, 4, stride), -7.0, dtype=torch.float32, device="cuda:0")
    ob[:, :, : 74 * w] = 0.0
    torch.cuda.synchronize()
    a.step_random_encode(5, 150, oa.data_ptr(), auto_reset=True, only_active=2)      # one launch, rows of the acting seats
    b.step_random_encode(5, 150, ob.data_ptr(), auto_reset=True, only_active=2)
    a.sync(); b.sync()
    assert torch.equal(oa, ob[:, :, : 74 * w]) and bool((ob[:, :, 74 * w:] == -7.0).all()) and float(oa.abs().sum()) > 0
    # host copies (every seat) and the per-step encoder
    assert (a.encode() == b.encode()).all()
    # the dense compact batch
    cap = n * 2
    ca = torch.zeros((cap, 74 * w), dtype=torch.float32, device="cuda:0")
    cb = torch.full((cap, stride), -7.0, dtype=torch.float32, device="cuda:0")
    ia, ib = (torch.zeros((cap,), dtype=torch.int32, device="cuda:0") for _ in range(2))
    na, nb = (torch.zeros((1,), dtype=torch.int32, device="cuda:0") for _ in range(2))
    torch.cuda.synchronize()      # (the fills run on torch's stream, the library on its own)
    a.encode_compact_device(ca.data_ptr(), ia.data_ptr(), cap, na.data_ptr())
    b.encode_compact_device(cb.data_ptr(), ib.data_ptr(), cap, nb.data_ptr())
    a.sync(); b.sync()
    k = int(na.item())
    assert k == int(nb.item()) and k > n // 2 and torch.equal(ia[:k], ib[:k])
    assert torch.equal(ca[:k], cb[:k, : 74 * w]) and bool((cb[:, 74 * w:] == -7.0).all())
    a.close(); b.close()
    # the torch wrapper: a strided view of the padded buffer, step_obs / obs / obs_compact
    ta = TorchVecEnv(512, game_mode=mode, seed=3, pad_rows=False)
    tb = TorchVecEnv(512, game_mode=mode, seed=3)                  # (padded rows are the default)
    assert tb.pad_rows and not ta.pad_rows and ta.obs().is_contiguous() and not tb.obs().is_contiguous()
    assert tb.obs().shape == ta.obs().shape == (512, 4, 74, w)
    for k in range(60):
        ids = ta.sample_ids(seed=k).clone()
        xa, xb = ta.step_obs(ids), tb.step_obs(ids)
    assert torch.equal(xa, xb) and torch.equal(ta.obs(), tb.obs())
    (pa, qa), (pb, qb) = ta.obs_compact(), tb.obs_compact()
    assert torch.equal(pa, pb) and torch.equal(qa, qb)


@pytest.mark.parametrize("mode", [2, 5])
def test_row_stride_off_the_16_byte_grid(mode):
    """A row stride of 74 x W + 2 floats (strides are even: rmj_set_encode_row_stride) puts consecutive rows alternately on and 8 bytes off
    the 16-byte grid in 4P too (dense 3P rows already alternate): the byte-staged encoder - its cells shifted by two bytes, two floats
    in front of the first 16-byte store - must write the same floats as with dense rows: fused step + encode rollout, per-step encoder,
    the extended encoder's base block."""
    import torch

    from riichienv_amd import vecenv

    n, w = 1031, (27 if mode >= 3 else 34)
    a = vecenv.VecRiichiEnv(n, game_mode=mode, seed=17, event_ring=64)
    b = vecenv.VecRiichiEnv(n, game_mode=mode, seed=17, event_ring=64)
    stride = 74 * w + 2
    b.set_encode_row_stride(stride)
    for e in (a, b):
        e.reset()
    oa = torch.zeros((n, 4, 74 * w), dtype=torch.float32, device="cuda:0")
    ob = torch.full((n, 4, stride), -7.0, dtype=torch.float32, device="cuda:0")
    ob[:, :, : 74 * w] = 0.0
    torch.cuda.synchronize()
    a.step_random_encode(9, 120, oa.data_ptr(), auto_reset=True, only_active=2)
    b.step_random_encode(9, 120, ob.data_ptr(), auto_reset=True, only_active=2)
    a.sync(); b.sync()
    assert torch.equal(oa, ob[:, :, : 74 * w]) and bool((ob[:, :, 74 * w:] == -7.0).all()) and float(oa.abs().sum()) > 0
    ea, eb = a.encode(), b.encode()          # every seat through the per-step encoder (host copies; b's rows at stride 74 x W + 2)
    assert (ea == eb).all() and float(np.abs(ea).sum()) > 0
    assert (a.encode_extended() == b.encode_extended()).all()
    a.close(); b.close()


@pytest.mark.parametrize("mode", [2, 5])
def test_poked_state_with_counts_outside_the_value_table(mode):
    """Channel 63 (tiles seen / 4) of a poked state whose rivers hold one tile type 60+ times: the byte-staged encoder decodes such codes
    arithmetically (its value table ends at a count of 23) - base encoder and the extended encoder's base block equal the oracle's."""
    import copy

    from oracle import oracle
    from riichienv_amd import vecenv

    n = 8
    env = vecenv.VecRiichiEnv(n, game_mode=mode, seed=23, event_ring=64)
    env.reset()
    env.step_random(3, 40, auto_reset=False)
    ref = env.encode()
    g = 2
    v = env.peek(g)
    w = copy.deepcopy(v)
    np_ = 3 if mode >= 3 else 4
    tile = 4 * 27 + 1                      # an East (a column of both layouts)
    for q in range(np_):
        w.players[q].n_discards = 20
        for j in range(20):
            w.players[q].discards[j] = tile if j % 5 else 4 * 31 + (j % 4)
    env.poke(g, w)
    o = oracle.Game(game_mode=mode, seed=1)
    o.reset()
    o.poke(w)
    got, ext = env.encode(), env.encode_extended()
    for pid in range(np_):
        want = np.asarray(o.encode(pid, sanma=mode >= 3), np.float32).reshape(got[g, pid].shape)
        assert (got[g, pid] == want).all(), pid
        assert float(want[63].max()) >= 12.0               # 48 (3P) / 64 (4P) copies of the tile seen: the code lies outside the table
        assert (ext[g, pid, :74] == np.asarray(o.encode_extended(pid), np.float32).reshape(ext[g, pid].shape)[:74]).all(), pid
    others = [k for k in range(n) if k != g]
    assert (got[others] == ref[others]).all()                # ... and the rows of the other games leave through the table as before
    env.close()
