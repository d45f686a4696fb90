"""Row N4: the trainer-side loop (torch tensors on the GPU, zero-copy masks, id-based stepping) against the oracle
driven with the same action ids through find_action semantics."""
from riichienv_amd.shard import game_seed
import numpy as np
import pytest

from riichienv_amd import abi

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode,shared", [(2, True), (5, True), (2, False)])
def test_torch_env_matches_oracle(mode, shared):
    torch = pytest.importorskip("torch")
    from oracle import oracle
    from riichienv_amd.torch_env import TorchVecEnv

    n, seed = 48, 700 + mode
    sanma = mode >= 3
    env = TorchVecEnv(n, game_mode=mode, seed=seed, extended=False, skip_mjai_logging=False, share_stream=shared)
    games = [oracle.Game(game_mode=mode, seed=game_seed(seed, g)) for g in range(n)]
    for o in games:
        o.reset()
    gen = torch.Generator(device=env.device)
    gen.manual_seed(1234)
    enc_fn = games[0].L.orc_action_encode_3p if sanma else games[0].L.orc_action_encode
    for step in range(700):
        act = env.active().cpu().numpy()
        mask = env.mask.cpu().numpy()
        ids = env.sample_random_ids(gen)
        ids_h = ids.cpu().numpy()
        for g, o in enumerate(games):
            oa, _, od = o.status()
            assert bool(env.done()[g]) == bool(od), (step, g)
            if od:
                o.reset()
                continue
            acts = [abi.NO_ACTION] * 4
            for s in range(4):
                assert act[g, s] == bool((oa >> s) & 1), (step, g, s)
                if act[g, s]:
                    assert (mask[g, s] == o.mask(s)).all(), (step, g, s)
                    cand = [a for a in o.legal(s) if enc_fn(a) == int(ids_h[g, s])]
                    assert cand, (step, g, s, int(ids_h[g, s]))
                    acts[s] = cand[0]
            o.step(acts)
        env.step(ids, auto_reset=True)
        if step % 100 == 0:
            obs = env.obs(only_active=False).cpu().numpy()
            for g in (0, n - 1):
                for s in range(3 if sanma else 4):
                    assert obs[g, s].tobytes() == games[g].encode(s, sanma).tobytes(), (step, g, s)
            # the dense batch of the acting seats: same rows, (game, seat) order
            cobs, cidx = env.obs_compact()
            a2 = env.active().cpu().numpy()
            assert cidx.cpu().numpy().tolist() == [g * 4 + s for g in range(n) for s in range(4) if a2[g, s]]
            ch = cobs.cpu().numpy()
            for j, gi in enumerate(cidx.cpu().numpy()):
                assert ch[j].tobytes() == obs[int(gi) >> 2, int(gi) & 3].tobytes(), (step, int(gi))
    sc = env.scores().cpu().numpy()
    rk = env.ranks().cpu().numpy()
    for g, o in enumerate(games):
        v = o.peek()
        assert list(sc[g]) == [v.players[p].score for p in range(4)]
    assert (rk == env.env.ranks()).all()


@pytest.mark.parametrize("mode", [2, 5])
def test_greedy_ids_parity(mode):
    """The id path (rmj_step_ids_device = Observation.find_action + step) where rounds end in wins: the oracle's tenpai-seeking
    policy (tests/mjsoul_util.greedy_actions) picks an action, its id goes to the device, and the oracle takes the FIRST legal
    action with that id like find_action does (observation/mod.rs:117-129: ids do not tell a red five from a plain one).  Status,
    lists, masks, waits after every step; whole MJAI logs at the end."""
    torch = pytest.importorskip("torch")
    import json

    from oracle import oracle
    from riichienv_amd.torch_env import TorchVecEnv
    from tests.mjsoul_util import greedy_actions
    from tests.test_gpu_step import _compare

    n, seed = 24, 8100 + mode
    sanma = mode >= 3
    env = TorchVecEnv(n, game_mode=mode, seed=seed, skip_mjai_logging=False, share_stream=True, event_ring=16384)
    games = [oracle.Game(game_mode=mode, seed=game_seed(seed, g)) for g in range(n)]
    for o in games:
        o.reset()
    enc_fn = games[0].L.orc_action_encode_3p if sanma else games[0].L.orc_action_encode
    rng = np.random.default_rng(seed)
    swapped = 0
    for step in range(1, 650):
        ids = np.full((n, 4), -1, dtype=np.int32)
        for g, o in enumerate(games):
            if o.status()[2]:
                continue
            picks = greedy_actions(o, rng, sanma)
            acts = [abi.NO_ACTION] * 4
            for s, a in enumerate(picks):
                if a == abi.NO_ACTION:
                    continue
                ids[g, s] = enc_fn(a)
                acts[s] = next(x for x in o.legal(s) if enc_fn(x) == ids[g, s])
                swapped += acts[s] != a
            o.step(acts)
        env.step(torch.from_numpy(ids).to(env.device), auto_reset=False)
        torch.cuda.synchronize()
        _compare(env.env, games, range(n), step, check_state=False)
        _compare(env.env, games, [step % n], step, check_state=True)
    kinds = {}
    for g, o in enumerate(games):
        log = o.log()
        assert env.env.mjai_log(g) == log, g
        for x in log:
            t = json.loads(x)["type"]
            kinds[t] = kinds.get(t, 0) + 1
    assert kinds.get("hora", 0) >= 30 and swapped > 0, (kinds, swapped)


@pytest.mark.parametrize("mode", [2, 5])
def test_fused_masked_sampler(mode):
    """rmj_sample_ids_device: ids are legal (mask bit set) exactly for the acting seats, deterministic in (seed, state),
    follow the logits (a dominant logit always wins, a masked id never does), and are uniform over the legal ids without."""
    torch = pytest.importorskip("torch")
    from riichienv_amd.torch_env import TorchVecEnv

    n = 4096
    A = 60 if mode >= 3 else 82
    env = TorchVecEnv(n, game_mode=mode, seed=77, share_stream=True)
    for k in range(60):
        env.step(env.sample_ids(seed=k))
    # a seat that is to act but has NO legal action gets -1: the reference can deadlock a 3P seat that declared Riichi and
    # then Kita (legal in the riichi stage, state_3p/legal_actions.rs:241-243) when no tenpai-keeping discard is left
    act = env.active() & (env.nlegal > 0)
    mask = env.mask.to(torch.bool)
    ids = env.sample_ids(seed=123).clone()
    assert ((ids >= 0) == act).all()
    g, s = torch.nonzero(act, as_tuple=True)
    assert mask[g, s, ids[g, s].long()].all() and (ids[g, s] < A).all()
    assert (env.sample_ids(seed=123) == ids).all() and not (env.sample_ids(seed=124) == ids).all()
    # logits: the largest legal id gets +50 -> always chosen; an illegal id with +1000 is never chosen
    legal_ids = torch.where(mask, torch.arange(82, device=env.device)[None, None, :], torch.full((1,), -1, device=env.device))
    top = legal_ids.max(-1).values
    logits = torch.zeros((n, 4, 82), dtype=torch.float32, device=env.device)
    logits[g, s, top[g, s]] = 50.0
    logits[~mask] = 1000.0
    got = env.sample_ids(logits, seed=5)
    assert (got[g, s].long() == top[g, s]).all()
    # uniformity: over many seeds, the discard ids of one acting seat with >= 10 legal ids are all drawn, none dominates
    cnt = mask.sum(-1)
    k = int(torch.nonzero(act & (cnt >= 10))[0, 0]), int(torch.nonzero(act & (cnt >= 10))[0, 1])
    draws = torch.stack([env.sample_ids(seed=1000 + i)[k[0], k[1]].clone() for i in range(600)]).long()
    hist = torch.bincount(draws, minlength=82)[mask[k[0], k[1]]]
    c = int(cnt[k[0], k[1]])
    assert (hist > 0).all() and hist.max() < 3.0 * 600 / c and hist.min() > 600 / c / 3.5, (hist, c)


def test_sharded_env_equals_one_env():
    """Four shards on four torch streams end in the same games as one environment (seeds, walls and the sampler's noise are
    keyed by the global game index)."""
    torch = pytest.importorskip("torch")
    from riichienv_amd.torch_env import ShardedTorchVecEnv, TorchVecEnv

    n, seed, steps = 512, 4321, 300
    one = TorchVecEnv(n, game_mode=2, seed=seed)
    sh = ShardedTorchVecEnv(n, parts=4, game_mode=2, seed=seed)
    for k in range(steps):
        one.step(one.sample_ids(seed=k + 1))
        sh.step_policy(lambda e, obs, index, count: e.sample_ids(seed=k + 1))
    torch.cuda.synchronize()
    assert (one.env.step_counts() == sh.step_counts()).all()
    assert torch.equal(one.scores().cpu(), sh.scores().cpu())
    assert int(one.env.step_counts().sum()) > n * steps // 2
    # the dense batch of a shard indexes the shard's own games
    obs, idx = sh.shards[1].obs_compact()
    assert obs.shape[0] == idx.shape[0] and int(idx.max()) < sh.per * 4


@pytest.mark.parametrize("shared", [True, False])
def test_readme_loop_compact_logits(shared):
    """The quick-start loop of README.md: a policy over the compact observation batch returns compact logits [k, 82];
    sample_ids(logits, index=index) scatters them to the acting seats.  A dominant logit on the largest legal id of every row
    must be the id that is played."""
    torch = pytest.importorskip("torch")
    from riichienv_amd.torch_env import TorchVecEnv

    n = 1024
    env = TorchVecEnv(n, game_mode=2, seed=31, share_stream=shared)
    for k in range(40):
        obs, index = env.obs_compact()
        assert obs.shape[1:] == (74, 34) and obs.shape[0] == index.shape[0]
        g, s = (index // 4).long(), (index % 4).long()
        mask = env.mask[g, s].to(torch.bool)                               # [k, 82]
        top = torch.where(mask, torch.arange(82, device=env.device)[None, :], torch.full((1,), -1, device=env.device)).max(-1).values
        logits = torch.zeros((obs.shape[0], 82), dtype=torch.float32, device=env.device)
        has = top >= 0
        logits[has, top[has]] = 60.0
        ids = env.sample_ids(logits=logits, seed=k, index=index)
        assert (ids[g[has], s[has]].long() == top[has]).all()
        assert int((ids >= 0).sum()) == int(has.sum())
        env.step(ids)


@pytest.mark.parametrize("mode", [2, 5])
def test_points_on_device(mode):
    """rmj_points_device / rmj_get_points against RiichiEnv.points restated from env.rs:673-727 in float64 (ranks by score, ties by
    seat; (score - base) / 1000 * weight + uma[rank - 1]); unknown rules raise like the reference."""
    torch = pytest.importorskip("torch")
    from riichienv_amd.torch_env import TorchVecEnv

    n = 2048
    env = TorchVecEnv(n, game_mode=mode, seed=5)
    for k in range(300):
        env.step(env.sample_ids(seed=k))
    sc = env.scores().cpu().numpy().astype(np.int64)
    npl = 3 if mode >= 3 else 4
    presets = {"basic": (1.0, 35000.0, [40.0, 0.0, -40.0])} if npl == 3 else \
        {"basic": (1.0, 25000.0, [50.0, 10.0, -10.0, -50.0]), "ouza-tyoujyo": (0.0, 25000.0, [100.0, 40.0, -40.0, -100.0]),
         "ouza-normal": (0.0, 25000.0, [50.0, 20.0, -20.0, -50.0])}
    assert len({tuple(r) for r in sc[:, :npl]}) > 10          # the games have diverged
    for name, (w, base, uma) in presets.items():
        want = np.zeros((n, 4))
        for g in range(n):
            order = sorted(range(npl), key=lambda p: (-sc[g, p], p))
            for rank, p in enumerate(order):
                want[g, p] = (float(sc[g, p]) - base) / 1000.0 * w + uma[rank]
        assert (env.points(name).cpu().numpy() == want).all(), name
        assert (env.env.points(name) == want).all(), name
    with pytest.raises(ValueError):
        env.points("ouza-normal" if npl == 3 else "nope")


@pytest.mark.parametrize("mode", [2, 5])
def test_step_obs_equals_step_then_obs(mode):
    """rmj_step_ids_encode_device (one launch) leaves the tensor and the games of step() + obs(only_active=True)"""
    torch = pytest.importorskip("torch")
    from riichienv_amd.torch_env import TorchVecEnv

    n = 4096
    a = TorchVecEnv(n, game_mode=mode, seed=41, share_stream=True)
    b = TorchVecEnv(n, game_mode=mode, seed=41, share_stream=True)
    oa = a.obs(only_active=True)
    ob = b.obs(only_active=True)
    for k in range(150):
        ids = a.sample_ids(seed=k).clone()
        oa = a.step_obs(ids)
        b.step(ids)
        ob = b.obs(only_active=True)
        if k % 25 == 0:
            assert torch.equal(oa, ob), k
    torch.cuda.synchronize()
    assert torch.equal(oa, ob) and torch.equal(a.mask, b.mask) and torch.equal(a.scores(), b.scores())
    assert (a.env.step_counts() == b.env.step_counts()).all() and float(oa.abs().sum()) > 0


@pytest.mark.parametrize("mode", [2, 5])
def test_env_loop_inside_a_hip_graph(mode):
    """policy -> sampler -> step + encode captured ONCE as a HIP graph (torch.cuda.CUDAGraph on a side stream the environment is
    bound to) and replayed: same games, same tensors as the eager loop of a twin environment"""
    torch = pytest.importorskip("torch")
    from riichienv_amd.torch_env import TorchVecEnv

    n, per_graph, replays = 2048, 8, 40
    width = 27 if mode >= 3 else 34
    torch.manual_seed(3)
    w = (torch.randn(74 * width, 82, device="cuda") * 0.05).contiguous()

    def iteration(e):
        logits = (e._obs.reshape(n * 4, 74 * width) @ w).view(n, 4, 82)      # (rows of seats that do not act are ignored)
        return e.step_obs(e.sample_ids(logits=logits.contiguous(), seed=9))

    a = TorchVecEnv(n, game_mode=mode, seed=77, share_stream=True)
    b = TorchVecEnv(n, game_mode=mode, seed=77, share_stream=True)
    a.obs(only_active=True)
    b.obs(only_active=True)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        a.bind_stream()
        iteration(a)                      # warm-up on the capture stream: every lazy buffer exists before the capture
    iteration(b)
    side.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        for _ in range(per_graph):
            iteration(a)
    s0 = int(a.env.total_steps())
    for _ in range(replays):
        g.replay()
    torch.cuda.synchronize()
    da = int(a.env.total_steps()) - s0
    a.bind_stream(torch.cuda.current_stream())
    sb0 = int(b.env.total_steps())
    for _ in range(replays * per_graph):
        iteration(b)
    torch.cuda.synchronize()
    assert da == int(b.env.total_steps()) - sb0 and da > replays * per_graph * n // 2      # (the capture itself stepped nothing)
    assert torch.equal(a._obs, b._obs) and torch.equal(a.mask, b.mask) and torch.equal(a.scores(), b.scores())
    assert (a.env.step_counts() == b.env.step_counts()).all()


@pytest.mark.parametrize("mode", [2, 5])
def test_ids_without_a_legal_action_are_illegal_actions(mode):
    """An id no legal action of the seat carries (Observation.find_action returns None, observation/python.rs:119-122), a missing id
    (-1) for a seat that is to act, an id for a seat that is not: the step takes the reference's illegal-action path (penalty round
    end, state/mod.rs:339-402).  The oracle is given an action it rejects the same way."""
    torch = pytest.importorskip("torch")
    from oracle import oracle
    from riichienv_amd.torch_env import TorchVecEnv
    from tests.test_gpu_step import _compare

    n, seed = 48, 8300 + mode
    sanma = mode >= 3
    env = TorchVecEnv(n, game_mode=mode, seed=seed, skip_mjai_logging=False, share_stream=True, event_ring=16384)
    games = [oracle.Game(game_mode=mode, seed=game_seed(seed, g)) for g in range(n)]
    for o in games:
        o.reset()
    enc_fn = games[0].L.orc_action_encode_3p if sanma else games[0].L.orc_action_encode
    rng = np.random.default_rng(seed)
    bogus = abi.pack_action(0x7F)
    bad = 0
    for step in range(1, 500):
        ids = np.full((n, 4), -1, dtype=np.int32)
        for g, o in enumerate(games):
            oa, _, od = o.status()
            if od:
                continue
            acts = [abi.NO_ACTION] * 4
            for s in range(4):
                if not (oa >> s) & 1:
                    if rng.random() < 0.003:        # a seat that is not to act sends an id: ignored
                        ids[g, s] = int(rng.integers(30))
                    continue
                legal = o.legal(s)
                a = legal[int(rng.integers(len(legal)))]
                ids[g, s] = enc_fn(a)
                acts[s] = next(x for x in legal if enc_fn(x) == ids[g, s])
                u = rng.random()
                if u < 0.004:                       # an id nothing legal has
                    have = {enc_fn(x) for x in legal}
                    ids[g, s] = next(i for i in range(82) if i not in have)
                    acts[s] = bogus
                    bad += 1
                elif u < 0.006:                     # no id at all
                    ids[g, s] = -1
                    acts[s] = abi.NO_ACTION
                    bad += 1
            o.step(acts)
        env.step(torch.from_numpy(ids).to(env.device), auto_reset=False)
        torch.cuda.synchronize()
        _compare(env.env, games, range(n), step, check_state=False)
        _compare(env.env, games, [step % n, (3 * step) % n], step, check_state=True)
    assert bad > 20
    for g, o in enumerate(games):
        assert env.env.mjai_log(g) == o.log(), g


@pytest.mark.parametrize("mode", [2, 5])
def test_step_sample_obs_equals_sample_then_step_obs(mode):
    """rmj_step_sample_encode_device: the policy's draw, the step and the next observations as ONE launch give the ids, the tensors and the
    states of sample_ids(logits, seed) followed by step_obs(ids), with and without logits."""
    import torch

    from riichienv_amd.torch_env import TorchVecEnv

    n = 2050
    a = TorchVecEnv(n, game_mode=mode, seed=31)
    b = TorchVecEnv(n, game_mode=mode, seed=31)
    a.obs(only_active=True); b.obs(only_active=True)
    g = torch.Generator(device="cpu")
    g.manual_seed(4)
    for k in range(90):
        logits = None if k % 3 == 0 else (torch.randn((n, 4, 82), generator=g) * 2.0).cuda()
        ids_a = a.sample_ids(logits, seed=1000 + k).clone()
        xa = a.step_obs(ids_a)
        ids_b, xb = b.step_sample_obs(logits, seed=1000 + k)
        assert torch.equal(ids_a, ids_b), k
        assert torch.equal(xa, xb), k
    assert (a.env.step_counts() == b.env.step_counts()).all() and (a.env.scores() == b.env.scores()).all()
    la, ca = a.env.legal()
    lb, cb = b.env.legal()
    assert (ca == cb).all() and (la == lb).all() and (a.env.mask() == b.env.mask()).all()
    assert int(a.env.step_counts().sum()) > 80 * n


@pytest.mark.parametrize("mode", [2, 5])
def test_one_launch_paths_with_several_waves_per_simd_behind_foreign_kernels(mode):
    """Regression of the round-5 flake (docs/journal_r06.md section 1): rmj_step_ids_encode_device / rmj_step_sample_encode_device must equal
    step() + obs() at EVERY step when their launch has more than one wave per SIMD (8 192 games = 2 048 waves on 1 024 SIMDs) and runs right
    behind another library's kernels (torch's copies) - the build that drew the row ballot's shift amount into the last allocated register
    dropped list entries in ~40 games per step under exactly these conditions, in waves that were not the first of their SIMD."""
    torch = pytest.importorskip("torch")
    from riichienv_amd.torch_env import TorchVecEnv

    n = 8192
    a = TorchVecEnv(n, game_mode=mode, seed=43, share_stream=True)
    b = TorchVecEnv(n, game_mode=mode, seed=43, share_stream=True)
    c = TorchVecEnv(n, game_mode=mode, seed=43, share_stream=True)
    for e in (a, b, c):
        e.obs(only_active=True)
    for k in range(60):
        ids = b.sample_ids(seed=k).clone()
        oa = a.step_obs(ids.clone())                       # one launch, first behind the copy
        ic, oc = c.step_sample_obs(seed=k)                 # one launch, the draw inside
        b.step(ids)
        ob = b.obs(only_active=True)
        torch.cuda.synchronize()
        assert torch.equal(ic, ids), k
        for name, e, o in (("step_obs", a, oa), ("step_sample_obs", c, oc)):
            assert torch.equal(e.nlegal, b.nlegal), (name, k, (e.nlegal != b.nlegal).any(1).nonzero().flatten()[:8].tolist())
            assert torch.equal(e.mask, b.mask) and torch.equal(o, ob) and torch.equal(e.legal, b.legal), (name, k)
    assert (a.env.step_counts() == b.env.step_counts()).all() and (c.env.step_counts() == b.env.step_counts()).all()


def test_ticket_rollout_inside_a_hip_graph_can_be_replayed():
    """the fused ticket rollout (k_step4_queue + k_step4_fixup) must be idempotent as a pair of launches - a rollout captured ONCE in a
    HIP graph and replayed steps every game again.  (Round 5 alternated two counter sets from the host: a replay of the one captured launch found
    its tickets exhausted and silently stepped nothing.  Now k_step4_fixup re-arms the set it has looked at.)"""
    import os

    torch = pytest.importorskip("torch")
    from riichienv_amd import vecenv
    from riichienv_amd.torch_env import TorchVecEnv

    n, k, pseed, replays = 4096, 40, 77, 3
    old = {v: os.environ.get(v) for v in ("RMJ_QUEUE_FORCE", "RMJ_QUEUE_CHUNK")}
    os.environ.update({"RMJ_QUEUE_FORCE": "1", "RMJ_QUEUE_CHUNK": "8"})
    try:
        a = TorchVecEnv(n, game_mode=2, seed=5, share_stream=True)
    finally:
        for v, x in old.items():
            os.environ.pop(v, None) if x is None else os.environ.__setitem__(v, x)
    if not int(a.env.bench_rollout(pseed, 0, k).queued):
        pytest.skip("this build / device does not run the rollout as tickets")
    b = vecenv.VecRiichiEnv(n, game_mode=2, seed=5)
    b.reset()
    b.step_random(pseed, k, auto_reset=True)              # (the bench call above)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        a.bind_stream()
        a.env.step_random(pseed, k, auto_reset=True)      # warm-up on the capture stream: every lazy buffer exists before the capture
    side.synchronize()
    s0 = int(a.env.total_steps())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        a.env.step_random(pseed, k, auto_reset=True)
    assert int(a.env.total_steps()) == s0                 # (the capture itself stepped nothing)
    per_replay = []
    for _ in range(replays):
        g.replay()
        torch.cuda.synchronize()
        per_replay.append(int(a.env.total_steps()))
    assert per_replay[0] - s0 == n * k and all(per_replay[i + 1] - per_replay[i] == n * k for i in range(replays - 1)), (s0, per_replay)
    for _ in range(1 + replays):
        b.step_random(pseed, k, auto_reset=True)
    a.bind_stream(torch.cuda.current_stream())
    assert (a.env.step_counts() == b.step_counts()).all() and (a.env.scores() == b.scores()).all()
