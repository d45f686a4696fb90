"""Checkpoint / fork (RiichiEnv.clone / __copy__ / __deepcopy__, riichienv-python/src/env.rs:358-372): rmj_clone
copies a whole batch, rmj_copy_games single games.  A copy must be indistinguishable from its source - same records, outputs,
logs, and the same future under the same actions - and independent of it afterwards."""
import copy

import numpy as np
import pytest

from riichienv_amd import abi
from tests.parity_util import diff_dict, normalize_view

pytestmark = pytest.mark.gpu


def _same(a, ga, b, gb):
    assert not diff_dict(normalize_view(a.peek(ga)), normalize_view(b.peek(gb)))
    la, ca = a.legal()
    lb, cb = b.legal()
    live = np.arange(la.shape[-1])[None, :] < ca[ga][:, None]      # (slab entries behind a seat's count are leftovers of earlier lists)
    assert (ca[ga] == cb[gb]).all() and (np.where(live, la[ga], 0) == np.where(live, lb[gb], 0)).all()
    assert (a.mask()[ga] == b.mask()[gb]).all() and (a.waits()[ga] == b.waits()[gb]).all()
    assert [x[ga] for x in a.status()] == [x[gb] for x in b.status()]
    assert a.mjai_log(ga) == b.mjai_log(gb)


@pytest.mark.parametrize("mode", [2, 5])
def test_clone_is_the_same_batch_with_the_same_future(mode):
    from riichienv_amd import vecenv

    n = 64
    env = vecenv.VecRiichiEnv(n, game_mode=mode, seed=321, event_ring=4096)
    env.reset()
    env.step_random(0xABC, 150, auto_reset=True)
    twin = env.clone()
    for g in (0, 1, n // 2, n - 1):
        _same(env, g, twin, g)
    assert (env.step_counts() == twin.step_counts()).all() and (env.scores() == twin.scores()).all()
    # same policy key -> same future; the fused rollout on one, single steps on the other
    env.step_random(0xDEF, 120, auto_reset=True)
    for _ in range(120):
        twin.step_random(0xDEF, 1, auto_reset=True)
    for g in range(0, n, 7):
        _same(env, g, twin, g)
    # ... and independent: stepping the twin further leaves the original where it is
    before = env.mjai_log(3)
    counts = env.step_counts().copy()
    twin.step_random(0x123, 50, auto_reset=True)
    assert env.mjai_log(3) == before and (env.step_counts() == counts).all()
    twin.close()
    env.close()


def test_copy_games_forks_positions():
    """fork game 5 into three other slots of the same batch and into a second batch: all copies take the same actions to the same
    states; different actions diverge without touching the source"""
    from oracle import oracle  # (only to pick legal actions on the host: random_actions is a library call here)
    from riichienv_amd import vecenv

    del oracle
    env = vecenv.VecRiichiEnv(16, game_mode=2, seed=77, event_ring=4096)
    env.reset()
    env.step_random(9, 90, auto_reset=False)
    pool = vecenv.VecRiichiEnv(4, game_mode=2, seed=1, event_ring=4096)
    pool.reset()
    env.copy_games([1, 2, 3], env, [5, 5, 5])
    pool.copy_games([0, 3], env, [5, 5])
    for g in (1, 2, 3):
        _same(env, 5, env, g)
    for g in (0, 3):
        _same(env, 5, pool, g)
    # the same host actions for the copies: take the first legal action of every acting seat, 40 times
    for _ in range(40):
        legal, cnt = env.legal()
        act = env.status()[0]
        row = np.full(4, abi.NO_ACTION, dtype=np.uint64)
        for s in range(4):
            if (act[5] >> s) & 1 and cnt[5, s]:
                row[s] = legal[5, s, 0]
        a = np.full((16, 4), abi.NO_ACTION, dtype=np.uint64)
        a[[1, 2, 3, 5]] = row
        env.step(a)
        b = np.full((4, 4), abi.NO_ACTION, dtype=np.uint64)
        b[[0, 3]] = row
        pool.step(b)
    for g in (1, 2, 3):
        _same(env, 5, env, g)
    for g in (0, 3):
        _same(env, 5, pool, g)
    with pytest.raises(vecenv.RmjError):
        env.copy_games([99], env, [0])
    sanma = vecenv.VecRiichiEnv(2, game_mode=5, seed=1, event_ring=4096)
    with pytest.raises(vecenv.RmjError):
        sanma.copy_games([0], env, [0])          # a 4-player record does not fit a 3-player batch


def test_scalar_env_clone_copy_deepcopy():
    """env.rs:358-372 on the reference-named scalar environment"""
    from riichienv_amd.compat import RandomAgent, RiichiEnv

    env = RiichiEnv(game_mode="4p-red-half", seed=5)
    obs = env.reset()
    agent = RandomAgent(seed=1)
    for _ in range(60):
        obs = env.step({p: agent.act(o) for p, o in obs.items()})
    for twin in (env.clone(), copy.copy(env), copy.deepcopy(env)):
        assert twin.mjai_log == env.mjai_log and twin.scores() == env.scores() and twin.hands == env.hands
        o1, o2 = env.get_observations(), twin.get_observations()
        assert list(o1) == list(o2)
        for p in o1:
            assert o1[p].encode() == o2[p].encode() and o1[p].mask() == o2[p].mask()
            assert [(a.action_type, a.tile) for a in o1[p].legal_actions()] == [(a.action_type, a.tile) for a in o2[p].legal_actions()]
        n_log = len(env.mjai_log)
        twin.step({p: o.legal_actions()[0] for p, o in o2.items() if o.legal_actions()})
        assert len(env.mjai_log) == n_log and len(twin.mjai_log) >= n_log


def test_torch_env_forks_on_the_device():
    """TorchVecEnv.copy_games (rmj_copy_games_device, index tensors on the GPU, the environment's own stream): fork the first
    quarter of a batch into the other three quarters, step all with the same per-slot policy seeds -> the quarters stay identical;
    out-of-range pairs are skipped"""
    torch = pytest.importorskip("torch")
    from riichienv_amd.torch_env import TorchVecEnv

    n, q = 256, 64
    env = TorchVecEnv(n, game_mode=2, seed=9, skip_mjai_logging=False, event_ring=4096)
    for k in range(80):
        env.step(env.sample_ids(seed=k))
    src = torch.arange(q, device=env.device).repeat(3)
    dst = torch.arange(q, n, device=env.device)
    env.copy_games(dst, env, src)
    env.copy_games(torch.tensor([n + 5], device=env.device), env, torch.tensor([0], device=env.device))     # skipped
    torch.cuda.synchronize()
    for g in (0, 17, 63):
        for k in (1, 2, 3):
            _same(env.env, g, env.env, g + k * q)
    for k in range(40):
        ids = env.sample_ids(seed=1000 + k)
        ids[q:] = ids[:q].repeat(3, 1)                      # the forks take their original's actions
        env.step(ids, auto_reset=False)
    torch.cuda.synchronize()
    for g in (0, 17, 63):
        for k in (1, 2, 3):
            _same(env.env, g, env.env, g + k * q)
