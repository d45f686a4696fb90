"""GPU parity of the auxiliary encoders (rmj_encode_aux: kawa overview / yaku possibility / furiten-ron possibility)
against the oracle: the hand-built cases of tests/aux_cases.py and every game of random rollouts, 4P and 3P."""
from riichienv_amd.shard import game_seed
import numpy as np
import pytest

from tests.aux_cases import CASES, apply_case
from tests.env_adapters import GpuEnv, OracleEnv
from tests.scenarios import setup

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", [2, 5])
@pytest.mark.parametrize("name", sorted(CASES))
def test_aux_encoders_hand_built_cases(name, mode):
    envs = []
    for make in (OracleEnv, GpuEnv):
        env = make(game_mode=mode, seed=5)
        setup(env, hands=[[4 * k + p for k in range(13)] for p in range(4)], drawn_tile=None,
              mutate=lambda v: apply_case(v, CASES[name], mode >= 3))
        envs.append(env)
    o, g = envs
    assert g.e.encode_kawa_overview()[0].tobytes() == o.g.encode_kawa_overview().tobytes()
    assert g.e.encode_yaku_possibility()[0].tobytes() == o.g.encode_yaku_possibility().tobytes()
    assert g.e.encode_furiten_ron_possibility()[0].tobytes() == o.g.encode_furiten_ron_possibility().tobytes()


@pytest.mark.parametrize("mode", [2, 5])
def test_aux_encoders_along_rollout(mode):
    from oracle import oracle
    from riichienv_amd import vecenv

    n, seed, pseed = 24, 777, 3
    env = vecenv.VecRiichiEnv(n, game_mode=mode, seed=seed)
    games = [oracle.Game(game_mode=mode, seed=game_seed(seed, g)) for g in range(n)]
    env.reset()
    for o in games:
        o.reset()
    checked_zero = 0
    for step in range(700):
        acts = np.array([games[g].random_actions(pseed, g) for g in range(n)], dtype=np.uint64)
        env.step(acts)
        for g in range(n):
            games[g].step([int(x) for x in acts[g]])
        if step % 35 == 0 or step == 699:
            k, y, f = env.encode_kawa_overview(), env.encode_yaku_possibility(), env.encode_furiten_ron_possibility()
            for g in range(n):
                rk, ry = games[g].encode_kawa_overview(), games[g].encode_yaku_possibility()
                assert k[g].tobytes() == rk.tobytes(), (step, g, np.argwhere(k[g] != rk)[:5])
                assert y[g].tobytes() == ry.tobytes(), (step, g, np.argwhere(y[g] != ry)[:5])
                assert (f[g] == 1.0).all()
                checked_zero += int((ry == 0).sum())
    assert checked_zero > 100   # melds and dead honors did occur


def test_compat_observation_aux_blocks():
    """reference-named accessors: Observation.encode_kawa_overview / encode_yaku_possibility / encode_furiten_ron_possibility
    return the bytes of the (np, 7, W) / (np, 21, 2) / (np, 21) arrays (src/riichienv/_riichienv.pyi:315-341)."""
    from riichienv_amd.compat import RiichiEnv

    env = RiichiEnv(game_mode="4p-red-half", seed=3)
    obs = env.reset()
    o = next(iter(obs.values()))
    assert len(o.encode_kawa_overview()) == 4 * 7 * 34 * 4
    assert len(o.encode_yaku_possibility()) == 4 * 21 * 2 * 4
    assert len(o.encode_furiten_ron_possibility()) == 4 * 21 * 4
    env3 = RiichiEnv(game_mode="3p-red-half", seed=3)
    o3 = next(iter(env3.reset().values()))
    assert len(o3.encode_kawa_overview()) == 3 * 7 * 27 * 4 and len(o3.encode_yaku_possibility()) == 3 * 21 * 2 * 4
    assert len(o3.encode_furiten_ron_possibility()) == 3 * 21 * 4
