"""Rows A14 / N3 under play that reaches the states a RandomAgent almost never does (riichi declared and accepted, ippatsu, kan dora,
furiten, tenpai hands, rounds that end with wins): every encoder against the oracle along rollouts of the greedy policy
(rmj_step_greedy on the device, orc_game_greedy_actions in the oracle - the two play the same games, tests/test_gpu_greedy.py)."""
import numpy as np
import pytest

from riichienv_amd import abi, vecenv
from riichienv_amd.shard import game_seed

pytestmark = pytest.mark.gpu


def _setup(mode, rule, seed, n, ring=4096):
    from oracle import oracle

    env = vecenv.VecRiichiEnv(n, game_mode=mode, seed=seed, rule_bits=rule, event_ring=ring)
    env.reset()
    games = [oracle.Game(game_mode=mode, seed=game_seed(seed, g), rule_bits=rule) for g in range(n)]
    for o in games:
        o.reset()
    return env, games


def _advance(env, games, pseed, rate, k):
    env.step_greedy(pseed, k, auto_reset=True, call_rate_256=rate)
    for _ in range(k):
        for g, o in enumerate(games):
            if o.status()[2]:
                o.reset()
                continue
            o.step([int(x) for x in o.greedy_actions(pseed, g, rate)])


@pytest.mark.parametrize("mode,rule,rate", [(2, abi.RULE_TENHOU, 64), (5, abi.RULE_MJSOUL, 64), (0, abi.RULE_MJSOUL, 0)])
def test_base_and_aux_encoders_along_greedy_play(mode, rule, rate):
    n, seed, pseed = 32, 9100 + mode, 0xFACE
    sanma = mode >= 3
    env, games = _setup(mode, rule, seed, n)
    riichi_rows = 0
    for it in range(80):
        _advance(env, games, pseed, rate, 11)
        enc = env.encode()
        act, ph, dn = env.status()
        enc_act = env.encode(only_active=True)
        k, y = env.encode_kawa_overview(), env.encode_yaku_possibility()
        for g, o in enumerate(games):
            assert (int(act[g]), int(ph[g]), int(dn[g])) == tuple(int(x) for x in o.status()), (it, g)
            for s in range(3 if sanma else 4):
                ref = o.encode(s, sanma)
                assert enc[g, s].tobytes() == ref.tobytes(), (it, g, s, np.argwhere(enc[g, s] != ref)[:6])
                if (act[g] >> s) & 1 and not dn[g]:
                    assert enc_act[g, s].tobytes() == ref.tobytes(), (it, g, s)
            rk, ry = o.encode_kawa_overview(), o.encode_yaku_possibility()
            assert k[g].tobytes() == rk.tobytes(), (it, g, np.argwhere(k[g] != rk)[:5])
            assert y[g].tobytes() == ry.tobytes(), (it, g, np.argwhere(y[g] != ry)[:5])
            riichi_rows += sum(1 for e in o.log()[-40:] if '"reach_accepted"' in e)
    assert riichi_rows > 50      # the sampled states did include accepted riichi
    env.close()


@pytest.mark.parametrize("mode,rule,rate", [(2, abi.RULE_TENHOU, 64), (5, abi.RULE_MJSOUL, 32)])
def test_extended_encoder_along_greedy_play(mode, rule, rate):
    """(the oracle's shanten of the extended encoder is an enumeration: few samples, taken late in the rounds)"""
    n, seed, pseed = 16, 9200 + mode, 0xD1CE
    sanma = mode >= 3
    env, games = _setup(mode, rule, seed, n)
    for it in range(24):
        _advance(env, games, pseed, rate, 37)
        enc = env.encode_extended()
        for g, o in enumerate(games):
            for s in range(3 if sanma else 4):
                ref = o.encode_extended(s)
                bad = np.argwhere(enc[g, s] != ref)
                assert enc[g, s].tobytes() == ref.tobytes(), (it, g, s, bad[:6])
    env.close()


@pytest.mark.parametrize("mode,seed", [(2, 21), (0, 22)])
def test_seq_features_along_greedy_play(mode, seed):
    from tests.test_gpu_seq_features import _check

    n, pseed = 12, 0xAB
    env, games = _setup(mode, abi.RULE_TENHOU, seed, n, ring=2048)
    for it in range(60):
        _advance(env, games, pseed, 64, 13)
        out = env.encode_seq(1)
        for g, o in enumerate(games):
            _check(out, g, o, it, all_seats=True)
    env.close()
