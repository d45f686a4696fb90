"""The greedy device policy (rmj_step_greedy: every win / riichi / kan / kita taken, shanten-greedy discards, calls at a rate)
against the oracle playing the same policy on its own legal lists and its own shanten (orc_game_greedy_actions).  Both sides
are deterministic functions of (seed, game, step, seat), so one wrong discard anywhere shows as a different game."""
import json

import pytest

from riichienv_amd import abi, vecenv
from riichienv_amd.shard import game_seed
from tests.parity_util import diff_dict, normalize_view
from tests.test_gpu_step import _compare

pytestmark = pytest.mark.gpu


def _oracle_games(mode, rule, seed, n, offset=0):
    from oracle import oracle

    games = [oracle.Game(game_mode=mode, seed=game_seed(seed, offset + g), rule_bits=rule) for g in range(n)]
    for o in games:
        o.reset()
    return games


def _oracle_play(games, pseed, steps, rate, auto_reset, offset=0):
    for _ in range(steps):
        for g, o in enumerate(games):
            if o.status()[2]:
                if auto_reset:
                    o.reset()
                continue
            o.step([int(x) for x in o.greedy_actions(pseed, offset + g, rate)])


@pytest.mark.parametrize("mode,rule,rate", [(2, abi.RULE_TENHOU, 64), (2, abi.RULE_MJSOUL, 0), (5, abi.RULE_MJSOUL, 64),
                                            (5, abi.RULE_TENHOU, 255), (0, abi.RULE_TENHOU, 128)])
def test_greedy_policy_step_by_step(mode, rule, rate):
    """one launch per step (k_step4<false, greedy>): status, ordered lists, masks, waits after every step, states on a sample"""
    n, seed, pseed, steps = 64, 4100 + mode, 0xBEEF, 900
    env = vecenv.VecRiichiEnv(n, game_mode=mode, seed=seed, rule_bits=rule, event_ring=16384)
    env.reset()
    games = _oracle_games(mode, rule, seed, n)
    wins_checked = 0
    for k in range(1, steps + 1):
        _oracle_play(games, pseed, 1, rate, auto_reset=False)
        env.step_greedy(pseed, 1, auto_reset=False, call_rate_256=rate)
        _compare(env, games, range(n), k, check_state=False)
        _compare(env, games, [k % n, (7 * k) % n], k, check_state=True)
        # RiichiEnv.win_results (env.rs:606-607): the winners' WinResult - yaku ids in order, han, fu, payments, pao payer - of every
        # game whose oracle holds one, and of a rotating game that may hold none
        for g in [g for g in range(n) if g == k % n or games[g].win_results()]:
            want = games[g].win_results()
            assert env.win_results(g) == want, (k, g, env.win_results(g), want)
            wins_checked += len(want)
    _compare(env, games, range(n), -1, check_state=True)
    assert wins_checked > 40, wins_checked
    kinds = set()
    for g, o in enumerate(games):
        log = o.log()
        assert env.mjai_log(g) == log, g
        kinds |= {json.loads(e)["type"] for e in log}
    assert {"hora", "reach", "reach_accepted"} | ({"pon"} if rate else set()) <= kinds, kinds   # the policy wins, declares riichi and calls
    env.close()


@pytest.mark.parametrize("mode,rule,n,queue", [(2, abi.RULE_TENHOU, 192, "0"), (5, abi.RULE_MJSOUL, 192, "0"), (2, abi.RULE_TENHOU, 1024, "1")])
def test_greedy_fused_rollout_equals_oracle(mode, rule, n, queue, monkeypatch):
    """the fused rollout (k_step4<true, greedy>, and as tickets: k_step4_queue<greedy>) with auto-reset: every game ends where the
    oracle's game ends, sampled whole logs are equal"""
    seed, pseed, rate, steps, off = 5200 + mode, 0xFACE, 64, 700, 3 * n
    if queue == "1":
        monkeypatch.setenv("RMJ_QUEUE_FORCE", "1")
    else:
        monkeypatch.setenv("RMJ_QUEUE_CHUNK", "0")
    env = vecenv.VecRiichiEnv(n, game_mode=mode, seed=seed, rule_bits=rule, event_ring=16384, game_offset=off)
    env.reset()
    sample = list(range(0, n, max(1, n // 96)))
    games = {g: o for g, o in zip(sample, _oracle_games_at(mode, rule, seed, [off + g for g in sample]))}
    for chunk in (300, 1, 399):
        env.step_greedy(pseed, chunk, auto_reset=True, call_rate_256=rate)
        for g, o in games.items():
            for _ in range(chunk):
                if o.status()[2]:
                    o.reset()
                    continue
                o.step([int(x) for x in o.greedy_actions(pseed, off + g, rate)])
    assert steps == 700
    act, ph, dn = env.status()
    legal, cnt = env.legal()
    for g in sample:
        o = games[g]
        oa, op, od = o.status()
        assert (act[g], ph[g], dn[g]) == (oa, op, od), g
        d = diff_dict(normalize_view(env.peek(g)), normalize_view(o.peek()))
        assert not d, (g, d[:10])
        for s in range(4):
            if (oa >> s) & 1 and not od:
                assert [int(x) for x in legal[g, s, : cnt[g, s]]] == o.legal(s), (g, s)
        log = o.log()
        dev = env.mjai_log(g)
        assert dev == log[len(log) - len(dev):] and len(dev) > 0, g
    env.close()


def _oracle_games_at(mode, rule, seed, globals_):
    from oracle import oracle

    games = [oracle.Game(game_mode=mode, seed=game_seed(seed, G), rule_bits=rule) for G in globals_]
    for o in games:
        o.reset()
    return games


@pytest.mark.parametrize("mode,bound", [(2, 0.001), (5, 0.004)])
def test_wins_and_round_ends_stay_in_the_four_games_per_wave_tier(mode, bound):
    """Round 4: settlements, yaku checks, exhaustive draws, next rounds and restarts are row-form code between the passes of a step
    (r4_round_end, r4_yaku_answers) - the serial full path is left with ankan in riichi, robbed kans, abortive draws (4P 0.02 %,
    3P 0.10 % of the game-steps, profiles/r04_bail_census.txt).  Guards against a change that silently sends those steps back
    (their parity is what the greedy rollout tests above and scripts/soak_parity.py compare)."""
    from riichienv_amd import vecenv

    env = vecenv.VecRiichiEnv(16384, game_mode=mode, seed=5, event_ring=64)
    env.reset()
    env.step_greedy(0xBEEF, 1200, auto_reset=True, call_rate_256=64)          # into the steady state: wins, riichi, kans, restarts
    s0, f0 = env.total_steps(), env.total_full_path()
    env.step_greedy(0xBEEF, 300, auto_reset=True, call_rate_256=64)           # fused (tickets)
    for _ in range(40):
        env.step_greedy(0xBEEF, 1, auto_reset=True, call_rate_256=64)         # one launch per step
    steps, full = env.total_steps() - s0, env.total_full_path() - f0
    assert steps > 16384 * 300 and full / steps < bound, (steps, full)
    env.close()
