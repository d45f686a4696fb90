"""The greedy policy's oracle twin (orc_game_greedy_actions, the checker of rmj_step_greedy) on the CPU: its choices come from the
legal lists, follow the class order of include/riichi_mi355x.h, are a pure function of (seed, game, step, seat), and the games it
plays end with wins and riichi - the transitions the RandomAgent almost never reaches."""
import json

import numpy as np
import pytest

from riichienv_amd import abi
from riichienv_amd.shard import game_seed

CLASS = {abi.TSUMO: 0, abi.RON: 0, abi.KITA: 1, abi.RIICHI: 2, abi.ANKAN: 3, abi.KAKAN: 4, abi.DAIMINKAN: 5, abi.PON: 6, abi.CHI: 7,
         abi.DISCARD: 8, abi.PASS: 9, abi.KYUSHU: 10}


@pytest.mark.parametrize("mode,rate", [(2, 64), (5, 255), (0, 0)])
def test_greedy_choices_follow_the_definition(mode, rate):
    from oracle import oracle

    sanma = mode >= 3
    n, pseed = 6, 4242
    games = [oracle.Game(game_mode=mode, seed=game_seed(31, g)) for g in range(n)]
    twins = [oracle.Game(game_mode=mode, seed=game_seed(31, g)) for g in range(n)]
    for o in games + twins:
        o.reset()
    kinds = set()
    for _ in range(2500):
        for g, (o, w) in enumerate(zip(games, twins)):
            act, _, done = o.status()
            if done:
                continue
            acts = o.greedy_actions(pseed, g, rate)
            assert acts == w.greedy_actions(pseed, g, rate)                     # a pure function of the state and the keys
            v = o.peek()
            for s in range(4):
                if not (act >> s) & 1:
                    assert acts[s] == abi.NO_ACTION
                    continue
                legal = o.legal(s)
                if not legal:
                    assert acts[s] == abi.NO_ACTION
                    continue
                assert acts[s] in legal
                ty = abi.unpack_action(acts[s])[0]
                best = min(CLASS[abi.unpack_action(a)[0]] for a in legal if abi.unpack_action(a)[0] not in (abi.PON, abi.CHI))
                assert CLASS[ty] <= best                                          # nothing of a better class was passed over (calls aside)
                if rate == 0:
                    assert ty not in (abi.PON, abi.CHI)
                if ty == abi.DISCARD:
                    # no other discard leaves a lower shanten
                    hand = list(v.players[s].hand[: v.players[s].hand_len])
                    cands = [a for a in legal if abi.unpack_action(a)[0] == abi.DISCARD]
                    cnt = np.zeros((len(cands), 34), np.uint8)
                    for i, a in enumerate(cands):
                        rest = list(hand)
                        rest.remove(abi.unpack_action(a)[1])
                        for t in rest:
                            cnt[i, t // 4] += 1
                    sh = oracle.shanten(cnt, sanma)
                    assert sh[cands.index(acts[s])] == sh.min()
            o.step(acts)
            w.step(acts)
    for o in games:
        kinds |= {json.loads(e)["type"] for e in o.log()}
    assert {"hora", "reach", "reach_accepted"} <= kinds, kinds


def test_usable_cores_is_bounded_by_the_affinity():
    import os

    import bench

    c = bench.usable_cores()
    assert 1 <= c <= (os.cpu_count() or 1)
    if hasattr(os, "sched_getaffinity"):
        assert c <= len(os.sched_getaffinity(0))
