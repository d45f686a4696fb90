"""Row N2, sanma part (the reference walks 3P logs with KyokuStepIterator3P, replay/mod.rs:112): the reference holds no 3P
log, so the log is produced here by the oracle - a sanma hanchan played by a policy that takes every win and riichi it is
offered and otherwise picks a seeded random legal action - and then replayed in lock-step on the GPU.  Every decision the
policy took must come back as a sample: same packed action, same 60-wide mask, the acting seat's 74 x 27 tensor byte-equal
to the oracle's, and kita / pon / reach / hora all present."""
import json

import numpy as np
import pytest

from riichienv_amd import abi, mjai
from riichienv_amd.shard import game_seed

pytestmark = pytest.mark.gpu


def play_sanma_log(seed):
    """(events, decisions): the full MJAI log of one 3p-red-half game and, per step, {seat: packed action} of the seats that
    did not pass."""
    from oracle import oracle

    o = oracle.Game(game_mode=5, seed=game_seed(4242, seed))
    o.reset()
    rng = np.random.default_rng(seed)
    taken = []
    for _ in range(6000):
        act, _, done = o.status()
        if done:
            break
        acts = [abi.NO_ACTION] * 4
        for s in range(3):
            if not (act >> s) & 1:
                continue
            legal = o.legal(s)
            if not legal:
                continue
            eager = [a for a in legal if abi.unpack_action(a)[0] in (abi.TSUMO, abi.RON, abi.RIICHI)]
            a = eager[0] if eager else legal[int(rng.integers(len(legal)))]
            acts[s] = a
            if abi.unpack_action(a)[0] != abi.PASS:
                taken.append(abi.unpack_action(a)[0])
        o.step(acts)
    assert o.status()[2]
    return [json.loads(s) for s in o.log()], taken


@pytest.mark.parametrize("seed", [2, 4])   # both contain reach + hora; 4 also ankan, kakan and a ron
def test_sanma_log_replays_into_the_policy_decisions(seed):
    from oracle import oracle
    from riichienv_amd import replay

    events, taken = play_sanma_log(seed)
    kinds = {e["type"] for e in events}
    assert {"kita", "pon", "reach", "hora", "dahai"} <= kinds, kinds
    rb = replay.ReplayBatch([events], game_mode=5, include_pass=False)
    o = oracle.Game(game_mode=5, seed=1)
    o.reset()
    k_prev = 0
    got = []
    for smp in rb.samples():
        while k_prev < smp["index"]:
            o.apply_event(events[k_prev], replay=True)
            k_prev += 1
        ev = events[smp["index"]]
        for j in range(len(smp["game"])):
            s = int(smp["seat"][j])
            legal = o.legal(s)
            v = o.peek()
            sel = mjai.select_action_from_mjai(legal, ev, None if v.drawn_tile < 0 else int(v.drawn_tile), True)
            assert sel == int(smp["action"][j]), (smp["index"], ev)
            assert smp["mask"].shape[1] == 60 and (smp["mask"][j] == np.asarray(o.mask(s))[:60]).all()
            assert smp["obs"][j].shape == (74, 27) and smp["obs"][j].tobytes() == o.encode(s, True).tobytes(), (smp["index"], s)
            assert smp["mask"][j][smp["action_id"][j]] == 1
            got.append(abi.unpack_action(sel)[0])
    # the replay yields exactly the non-pass decisions of the policy, in order (a riichi step is reach + its discard)
    assert got == taken, (len(got), len(taken))
