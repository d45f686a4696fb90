"""Row N2 (MJAI part): lock-step replay of logs into (obs, action) samples.  The real hanchan log of the reference's
tests (tests/data/126_204_0_mjai.jsonl) must yield a legal, selectable action for every decision event, and the samples
must equal the same pipeline run on the oracle."""
import os

import numpy as np
import pytest

from riichienv_amd import abi, mjai

pytestmark = pytest.mark.gpu
LOG = os.path.join(os.path.dirname(__file__), "golden", "126_204_0_mjai.jsonl")


def test_real_log_yields_every_decision():
    from oracle import oracle
    from riichienv_amd import replay

    events = replay.load_mjai_jsonl(LOG)
    rb = replay.ReplayBatch([events, events[:400]], game_mode=2, include_pass=False)
    o = oracle.Game(game_mode=2, seed=1)
    o.reset()
    want = {"dahai": 0, "pon": 0, "chi": 0, "reach": 0, "hora": 0, "ankan": 0}
    for e in events:
        if e["type"] in want:
            want[e["type"]] += 1
    got = dict.fromkeys(want, 0)
    k_prev = 0
    n_samples = 0
    for smp in rb.samples():
        # bring the oracle to the same event index, then compare game 0's samples
        while k_prev < smp["index"]:
            o.apply_event(events[k_prev])
            k_prev += 1
        for j in np.where(smp["game"] == 0)[0]:
            s = int(smp["seat"][j])
            ev = events[smp["index"]]
            got[ev["type"] if ev["type"] in got else "dahai"] += 1
            legal = o.legal(s)
            v = o.peek()
            sel = mjai.select_action_from_mjai(legal, ev, None if v.drawn_tile < 0 else int(v.drawn_tile), False)
            assert sel == int(smp["action"][j]), (smp["index"], ev)
            assert (smp["mask"][j] == np.asarray(o.mask(s))).all()
            assert smp["obs"][j].tobytes() == o.encode(s, False).tobytes(), (smp["index"], s)
            assert smp["mask"][j][smp["action_id"][j]] == 1
            n_samples += 1
    assert got == want, (got, want)            # every decision event of the log produced a sample
    assert n_samples == sum(want.values())


def test_pass_samples_and_extended_features():
    from riichienv_amd import replay

    events = replay.load_mjai_jsonl(LOG)[:700]
    rb = replay.ReplayBatch([events], game_mode=2, include_pass=True, extended=True)
    n_pass = n_all = 0
    for smp in rb.samples():
        assert smp["obs"].shape[1:] == (215, 34)
        for a, aid in zip(smp["action"], smp["action_id"]):
            n_all += 1
            if abi.unpack_action(int(a))[0] == abi.PASS:
                n_pass += 1
                assert aid == 81
    assert n_pass > 0 and n_all > n_pass
