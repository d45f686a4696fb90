"""Oracle pinned on outputs of the reference itself (tests/ui_example_util.py): the twelve wins and the 477 discards of the
reference's UI example log."""
from oracle import oracle
from riichienv_amd.replay import MjaiReplay
from tests import ui_example_util as U


def test_the_twelve_wins_score_as_the_reference_scored_them():
    events = U.load()
    ctxs = [c for k in MjaiReplay.from_jsonl(U.LOG).take_kyokus() for c in k.take_win_result_contexts()]
    for c, r in zip(ctxs, oracle.eval_hands([c.hand_case() for c in ctxs])):
        c.actual = r
    U.check_scores(events, ctxs)


def test_waits_after_every_discard_are_the_reference_s():
    events = U.load()
    o = oracle.Game(game_mode=2, seed=1)
    o.reset()
    cases, want = [], []
    for e in events:
        o.apply_event(U.plain(e))
        if e["type"] == "dahai":
            cases.append(U.hand_case_of(o.peek().players[e["actor"]]))
            want.append(e.get("meta", {}).get("waits", []))
            assert U.wait_names(o.waits(e["actor"])) == want[-1]          # the state machine's wait cache of the discarder
    got = [U.wait_names(r.waits) for r in oracle.eval_hands(cases)]
    assert got == want and len(want) == 477 and sum(bool(w) for w in want) == 64
