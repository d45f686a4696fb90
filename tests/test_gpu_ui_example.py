"""The reference's UI example log (tests/ui_example_util.py: a real hanchan carrying the reference's own han / fu / yaku /
points of its twelve wins and its waits after every discard) through the GPU paths: rmj_eval_hands on the reconstructed win
contexts, rmj_apply_events + rmj_eval_hands on the discarder's hand and melds after each discard."""
import pytest

from tests import ui_example_util as U

pytestmark = pytest.mark.gpu


def test_wins_and_waits_of_the_ui_example_on_the_gpu():
    from riichienv_amd import vecenv
    from riichienv_amd.replay import MjaiReplay, evaluate_win_contexts

    events = U.load()
    ctxs = evaluate_win_contexts([c for k in MjaiReplay.from_jsonl(U.LOG).take_kyokus() for c in k.take_win_result_contexts()])
    U.check_scores(events, ctxs)
    env = vecenv.VecRiichiEnv(1, game_mode=2, seed=1, skip_mjai_logging=True)
    env.reset()
    cases, want = [], []
    for e in events:
        env.apply_events([U.plain(e)])
        if e["type"] == "dahai":
            want.append(e.get("meta", {}).get("waits", []))
            assert int(env.waits()[0, e["actor"]]) == 0          # (published for the seats that get an observation, env.rs:870-871)
            cases.append(U.hand_case_of(env.peek(0).players[e["actor"]]))
    got = [U.wait_names(r.waits) for r in vecenv.eval_hands(cases)]
    assert got == want and sum(bool(w) for w in want) == 64
    env.close()


def test_every_decision_of_the_ui_example_is_selectable():
    """ReplayBatch over the same log: each dahai / pon / chi / reach / hora of the log is found among the legal actions the
    device published for its actor (477 + 16 + 2 + 8 + 12 decisions), and its id is set in the mask."""
    from riichienv_amd import abi, replay

    events = [U.plain(e) for e in U.load()]
    want = {"dahai": 477, "pon": 16, "chi": 2, "reach": 8, "hora": 12}
    got = dict.fromkeys(want, 0)
    rb = replay.ReplayBatch([events], game_mode=2, include_pass=False)
    for smp in rb.samples():
        ty = events[smp["index"]]["type"]
        for j in range(len(smp["seat"])):
            assert smp["mask"][j][smp["action_id"][j]] == 1
            assert int(smp["seat"][j]) == events[smp["index"]]["actor"]
            got[ty] += 1
            kind = abi.unpack_action(int(smp["action"][j]))[0]
            assert kind == {"dahai": abi.DISCARD, "pon": abi.PON, "chi": abi.CHI, "reach": abi.RIICHI}.get(ty, kind)
    assert got == want
