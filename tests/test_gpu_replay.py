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
            o.apply_event(events[k_prev], replay=True)
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


def test_win_contexts_of_the_real_log_evaluated_in_one_gpu_batch(tmp_path):
    """Row N2: WinResultContextIterator over every kyoku of the reference's real hanchan log, all hora evaluated by ONE
    rmj_eval_hands launch (evaluate_win_contexts): the points must be the payments the log records, and equal the oracle's."""
    from oracle import oracle
    from riichienv_amd.replay import MjaiReplay, evaluate_win_contexts
    from tests.win_context_util import check_points, contexts_with_deltas, synthetic_log, write_jsonl

    items = contexts_with_deltas()
    ctxs = evaluate_win_contexts([c for _, c, _ in items])
    ref = oracle.eval_hands([c.hand_case() for c in ctxs])
    assert len(ctxs) == 9
    for (k, c, h), o in zip(items, ref):
        check_points(k, c, h, c.actual)
        for f in ("is_win", "yakuman", "han", "fu", "ron_agari", "tsumo_agari_oya", "tsumo_agari_ko", "n_yaku"):
            assert getattr(c.actual, f) == getattr(o, f), (k.chang, k.ju, f)
        assert list(c.actual.yaku[: c.actual.n_yaku]) == list(o.yaku[: o.n_yaku])
    p = tmp_path / "s.jsonl"
    write_jsonl(p, synthetic_log())
    (k,) = list(MjaiReplay.from_jsonl(str(p)).take_kyokus())
    (c,) = evaluate_win_contexts(k.take_win_result_contexts())
    assert c.actual.is_win and 3 in list(c.actual.yaku[: c.actual.n_yaku])   # chankan


def test_mjsoul_records_verified_in_one_gpu_batch():
    """MjSoulReplay.verify with its default evaluator (all wins of all rounds in ONE rmj_eval_hands launch): records converted
    from two oracle-played games carry the oracle's han / fu / yaku as expectations; the GPU must confirm every one."""
    from oracle import oracle
    from riichienv_amd.replay import MjSoulReplay
    from tests.mjsoul_util import play_logged_game, to_mjsoul_rounds

    total = 0
    for mode, seed in ((2, 1), (5, 3)):
        events, walls = play_logged_game(mode, seed)
        plain = MjSoulReplay.from_dict(to_mjsoul_rounds(events, walls))
        ctxs = [c for k in plain.take_kyokus() for c in k.take_win_result_contexts()]
        res = oracle.eval_hands([c.hand_case() for c in ctxs])
        exp = {i: dict(count=r.han, fu=r.fu, fans=list(r.yaku[: r.n_yaku])) for i, r in enumerate(res)}
        n, bad = MjSoulReplay.from_dict(to_mjsoul_rounds(events, walls, expectations=exp)).verify()
        assert n == len(ctxs) and bad == 0, (mode, seed, n, bad)
        total += n
    assert total >= 4


def test_mjsoul_game_end_scores_with_the_gpu_tenpai_batch():
    """MjSoulReplay.from_dict / from_dicts with their default is_tenpai evaluator (rmj_eval_hands, one launch for all the
    exhaustive draws): every round of two oracle-played games, each as the last round of its own record, ends with the scores
    the next round starts with; the whole record ends with the scores of the oracle's game (mjsoul_replay.rs:259-339)."""
    from oracle import oracle
    from riichienv_amd.replay import MjSoulReplay
    from tests.mjsoul_util import play_logged_game, to_mjsoul_rounds

    draws = 0
    for mode, seed in ((2, 2), (5, 2)):
        events, walls, final = play_logged_game(mode, seed, with_scores=True)
        plain = MjSoulReplay.from_dict(to_mjsoul_rounds(events, walls))
        ctxs = [c for k in plain.take_kyokus() for c in k.take_win_result_contexts()]
        res = oracle.eval_hands([c.hand_case() for c in ctxs])
        exp = {i: dict(count=r.han, fu=r.fu, fans=list(r.yaku[: r.n_yaku]), yiman=bool(r.yakuman), point_rong=r.ron_agari,
                       point_zimo_qin=r.tsumo_agari_oya, point_zimo_xian=r.tsumo_agari_ko) for i, r in enumerate(res)}
        rounds = to_mjsoul_rounds(events, walls, expectations=exp)
        whole = MjSoulReplay.from_dict({"data": rounds})
        ks = list(whole.take_kyokus())
        assert ks[0].game_end_scores == final and ks[-1].end_scores == final
        singles = MjSoulReplay.from_dicts([[r] for r in rounds])
        for i, x in enumerate(singles):
            want = ks[i + 1].scores if i + 1 < len(ks) else final
            assert x.rounds[0].game_end_scores == want, (mode, seed, i)
            draws += ks[i].actions[-1]["name"] == "NoTile"
    assert draws >= 4


@pytest.mark.parametrize("mode,seed", [(2, 5), (5, 4)])
def test_mjsoul_records_replay_into_the_same_samples_as_their_mjai_log(mode, seed):
    """Per-step observations of Mahjong Soul records (the reference's LogKyoku.steps over a record, replay/mod.rs:1094-1290):
    MjSoulReplay.to_mjai feeds ReplayBatch.  A record written from an oracle-played game and the game's own MJAI log, replayed
    side by side in ONE batch, must give the same decisions - seat, action id, mask - and the same feature tensors, except at
    the discard after an open kan, where the record reveals the new indicator with the discard and the log just before it."""
    from riichienv_amd import replay
    from tests.mjsoul_util import play_logged_game, to_mjsoul_rounds

    events, walls = play_logged_game(mode, seed)
    oracle_tenpai = None      # (the GPU default: rmj_eval_hands)
    rec = replay.MjSoulReplay.from_dict(to_mjsoul_rounds(events, walls), tenpai=oracle_tenpai).to_mjai()
    moved = 0
    strip = [e for e in events if e["type"] != "dora"]
    assert len(rec) == len(events)
    for a, b in zip(events, rec):
        moved += (a["type"] == "dora") != (b["type"] == "dora")
    assert len(strip) < len(events)
    rb = replay.ReplayBatch([events, rec], game_mode=mode, include_pass=True)
    n = differ = 0
    for smp in rb.samples():
        g = smp["game"]
        i0, i1 = np.where(g == 0)[0], np.where(g == 1)[0]
        if moved and len(i0) != len(i1):
            continue                                         # an index where one stream holds the dora event, the other the discard
        assert len(i0) == len(i1), smp["index"]
        for x, y in zip(i0, i1):
            assert (int(smp["seat"][x]), int(smp["action_id"][x])) == (int(smp["seat"][y]), int(smp["action_id"][y])), smp["index"]
            assert (smp["mask"][x] == smp["mask"][y]).all()
            differ += smp["obs"][x].tobytes() != smp["obs"][y].tobytes()
            n += 1
    assert n > 100 and differ <= moved, (n, differ, moved)


_REACH_TEHAIS = [["1p", "1p", "2p", "2p", "2p", "3p", "3p", "3p", "4p", "4p", "4p", "5z", "5z"],
                 ["1s", "1s", "1s", "2s", "2s", "2s", "3s", "3s", "3s", "4s", "4s", "4s", "6z"],
                 ["1z", "1z", "2z", "2z", "3z", "3z", "4z", "4z", "5z", "5z", "6z", "6z", "7z"],
                 ["5m", "5m", "6m", "6m", "7m", "7m", "8m", "8m", "9m", "9m", "1m", "1m", "2m"]]


@pytest.mark.parametrize("n", [3, 4])
def test_steps_reach_discard_observation_is_not_duplicated_state(n, tmp_path):
    """tests/test_mjai_replay.py:103-150 (3P, rule="mjsoul") and :208-257 (4P): Kyoku.steps replays reach + dahai as two decisions -
    Riichi is legal at the first, no longer at the second"""
    from riichienv_amd.compat import ActionType
    from riichienv_amd.replay import MjaiReplay
    from tests.win_context_util import write_jsonl

    data = [{"type": "start_game", "names": ["A", "B", "C", "D"][:n], "id": "test_reach"},
            {"type": "start_kyoku", "bakaze": "E", "kyoku": 1, "honba": 0, "kyoutaku": 0, "oya": 0, "scores": [35000 if n == 3 else 25000] * n,
             "dora_marker": "1p", "tehais": _REACH_TEHAIS[:n]},
            {"type": "tsumo", "actor": 0, "pai": "1p"}, {"type": "reach", "actor": 0}, {"type": "dahai", "actor": 0, "pai": "1p", "tsumogiri": True},
            {"type": "ryukyoku", "reason": "test"}, {"type": "end_kyoku"}, {"type": "end_game"}]
    p = tmp_path / "reach.jsonl"
    write_jsonl(p, data)
    kyoku = list(MjaiReplay.from_jsonl(str(p), rule="mjsoul" if n == 3 else None).take_kyokus())[0]
    steps = list(kyoku.steps(0, skip_single_action=False))
    assert len(steps) >= 2
    (riichi_obs, riichi_act), (discard_obs, discard_act) = steps[0], steps[1]
    assert riichi_act.action_type == ActionType.RIICHI and discard_act.action_type == ActionType.DISCARD
    assert ActionType.RIICHI in [a.action_type for a in riichi_obs.legal_actions()]
    assert ActionType.RIICHI not in [a.action_type for a in discard_obs.legal_actions()]
    # the dataset loop of riichienv_ml/datasets/mjai_logs.py:105-113: the action id is legal in the observation's mask
    for obs, act in steps:
        aid = act.encode_3p() if n == 3 else act.encode()
        mask = np.frombuffer(obs.mask(), dtype=np.uint8)
        assert mask.shape[0] == obs.action_space_size and mask[aid] == 1
        assert len(obs.encode()) == 74 * (27 if n == 3 else 34) * 4


def test_steps_of_a_real_round_order_and_filters():
    """Kyoku.steps on the first round of the reference's real hanchan log: all seats = the union of the per-seat iterators;
    Pass decisions come before the claim they lost to; skip_single_action drops the forced decisions only."""
    from riichienv_amd.compat import ActionType
    from riichienv_amd.replay import MjaiReplay

    k = next(iter(MjaiReplay.from_jsonl(LOG).take_kyokus()))
    every = list(k.steps(skip_single_action=False))
    assert len(every) > 60 and all(len(x) == 3 for x in every)
    for seat in range(4):
        mine = list(k.steps(seat, skip_single_action=False))
        ref = [(o, a) for s, o, a in every if s == seat]
        assert [(a.action_type, a.tile) for _, a in mine] == [(a.action_type, a.tile) for _, a in ref]
        assert all(o1.encode() == o2.encode() for (o1, _), (o2, _) in zip(mine, ref))
    kept = list(k.steps(skip_single_action=True))
    assert 0 < len(kept) <= len(every) and all(len(o.legal_actions()) > 1 for _, o, _ in kept)
    assert len(kept) == sum(len(o.legal_actions()) > 1 for _, o, _ in every)
    kinds = [a.action_type for _, _, a in every]
    assert ActionType.DISCARD in kinds and ActionType.PASS in kinds
    # every decision's action is one of the observation's legal actions
    for s, o, a in every:
        assert any(a.action_type == l.action_type and a.tile == l.tile and a.consume_tiles == l.consume_tiles for l in o.legal_actions())


def test_steps_reuses_its_environment_without_leaking_state():
    """Kyoku.steps keeps a pool of one-game environments; walking all twelve rounds of the real log (the dataset loop of
    riichienv_ml/datasets/mjai_logs.py:100-113: every round, every seat) on reused environments gives the decisions a fresh
    environment gives, and two interleaved iterators do not share one"""
    from riichienv_amd import replay
    from riichienv_amd.replay import MjaiReplay

    ks = list(MjaiReplay.from_jsonl(LOG).take_kyokus())
    first = [[(s, a.action_type, a.tile, o.encode()) for s, o, a in k.steps(skip_single_action=False)] for k in ks]
    replay._STEP_ENVS.clear()
    fresh = []
    for k in ks:
        fresh.append([(s, a.action_type, a.tile, o.encode()) for s, o, a in k.steps(skip_single_action=False)])
        replay._STEP_ENVS.clear()                  # a new environment for every round
    assert first == fresh and sum(len(x) for x in first) > 700
    per_seat = sum(len(list(k.steps(seat))) for k in ks for seat in range(4))
    assert per_seat == sum(len(list(k.steps())) for k in ks)
    a, b = ks[0].steps(skip_single_action=False), ks[1].steps(skip_single_action=False)
    mixed = [x for pair in zip(a, b) for x in pair]
    want = [x for pair in zip(first[0], first[1]) for x in pair]
    assert [(s, act.action_type, act.tile) for s, _, act in mixed] == [(s, t, tile) for s, t, tile, _ in want]


def test_pass_samples_carry_the_missed_ron_furiten():
    """tests/env/test_apply_event.py:535-632 (TestReplayFuriten) on ReplayBatch: a seat that lets a Ron go is in same-turn furiten
    until its own discard (the second 3m is offered again), in riichi for good (the second 3m yields no sample at all)."""
    from riichienv_amd import replay
    from tests.apply_events_util import furiten_log

    rb = replay.ReplayBatch([furiten_log(False), furiten_log(True)], game_mode=0, include_pass=True)
    got = {0: [], 1: []}
    for smp in rb.samples():
        for j in range(len(smp["game"])):
            if int(smp["seat"][j]) == 1 and abi.unpack_action(int(smp["action"][j]))[0] == abi.PASS:
                got[int(smp["game"][j])].append(bool(smp["mask"][j][79]))      # id 79 = Ron / Tsumo
    assert got == {0: [True, True], 1: [True]}


@pytest.mark.parametrize("mode,rate", [(2, 96), (5, 64)])
def test_logs_of_winning_play_yield_every_decision(mode, rate):
    """The same over logs that hold what the one real hanchan does not hold often enough: claims on a riichi declaration tile (the
    claimer decides AFTER reach_accepted), kans of every kind with their indicators, kita, wins of every kind - logs of the oracle
    playing the greedy policy.  Every decision event of every log must produce a sample with the oracle's legal list, mask and tensor."""
    import json

    from oracle import oracle
    from riichienv_amd import replay

    sanma = mode >= 3
    logs = []
    for g in range(8):
        o = oracle.Game(game_mode=mode, seed=1300 + g)
        o.reset()
        for _ in range(2500):
            if o.status()[2]:
                break
            o.step([int(x) for x in o.greedy_actions(53, g, rate)])
        logs.append([json.loads(x) for x in o.log()])
    types = ("dahai", "pon", "chi", "reach", "hora", "ankan", "daiminkan", "kakan", "kita")
    # what the same pipeline yields on the oracle: a decision event counts when the replayed state offers its action to the actor (the
    # event handler sets up no chankan claims, event_handler.rs:290-305, so the Ron on a robbed kan is not offered; later winners of a
    # multi-Ron find a finished round - the reference's iterator yields the first winner only, replay/mod.rs:483-485).  A win on the
    # replacement tile IS offered: replay mode keeps the walker's is_after_kan (apply_log_action, event_handler.rs:428).
    want, raw = dict.fromkeys(types, 0), dict.fromkeys(types, 0)
    robbed = set()      # (log, index) of the hora that rob a kan: yielded like the reference's walker does (replay/mod.rs:483-527)
    for li, log in enumerate(logs):
        r = oracle.Game(game_mode=mode, seed=1)
        r.reset()
        for k, e in enumerate(log):
            if e["type"] in want:
                raw[e["type"]] += 1
                a, _, dn = r.status()
                s = int(e["actor"])
                if (a >> s) & 1 and not dn:
                    v = r.peek()
                    want[e["type"]] += mjai.select_action_from_mjai(r.legal(s), e, None if v.drawn_tile < 0 else int(v.drawn_tile), sanma) is not None
                elif e["type"] == "hora" and not dn and e["actor"] != e["target"]:
                    j = k - 1
                    while log[j]["type"] == "dora":
                        j -= 1
                    if log[j]["type"] in ("kakan", "ankan") and log[j]["actor"] == e["target"] and log[k - 1]["type"] != "hora":
                        robbed.add((li, k))
                        want["hora"] += 1
            r.apply_event(e, replay=True)
    assert raw["hora"] > 20 and raw["reach"] > 20 and raw["pon"] > 10 and raw["daiminkan"] + raw["kakan"] + raw["ankan"] > 3, raw
    assert all(want[t] == raw[t] for t in types if t != "hora") and raw["hora"] - want["hora"] <= 4, (want, raw)
    rb = replay.ReplayBatch(logs, game_mode=mode, include_pass=False)
    games = [oracle.Game(game_mode=mode, seed=1) for _ in logs]
    for o in games:
        o.reset()
    cursor = [0] * len(logs)
    got = dict.fromkeys(types, 0)
    for smp in rb.samples():
        k = smp["index"]
        for j in range(len(smp["game"])):
            g, s = int(smp["game"][j]), int(smp["seat"][j])
            o = games[g]
            while cursor[g] < k:
                o.apply_event(logs[g][cursor[g]], replay=True)
                cursor[g] += 1
            ev = logs[g][k]
            got[ev["type"]] += 1
            v = o.peek()
            if (g, k) in robbed:     # the walker's observation of a robbed kan: [Ron on the kan tile, Pass] and nothing else in the mask
                t, tile, cons = abi.unpack_action(int(smp["action"][j]))
                kan = logs[g][k - 1] if logs[g][k - 1]["type"] != "dora" else logs[g][k - 2]
                assert t == abi.RON and s == int(ev["actor"]) and tile == abi.mjai_to_tid(kan["pai"] if kan["type"] == "kakan" else kan["consumed"][0])
                assert [abi.unpack_action(int(a))[0] for a in smp["legal"][j]] == [abi.RON, abi.PASS]
                ids = np.flatnonzero(smp["mask"][j])
                assert list(ids) == ([56, 58] if sanma else [79, 81]) and int(smp["action_id"][j]) == ids[0]
                assert smp["obs"][j].tobytes() == o.encode(s, sanma).tobytes(), (g, k, s)
                robbed.discard((g, k))
                continue
            sel = mjai.select_action_from_mjai(o.legal(s), ev, None if v.drawn_tile < 0 else int(v.drawn_tile), sanma)
            assert sel == int(smp["action"][j]), (g, k, ev)
            assert (smp["mask"][j] == np.asarray(o.mask(s))[: len(smp["mask"][j])]).all(), (g, k, ev)
            assert smp["obs"][j].tobytes() == o.encode(s, sanma).tobytes(), (g, k, s)
    assert got == want and not robbed, (got, want, robbed)


@pytest.mark.parametrize("mode,picks", [(2, (17, 22, 39)), (5, (5, 10, 17))])   # (found again in round 6: the policy key changed)
def test_the_ron_on_a_robbed_kakan_is_a_sample(mode, picks, tmp_path):
    """games in which a kakan is robbed (oracle, greedy policy, calls at 160 / 256 - seeds found by search): the
    reference's iterator yields the chankan Ron (replay/mod.rs:483-527 builds it from last_discard = the kakan tile,
    state/event_handler.rs:649-661; get_observation_for_replay pushes it into the seat's - empty - claims, state/mod.rs:265-325),
    so must ReplayBatch and Kyoku.steps(): every hora of these logs that is the first of its round is a decision."""
    import json

    from oracle import oracle
    from riichienv_amd import replay

    sanma = mode >= 3
    logs = []
    for g in picks:
        o = oracle.Game(game_mode=mode, seed=5000 + g)
        o.reset()
        for _ in range(2500):
            if o.status()[2]:
                break
            o.step([int(x) for x in o.greedy_actions(71, g, 160)])
        logs.append([json.loads(x) for x in o.log()])
    first_hora = [[k for k, e in enumerate(log) if e["type"] == "hora" and log[k - 1]["type"] != "hora"] for log in logs]
    chankan = [[k for k in ks if log[k]["actor"] != log[k]["target"] and
                [e for e in log[:k] if e["type"] != "dora"][-1]["type"] == "kakan"] for log, ks in zip(logs, first_hora)]
    assert all(len(c) >= 1 for c in chankan), chankan
    rb = replay.ReplayBatch(logs, game_mode=mode, include_pass=False)
    seen = [set() for _ in logs]
    for smp in rb.samples():
        for j in range(len(smp["game"])):
            g, k = int(smp["game"][j]), smp["index"]
            if logs[g][k]["type"] == "hora":
                seen[g].add(k)
                t, tile, _ = abi.unpack_action(int(smp["action"][j]))
                assert int(smp["seat"][j]) == logs[g][k]["actor"]
                if k in chankan[g]:
                    assert t == abi.RON and smp["mask"][j].sum() == 2 and len(smp["legal"][j]) == 2
    assert [sorted(s) for s in seen] == first_hora, (seen, first_hora)
    # the per-round iterator of the reference's dataset loop (LogKyoku.steps): the chankan is one of the winner's decisions
    path = tmp_path / "chankan.jsonl"
    path.write_text("\n".join(json.dumps(e) for e in logs[0]) + "\n")
    n_ron = 0
    for ky in replay.MjaiReplay.from_jsonl(str(path)).take_kyokus():
        for seat, obs, act in ky.steps(skip_single_action=False):
            n_ron += int(act.action_type) == abi.RON
    assert n_ron >= len([k for k in first_hora[0] if logs[0][k]["actor"] != logs[0][k]["target"]])
