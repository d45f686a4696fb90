#!/usr/bin/env python3
"""What a device-policy rollout exercises, and an end-to-end consistency check over it (GPU only, no oracle): the MJAI logs the
step kernel wrote are read back by the MJAI reader, every win is reconstructed by WinResultContextIterator, evaluated in one
rmj_eval_hands batch, and the payments of the evaluation must be the deltas the step kernel booked (honba, pao and multiple Ron
accounted for).  Prints a census of event types, draw reasons, kinds of wins and yaku ids seen.
usage: python scripts/parity_coverage.py [games] [steps]"""
import collections
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from riichienv_amd import abi, vecenv  # noqa: E402
from riichienv_amd.replay import MjaiReplay, evaluate_win_contexts  # noqa: E402


def rollout_logs(mode, rule, n, steps, seed=99):
    env = vecenv.VecRiichiEnv(n, game_mode=mode, seed=seed, rule_bits=rule, event_ring=16384)
    env.reset()
    env.step_random(0xBEEF + mode, steps, auto_reset=False)
    cnt = env.event_counts()
    logs = [[json.loads(s) for s in env.mjai_log(g)] for g in range(n) if cnt[g] <= 16384]
    env.close()
    return logs


def census_logs(logs, label=""):
    """logs: lists of MJAI event dicts written by the device.  Every win is reconstructed, evaluated on the GPU and its
    payments compared with the logged deltas."""
    types, reasons, wins = collections.Counter(), collections.Counter(), collections.Counter()
    yaku = collections.Counter()
    ctx_all, expect = [], []
    with tempfile.TemporaryDirectory() as td:
        for g, log in enumerate(logs):
            for e in log:
                types[e["type"]] += 1
                if e["type"] == "ryukyoku":
                    r = e.get("reason", "")
                    reasons[r if not r.startswith("Error") else "illegal action"] += 1
            p = os.path.join(td, f"{g}.jsonl")
            with open(p, "w") as f:
                for e in log:
                    f.write(json.dumps(e) + "\n")
            for k in MjaiReplay.from_jsonl(p).take_kyokus():
                horas = [e for e in k.mjai_events if e.get("type") == "hora"]
                ctxs = list(k.take_win_result_contexts())
                if len(ctxs) != len(horas):
                    wins["context count mismatch"] += 1
                    continue
                if len(horas) > 1:
                    wins[f"{len(horas)}-fold ron"] += 1
                # A tsumo on the replacement draw of a kan / kita: the reference's iterator recognises it by the `doras` list of a
                # Mahjong Soul DealTile (replay/mod.rs:1806-1809) and drops the flag at any other draw (:1735-1736), so for MJAI
                # logs it never sets rinshan; the census marks those wins itself
                hist = [e for e in k.mjai_events if e.get("type") not in ("dora", "reach", "reach_accepted")]
                for i, (c, h) in enumerate(zip(ctxs, horas)):
                    at = next(j for j, e in enumerate(hist) if e is h)
                    if h["actor"] == h["target"] and at >= 2 and hist[at - 1]["type"] == "tsumo" and hist[at - 2]["type"] in ("ankan", "kakan", "daiminkan", "kita"):
                        c.conditions["rinshan"], c.conditions["haitei"] = True, False
                        wins["rinshan (marked by the census)"] += 1
                    # Two more places where the reference's MJAI reader and its own environment disagree (both mirrored by
                    # riichienv_amd.replay, both set right here so that the device's bookings can be checked):
                    # a kita before the riichi ends the first turn in the environment (state_3p/sanma.rs:39) but is no "call" for
                    # the reader's double-riichi flag (mjai_replay.rs:415, :606-611) ...
                    if c.conditions["double_riichi"]:
                        reach_at = next(j for j, e in enumerate(k.mjai_events) if e.get("type") == "reach" and e.get("actor") == c.seat)
                        # (up to the riichi DISCARD: a seat may declare, take a kita and only then discard)
                        reach_at = next(j for j, e in enumerate(k.mjai_events) if j > reach_at and e.get("type") == "dahai" and e.get("actor") == c.seat)
                        if any(e.get("type") == "kita" for e in k.mjai_events[:reach_at]):
                            c.conditions["double_riichi"] = False
                            wins["riichi after a kita (reader says double)"] += 1
                    # ... and a Ron on a kita finds no winning tile in a hora event without `pai` (mjai_replay.rs:541-559 has no
                    # BaBei case: tile 0)
                    if h["actor"] != h["target"] and at >= 1 and hist[at - 1]["type"] in ("kita", "hora") and h.get("pai") is None:
                        prev = next(e for e in reversed(hist[:at]) if e["type"] != "hora")
                        if prev["type"] == "kita":
                            north = abi.mjai_to_tid("N")
                            c.tiles = list(c.tiles[:-1]) + [north]
                            c.agari_tile = north
                            wins["ron on a kita"] += 1
                    # ... nor does a Ron on a kakan whose kan flushed the pending indicator of an earlier open kan: the `dora` event
                    # sits between the kakan and the hora, the reader's last action is Dora (tile 0, and no chankan)
                    if h["actor"] != h["target"] and h.get("pai") is None:
                        full = [e for e in k.mjai_events if e.get("type") not in ("reach", "reach_accepted")]
                        at_f = next(j for j, e in enumerate(full) if e is h)
                        before = [e for e in full[:at_f] if e["type"] != "hora"]
                        if len(before) >= 2 and before[-1]["type"] == "dora" and before[-2]["type"] == "kakan":
                            t = abi.mjai_to_tid(before[-2]["pai"])
                            c.tiles = list(c.tiles[:-1]) + [t]
                            c.agari_tile = t
                            c.conditions["chankan"] = True
                            wins["chankan behind a dora event"] += 1
                    ctx_all.append(c)
                    expect.append((k, h, i))
    evaluate_win_contexts(ctx_all)
    bad = 0
    for c, (k, h, i) in zip(ctx_all, expect):
        r, d, n_pl = c.actual, h["deltas"], len(h["deltas"])
        cond = c.conditions
        wins["tsumo" if cond["tsumo"] else "ron"] += 1
        for f in ("rinshan", "chankan", "haitei", "houtei", "double_riichi", "ippatsu", "tsumo_first_turn"):
            if cond[f]:
                wins[f] += 1
        for y in r.yaku[: r.n_yaku]:
            yaku[int(y)] += 1
        if r.yakuman:
            wins["yakuman"] += 1
        if not r.is_win:
            bad += 1
            wins["NOT A WIN"] += 1
            print("not a win", label, k.chang, k.ju, k.ben, c.seat, h, cond)
            continue
        paid = sorted(-x for x in d if x < 0)
        if cond["tsumo"]:
            oya = k.ju
            want = sorted((r.tsumo_agari_oya if s == oya else r.tsumo_agari_ko) + 100 * k.ben for s in range(n_pl) if s != c.seat)
            ok = paid == want
            if not ok and r.yakuman and sum(paid) == sum(want):
                wins["pao tsumo"] += 1
                ok = True
        else:
            hb = 100 * (n_pl - 1) * k.ben if i == 0 else 0
            ok = paid == [r.ron_agari + hb]
            if not ok and r.yakuman and sum(paid) == r.ron_agari + hb:
                wins["pao ron"] += 1
                ok = True
        if not ok:
            bad += 1
            wins["PAYMENT MISMATCH"] += 1
            print("mismatch", label, k.chang, k.ju, k.ben, c.seat, h, r.han, r.fu, list(r.yaku[: r.n_yaku]), r.ron_agari, r.tsumo_agari_oya, r.tsumo_agari_ko, cond)
    return dict(types=dict(types), reasons=dict(reasons), wins=dict(wins), yaku=dict(sorted(yaku.items())), checked=len(ctx_all), bad=bad)


def census(mode, rule, n, steps, seed=99):
    return census_logs(rollout_logs(mode, rule, n, steps, seed), f"mode {mode}")


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
    total_bad = 0
    for mode, rule, name in ((2, abi.RULE_TENHOU, "4p-red-half tenhou"), (2, abi.RULE_MJSOUL, "4p-red-half mjsoul"), (5, abi.RULE_MJSOUL, "3p-red-half mjsoul")):
        c = census(mode, rule, n, steps)
        total_bad += c["bad"]
        print(json.dumps({"workload": f"{n} games x {steps} steps, {name}", **c}))
    print("coverage ok" if total_bad == 0 else f"MISMATCHES: {total_bad}")
    sys.exit(1 if total_bad else 0)


if __name__ == "__main__":
    main()
