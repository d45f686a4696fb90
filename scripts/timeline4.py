#!/usr/bin/env python3
"""Timeline of single launches of k_step4<false> (one step of 65 536 games per launch): where a wave's time goes between the
outer marks of the step (core cycles), when waves start and end on the 100 MHz clock, how long the waves live that take a
game through the full path, and which tier-0 exits (R4BAIL sites of rmj_step4.hip.h) sent them there.
Needs the timeline build: scripts/build_tl4.sh (-DRMJ_TL4 -> riichienv_amd/libriichi_mi355x_tl4.so); never the shipped library."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from riichienv_amd import vecenv  # noqa: E402

vecenv.LIB_PATH = os.path.join(ROOT, "riichienv_amd", "libriichi_mi355x_tl4.so")
FULL = ["load record", "step up to the exhaustive draw (discard, claims)", "accept_riichi + tenpai of the seats", "payments + ryukyoku event",
        "next-round decision + end_kyoku", "wall shuffle", "round reset, wall to HBM, deal, four sorts", "start_kyoku + tehai events", "list: waits + tsumo check",
        "first draw", "list: kyushu / kita + rest", "(step_game returns)", "masks, lists, status", "record store", "list: discards + riichi probe", "list: kans"]
NAMES = ["load records", "policy", "apply + claims", "drawer's list", "publication", "record store", "bailed games (full path)"]


def main():
    games = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    mode = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    os.environ["RMJ_STEP4"] = "1"          # a launch per step
    os.environ["RMJ_STEP_STREAMS"] = "1"
    L = vecenv.load_lib()
    waves = (games + 3) // 4
    waves += waves // 4          # (heavy-first order: the front blocks)
    buf = np.zeros((waves, 32), dtype=np.uint64)
    L.rmj_tl4_fetch(buf.ctypes.data_as(C.c_void_p), waves)       # allocates the rows before any kernel runs
    env = vecenv.VecRiichiEnv(games, game_mode=mode, seed=0)
    env.reset()
    env.step_random(0xC0FFEE, 600, auto_reset=True)
    env.step_random(0xC0FFEE, 1, auto_reset=True)
    L.rmj_tl4_fetch(buf.ctypes.data_as(C.c_void_p), waves)       # (clears the rows the warm-up rollout wrote)
    n = 40
    sec = np.zeros(7)
    acc = {k: 0.0 for k in ("last_start", "last_end", "last_end_no_bail", "life", "life_bail_p50", "life_bail_p90", "life_bail_max",
                            "life_other_p50", "life_other_max", "bail_waves", "multi_bail_waves", "bail_rows",
                            "round_end_waves", "life_round_end_p50", "life_round_end_p90", "life_round_end_max", "last_end_plain")}
    alive = np.zeros(12)
    sites = {}
    fsec = np.zeros(16)
    fgames = 0
    for _ in range(n):
        env.step_random(0xC0FFEE, 1, auto_reset=True)
        L.rmj_tl4_fetch(buf.ctypes.data_as(C.c_void_p), waves)
        sec += buf[:, :7].astype(np.float64).mean(axis=0)
        ran = buf[:, 9] > 0          # blocks that served a unit (heavy-first order: the others left at once)
        buf_all, buf = buf, buf[ran]
        t0, t1 = buf[:, 8].astype(np.int64), buf[:, 9].astype(np.int64)
        z = t0.min()
        life = (t1 - t0) / 100.0
        b = buf[:, 7] > 0
        acc["last_start"] += (t0.max() - z) / 100.0
        acc["last_end"] += (t1.max() - z) / 100.0
        acc["last_end_no_bail"] += (t1[~b].max() - z) / 100.0
        acc["life"] += life.mean()
        p = np.percentile(life[b], [50, 90, 100]) if b.any() else [0.0, 0.0, 0.0]
        acc["life_bail_p50"] += p[0]; acc["life_bail_p90"] += p[1]; acc["life_bail_max"] += p[2]
        re_ = buf[:, 15] > 0                                           # waves that ended rounds in row form (r4_round_end + pass 2)
        acc["round_end_waves"] += int(re_.sum())
        pr = np.percentile(life[re_], [50, 90, 100]) if re_.any() else [0.0, 0.0, 0.0]
        acc["life_round_end_p50"] += pr[0]; acc["life_round_end_p90"] += pr[1]; acc["life_round_end_max"] += pr[2]
        plain = ~b & ~re_
        acc["last_end_plain"] += (t1[plain].max() - z) / 100.0
        p = np.percentile(life[~b & ~(buf[:, 15] > 0)], [50, 100])
        acc["life_other_p50"] += p[0]; acc["life_other_max"] += p[1]
        acc["bail_waves"] += int(b.sum())
        acc["multi_bail_waves"] += int((buf[:, 7] > 1).sum())
        acc["bail_rows"] += int(buf[:, 7].sum())
        if len(sys.argv) > 3 and sys.argv[3] == "detail" and _ < 6:   # the waves that end last: when they started, what they were
            order = np.argsort(t1)[::-1][:8]
            heavy_front = waves - (games + 3) // 4
            ids = np.nonzero(ran)[0]
            print("launch %d: kernel span %.1f us" % (_, (t1.max() - z) / 100.0), file=sys.stderr)
            for w in order:
                print("   block %5d (%s) start %5.1f end %5.1f life %5.1f  round_end %d (reasons %s) bail %d  sections(core cycles) %s" % (
                    ids[w], "front" if ids[w] < heavy_front else "in place", (t0[w] - z) / 100.0, (t1[w] - z) / 100.0, life[w], int(buf[w, 15]), [k for k in range(1, 8) if (int(buf[w, 14]) >> k) & 1], int(buf[w, 7]),
                    [int(x) for x in buf[w, :7]]), file=sys.stderr)
            late = re_ & ((t0 - z) / 100.0 > 25.0)
            for k in range(1, 8):
                hit = re_ & (((buf[:, 14].astype(np.int64) >> k) & 1) > 0)
                print("   reason %d: %d waves, %d of them in place, %d started after 25 us" % (k, int(hit.sum()), int((hit & (ids >= heavy_front)).sum()), int((hit & late).sum())), file=sys.stderr)
            print("   round-end waves %d, of them started after 25 us: %d (in place: %d)" % (int(re_.sum()), int(late.sum()), int((late & (ids >= heavy_front)).sum())), file=sys.stderr)
        edges = np.linspace(z, t1.max(), 13)
        for k in range(12):
            mid = 0.5 * (edges[k] + edges[k + 1])
            alive[k] += ((t0 <= mid) & (t1 > mid)).sum()
        ex = (buf[:, 10:14] == 8).any(axis=1) & (buf[:, 7] == 1)      # waves whose ONE full-path game was an exhaustive draw (site 7)
        fsec += buf[ex, 16:32].astype(np.float64).sum(axis=0)
        fgames += int(ex.sum())
        why = buf[:, 10:14].astype(np.int64).ravel()
        for s_, c_ in zip(*np.unique(why[why > 0] - 1, return_counts=True)):
            sites[int(s_)] = sites.get(int(s_), 0) + int(c_)
        buf = buf_all
    out = {"waves_per_launch": waves, "launches": n,
           "sections_core_cycles_per_wave": {name: sec[k] / n for k, name in enumerate(NAMES)},
           "us": {k: v / n for k, v in acc.items() if k not in ("bail_waves", "multi_bail_waves", "bail_rows", "round_end_waves")},
           "waves_that_end_rounds_in_row_form_per_launch": acc["round_end_waves"] / n,
           "full_path_games_per_launch": acc["bail_rows"] / n, "waves_with_a_full_path_game_per_launch": acc["bail_waves"] / n, "of_those_with_two_or_more": acc["multi_bail_waves"] / n,
           "waves_alive_in_twelfths_of_the_launch": [h / n for h in alive],
           "full_path_games_per_launch_by_tier0_exit": {str(k): v / n for k, v in sorted(sites.items())},
           "full_path_of_an_exhaustive_draw_core_cycles": {name: fsec[k] / max(fgames, 1) for k, name in enumerate(FULL)},
           "exhaustive_draw_games_sampled": fgames}
    r = env.bench_rollout(0xC0FFEE, 0, 200)
    out["instrumented_kernel_ms"] = r.step_kernel_ms
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
