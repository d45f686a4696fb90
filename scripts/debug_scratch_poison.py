"""Round 6: the harness that root-caused the k_step4_act_enc flake of journal r05 section 7 (docs/journal_r06.md section 1), kept as the way to
look at the one-launch trainer path (rmj_step_ids_encode_device) under hostile conditions.

Two identical environments: A steps through rmj_step_ids_encode_device (k_step4_act_enc, one launch), B through step + obs (k_step4<false>,
k_encode_base).  Every step the lists / masks / tensors of the two are compared.  Knobs:
  where = none | a | b | ab   fill the queue's private segment (scratch) with a pattern from a kernel of another library in front of A's launch /
                              B's launches (scripts/micro/poison_scratch.hip) - how "reads scratch it never wrote" was ruled out
  --filler N                  N do-nothing waves on another stream take the first wave slots of the SIMDs while A's launch runs
  --census                    do not stop at the first difference: re-synchronise A from B, count failing (step, game) pairs, show what the lists lack
  --hwid                      library built with -DRMJ_DEBUG_HWID=<blocks>: which SIMD / wave slot / register base the failing waves had
To see the failure again: build the library with
  -mllvm -disable-machine-licm -DRMJ_STEP4_ENC_WAVES=5 -DRMJ_ROW_BALLOT_SHIFT64=1 -DRMJ_DEBUG_HWID=4096
(scripts/lint_isa_last_vgpr.py flags that build: four 64-bit shifts by v87 in an 88-register kernel) and run
  RMJ_LIB_PATH=<that library> python scripts/debug_scratch_poison.py none --census --hwid --n 8192 --steps 6
- ~40 games per step publish lists with entries missing, all of them in waves that are not the first of their SIMD; -DRMJ_DEBUG_PAD_VGPR=95 (the same
instructions, 96 registers allocated) or RMJ_ROW_BALLOT_SHIFT64=0 (the shipped row ballot) make it clean.

usage: RMJ_LIB_PATH=... python scripts/debug_scratch_poison.py [none|a|b|ab] [--mode 2] [--steps 150] [--pattern 0xDEADBEEF] [--salt 0] [--n 4096] [--reps 3]"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from riichienv_amd.torch_env import TorchVecEnv  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("where", nargs="?", default="a")
ap.add_argument("--mode", type=int, default=2)
ap.add_argument("--steps", type=int, default=150)
ap.add_argument("--pattern", type=lambda s: int(s, 0), default=0xDEADBEEF)
ap.add_argument("--salt", type=lambda s: int(s, 0), default=0)
ap.add_argument("--n", type=int, default=4096)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--bytes", type=int, default=1024)
ap.add_argument("--filler", type=int, default=0, help="blocks of a do-nothing kernel started on another stream just before A's launch (they occupy the first wave slots)")
ap.add_argument("--hwid", action="store_true", help="library built with -DRMJ_DEBUG_HWID=<blocks>: where the failing waves of a launch ran")
ap.add_argument("--census", action="store_true", help="do not stop at the first difference: re-synchronise A from B and count (step, game) failures")
args = ap.parse_args()

P = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "micro", "libpoison_scratch.so"))
P.poison_scratch.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
P.filler.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64]
side = torch.cuda.Stream() if args.filler else None


def poison():
    rc = P.poison_scratch(C.c_void_p(torch.cuda.current_stream().cuda_stream), args.bytes, args.pattern & 0xFFFFFFFF, args.salt & 0xFFFFFFFF, 0)
    assert rc == 0, rc


total_bad = 0
census = []
for rep in range(args.reps):
    a = TorchVecEnv(args.n, game_mode=args.mode, seed=41 + rep, share_stream=True)
    b = TorchVecEnv(args.n, game_mode=args.mode, seed=41 + rep, share_stream=True)
    a.obs(only_active=True)
    b.obs(only_active=True)
    first = None
    for k in range(args.steps):
        ids = a.sample_ids(seed=k).clone()
        if "a" in args.where:
            poison()
        if args.filler:
            import time
            torch.cuda.synchronize()
            assert P.filler(C.c_void_p(side.cuda_stream), args.filler, 300000) == 0     # 3 ms
            time.sleep(0.0005)
        oa = a.step_obs(ids)
        if "b" in args.where:
            poison()
        b.step(ids)
        ob = b.obs(only_active=True)
        torch.cuda.synchronize()
        same = torch.equal(oa, ob) and torch.equal(a.mask, b.mask) and torch.equal(a.nlegal, b.nlegal)
        if not same:
            dm = (a.mask != b.mask).flatten(1).any(1).nonzero().flatten().tolist()
            do = (oa != ob).flatten(1).any(1).nonzero().flatten().tolist()
            dn = (a.nlegal != b.nlegal).any(1).nonzero().flatten().tolist()
            print(f"rep {rep} step {k}: A != B: masks of {len(dm)} games {dm[:6]}, tensors of {len(do)} games {do[:6]}, nlegal of {len(dn)} games {dn[:6]}")
            for g in dn[:3]:
                print(f"    game {g}: nlegal A {a.nlegal[g].tolist()} B {b.nlegal[g].tolist()} status A {int(a.status_raw[g]):#x} B {int(b.status_raw[g]):#x} ids {ids[g].tolist()}")
            first = k
            if not args.census:
                break
            import collections
            import numpy as np
            from tests.parity_util import fmt_action
            la, ca = a.env.legal(); lb, cb = b.env.legal()
            for g in dn:
                seat = int((ca[g] != cb[g]).nonzero()[0][0])
                A = [fmt_action(int(x)) for x in la[g, seat, : ca[g, seat]]]; B = [fmt_action(int(x)) for x in lb[g, seat, : cb[g, seat]]]
                miss = [i for i, x in enumerate(B) if x not in A]
                census.append((rep, k, g, seat, int(ca[g, seat]), int(cb[g, seat]), miss))
                if len(census) <= 6:
                    print(f"    game {g} seat {seat}: B list {B}\n        A list {A}\n        positions of B's entries missing in A: {miss}; mates' status A {[hex(int(a.status_raw[x])) for x in range(g // 4 * 4, g // 4 * 4 + 4)]}")
            if args.hwid:
                nb = args.n // 4
                hw = np.zeros((nb, 4), dtype=np.uint64)
                a.env.L.rmj_debug_hwid_fetch.argtypes = [C.c_void_p, C.c_uint32]
                assert a.env.L.rmj_debug_hwid_fetch(hw.ctypes.data, nb) == 0
                hwid = (hw[:, 2] & 0xFFFFFFFF).astype(np.int64); xcc = (hw[:, 2] >> 32).astype(np.int64) & 15
                gpr = (hw[:, 3] & 0xFFFFFFFF).astype(np.int64); lds = (hw[:, 3] >> 32).astype(np.int64)
                f = {"wave": hwid & 15, "simd": (hwid >> 4) & 3, "cu": (hwid >> 8) & 15, "sh": (hwid >> 12) & 1, "se": (hwid >> 13) & 7, "xcc": xcc,
                     "vgpr_base": gpr & 63, "sgpr_base": (gpr >> 16) & 63, "lds_base": lds & 255, "lds_size": (lds >> 12) & 511}
                t0 = hw[:, 0].astype(np.int64) - int(hw[:, 0].min()); t1 = hw[:, 1].astype(np.int64) - int(hw[:, 0].min())
                badb = sorted({g // 4 for g in dn})
                isbad = np.zeros(nb, dtype=bool); isbad[badb] = True
                print(f"    launch: {nb} blocks, start 0..{int(t0.max())} ticks, end {int(t1.min())}..{int(t1.max())} (100 MHz); failing blocks {badb[:12]}")
                for name, v in f.items():
                    allc = collections.Counter(v.tolist()); badc = collections.Counter(v[isbad].tolist())
                    print(f"      {name:9s} failing {sorted(badc.items())}  | all {sorted(allc.items()) if len(allc) <= 16 else str(len(allc)) + ' values'}")
                # waves that shared the failing waves' SIMD (same xcc, se, sh, cu, simd) and overlapped in time
                key = (f["xcc"] << 20) | (f["se"] << 16) | (f["sh"] << 12) | (f["cu"] << 4) | f["simd"]
                for bidx in badb[:8]:
                    mates = [int(x) for x in np.nonzero((key == key[bidx]))[0] if x != bidx]
                    cum = [int(x) for x in np.nonzero(((key >> 4) == (key[bidx] >> 4)))[0] if x != bidx]
                    print(f"      block {bidx}: t {int(t0[bidx])}..{int(t1[bidx])} xcc {int(f['xcc'][bidx])} se {int(f['se'][bidx])} cu {int(f['cu'][bidx])} simd {int(f['simd'][bidx])} wave {int(f['wave'][bidx])} lds_base {int(f['lds_base'][bidx])} vgpr_base {int(f['vgpr_base'][bidx])}; same SIMD: {[(m, int(t0[m]), int(t1[m]), int(f['wave'][m])) for m in mates]}; same CU: {[(m, int(f['simd'][m]), int(t0[m]), int(t1[m]), int(f['lds_base'][m])) for m in cum]}")
            bad = np.array(sorted(set(dm + do + dn)), dtype=np.uint32)
            a.env.copy_games(bad, b.env, bad)
            a.obs(only_active=True)
            oa2 = a._obs
            torch.cuda.synchronize()
            assert torch.equal(a.nlegal, b.nlegal) and torch.equal(a.mask, b.mask), "re-synchronisation failed"
    if first is None:
        print(f"rep {rep}: {args.steps} steps, A == B at every step (poison: {args.where}, pattern {args.pattern:#x}, salt {args.salt:#x})")
    else:
        total_bad += 1
if args.census and census:
    import collections
    print("failing (step, game) pairs:", len(census), "in", len({(c[0], c[1]) for c in census}), "of", args.reps * args.steps, "steps")
    print("  by game // 256:", sorted(collections.Counter(c[2] // 256 for c in census).items()))
    print("  by game % 4 (row):", sorted(collections.Counter(c[2] % 4 for c in census).items()))
    print("  by (n_A, n_B):", sorted(collections.Counter((c[4], c[5]) for c in census).items()))
    print("  missing positions:", sorted(collections.Counter(tuple(c[6]) for c in census).items(), key=lambda kv: -kv[1])[:12])
    print("  by step:", sorted(collections.Counter(c[1] for c in census).items())[:40])
print("RESULT", "FAIL" if total_bad else "CLEAN", f"{total_bad}/{args.reps} repetitions differed; lib {os.environ.get('RMJ_LIB_PATH', 'shipped')}; poison {args.where}")
