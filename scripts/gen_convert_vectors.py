#!/usr/bin/env python3
"""Build container only: imports the reference's pure-Python src/riichienv/convert.py by file path and writes the outputs of its
functions to tests/golden/convert_vectors.json (every id, every name, list and wall cases incl. the real wall string of the
reference's tests/env/test_paishan.py).  riichienv_amd/convert.py is checked against the file by tests/test_convert.py."""
import importlib.util
import json
import os
import random

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("ref_convert", "/root/reference/src/riichienv/convert.py")
cvt = importlib.util.module_from_spec(spec)
spec.loader.exec_module(cvt)

mpsz = [f"{n}{s}" for s in "mps" for n in range(10)] + [f"{n}z" for n in range(1, 8)]
mjai = [f"{n}{s}" for s in "mps" for n in range(1, 10)] + ["5mr", "5pr", "5sr"] + list("ESWNPFC")
rng = random.Random(7)
walls = []
real = ("3s9s1m5s9s3m9s4p7z1z3m6p3m3p5z1z2s7m5z2m7p3z7p7z5m5m5s6m6p4p3p4s7m7s4m6s9m5p5m6m3s2m3s9m3z4z4z1s8p4s7z8s1p1m9p"
        "9m8s4z6z2z1s4s2m3m8s3p1m7s8m2s1p2m6s1z9p3z8p6z5z2p2z2z1m7p4p7s6z6z6s5p8m9m3p2p3s7s7p6p2s9p6m1p5p1z6p2p4m7m5z9s2s4p5s0s4m3z8m1s"
        "2z6m7m0m6s1p8s8m8p4z1s0p9p4s4m2p7z8p")
walls.append(real)
for _ in range(3):
    ids = list(range(136))
    rng.shuffle(ids)
    walls.append("".join(cvt.tid_to_mpsz(t) for t in ids))
lists = []
for _ in range(20):
    ids = rng.sample(range(136), 14)
    lists.append(ids)


def errs(fn, args):
    out = []
    for a in args:
        try:
            fn(a)
            out.append(None)
        except ValueError:
            out.append("ValueError")
    return out


bad_mpsz = ["", "1x", "xm", "8z", "0z", "10m", "m"]
bad_mjai = ["5zr", "1mr", "X", "", "0p"]
data = {
    "tid_to_mpsz": [cvt.tid_to_mpsz(t) for t in range(136)],
    "tid_to_mjai": [cvt.tid_to_mjai(t) for t in range(136)],
    "mpsz_to_tid": {s: cvt.mpsz_to_tid(s) for s in mpsz},
    "mjai_to_tid": {s: cvt.mjai_to_tid(s) for s in mjai},
    "mpsz_to_mjai": {s: cvt.mpsz_to_mjai(s) for s in mpsz},
    "mjai_to_mpsz": {s: cvt.mjai_to_mpsz(s) for s in mjai},
    "lists": [{"tids": ids, "mpsz": cvt.tid_to_mpsz_list(ids), "mjai": cvt.tid_to_mjai_list(ids),
               "mpsz_back": cvt.mpsz_to_tid_list(cvt.tid_to_mpsz_list(ids)), "mjai_back": cvt.mjai_to_tid_list(cvt.tid_to_mjai_list(ids)),
               "mpsz_mjai": cvt.mpsz_to_mjai_list(cvt.tid_to_mpsz_list(ids)), "mjai_mpsz": cvt.mjai_to_mpsz_list(cvt.tid_to_mjai_list(ids))} for ids in lists],
    "walls": [{"paishan": w, "wall": cvt.paishan_to_wall(w)} for w in walls],
    "bad_mpsz": dict(zip(bad_mpsz, errs(cvt.mpsz_to_tid, bad_mpsz))),
    "bad_mjai": dict(zip(bad_mjai, errs(cvt.mjai_to_tid, bad_mjai))),
    "bad_tid": dict(zip(["-1", "136"], errs(cvt.tid_to_mpsz, [-1, 136]))),
}
with open(os.path.join(ROOT, "tests", "golden", "convert_vectors.json"), "w") as f:
    json.dump(data, f)
print("written", sum(len(v) for v in data.values()), "groups")
