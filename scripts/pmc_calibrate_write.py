#!/usr/bin/env python3
"""WRITE_SIZE / FETCH_SIZE calibration (the guide calls WRITE_SIZE uncalibrated on gfx950): known amounts of HBM traffic from kernels whose
byte counts are not in doubt - a fill of a 1 GiB tensor (1 GiB written, nothing read), a copy of 1 GiB (1 GiB read + 1 GiB written), and the library's base
encoder writing its dense 7 992-B rows (rmj_encode_device, only_active = 0: every row, n x 4 x 7 992 B) and its rows padded to 256 B.

  run   (under rocprofv3 --pmc WRITE_SIZE FETCH_SIZE --output-format csv -d <dir> -- python3 scripts/pmc_calibrate_write.py run): issues the launches
  report <dir>: per kernel the mean counter value per launch, the bytes the launch is known to move, and the ratio (counter unit = KiB as printed by the tool)"""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GIB = 1 << 30


def run():
    import ctypes as C

    import torch

    from riichienv_amd import vecenv
    x = torch.empty(GIB // 4, dtype=torch.float32, device="cuda")
    y = torch.empty_like(x)
    torch.cuda.synchronize()
    for _ in range(6):
        x.fill_(1.0)
    for _ in range(6):
        y.copy_(x)
    torch.cuda.synchronize()
    n = 65536
    env = vecenv.VecRiichiEnv(n, game_mode=5, seed=0, event_ring=64)
    env.reset()
    env.step_random(1, 50, auto_reset=True)
    dense = torch.zeros((n, 4, 74 * 27), dtype=torch.float32, device="cuda")
    stride = env.padded_row_stride()
    padded = torch.zeros((n, 4, stride), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    for _ in range(6):
        vecenv._chk(env.L.rmj_encode_device(env.h, 0, C.c_void_p(dense.data_ptr())))
    env.sync()
    env.set_encode_row_stride(stride)
    for _ in range(6):
        vecenv._chk(env.L.rmj_encode_device(env.h, 0, C.c_void_p(padded.data_ptr())))
    env.sync()
    print("issued: 6 fills and 6 copies of 1 GiB, 6 + 6 encoder launches over", n, "x 4 rows (dense, then stride", stride, "floats)")


def report(root):
    acc = {}
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            key = (row["Kernel_Name"], row["Counter_Name"], row.get("Dispatch_Id"))
            acc[key] = acc.get(key, 0.0) + float(row["Counter_Value"])
    per = {}
    for (k, c, _), v in acc.items():
        per.setdefault((k, c), []).append(v)
    rows_3p = 65536 * 4 * 74 * 27 * 4
    known = [("fill", "elementwise", "FillFunctor", GIB, 0), ("copy", "elementwise", "direct_copy", GIB, GIB), ("k_encode_base", "k_encode_base", "", rows_3p, 65536 * 640)]
    print("kernel (substring) | launches | WRITE_SIZE per launch [KiB as reported] | bytes written (known) | ratio reported / known | FETCH_SIZE x 2 per launch | bytes read (known)")
    for (k, c), vals in sorted(per.items()):
        if c != "WRITE_SIZE":
            continue
        fetch = per.get((k, "FETCH_SIZE"), [0.0])
        tag = next((t for t in known if t[1] in k and t[2] in k), None)
        kb = sum(vals) / len(vals)
        line = f"{k[:90]:90s} | {len(vals):3d} | {kb:14.1f} KiB = {kb * 1024 / 1e6:9.1f} MB"
        if tag:
            line += f" | {tag[3] / 1e6:9.1f} MB | {kb * 1024 / tag[3]:.3f} | {2 * sum(fetch) / len(fetch) * 1024 / 1e6:9.1f} MB | {tag[4] / 1e6:9.1f} MB"
        print(line)


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        report(sys.argv[2])
