#!/usr/bin/env python3
"""Encoder kernels in isolation (for rocprofv3): after a device rollout that brought 65 536 games into mid-game states,
time rmj_encode_device / rmj_encode_extended_device (only_active = 2: the acting seats' rows) with HIP events on the handle's
stream, 4P and 3P.  Prints one JSON object with the average launch duration, the acting seats and the algorithmic bytes
(74 x W x 4 resp. 215 x W x 4 per acting seat, docs/FEATURE_ENCODING.md:8-82)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from riichienv_amd import vecenv  # noqa: E402

if len(sys.argv) > 1:   # a variant build: riichienv_amd/libriichi_mi355x_<suffix>.so
    vecenv.LIB_PATH = vecenv.LIB_PATH.replace(".so", "_" + sys.argv[1] + ".so")
out = {}
for mode in (2, 5):
    w = 27 if mode >= 3 else 34
    env = vecenv.VecRiichiEnv(65536, game_mode=mode, seed=0, event_ring=64)
    env.reset()
    env.step_random(0xC0FFEE, 400, auto_reset=True)
    act, _, dn = env.status()
    acting = int(sum(bin(int(a)).count("1") for a, d in zip(act, dn) if not d))
    for ext in (False, True):
        ch = 215 if ext else 74
        buf = torch.zeros((65536, 4, ch, w), dtype=torch.float32, device="cuda:0")
        ms = env.bench_encode(buf.data_ptr(), 40, extended=ext, only_active=2)
        b = ch * w * 4 * acting
        out[f"{'k_encode_ext' if ext else 'k_encode'}_{'3p' if mode >= 3 else '4p'}"] = {
            "kernel_ms": ms, "acting_seats": acting, "bytes_per_launch": b, "achieved_GBps": b / (ms * 1e-3) / 1e9,
            "frac_of_8TBps": b / (ms * 1e-3) / 8e12}
        del buf
    # the same observations as one dense batch (rmj_encode_compact_device: slot scan + encoder)
    cap = acting + 64
    buf = torch.zeros((cap, 74, w), dtype=torch.float32, device="cuda:0")
    index = torch.zeros((cap,), dtype=torch.int32, device="cuda:0")
    count = torch.zeros((1,), dtype=torch.int32, device="cuda:0")
    ms = env.bench_encode_compact(buf.data_ptr(), index.data_ptr(), cap, count.data_ptr(), 40)
    assert int(count.item()) == acting
    b = 74 * w * 4 * acting
    out[f"k_encode_compact_{'3p' if mode >= 3 else '4p'}"] = {
        "kernel_ms": ms, "acting_seats": acting, "bytes_per_launch": b, "achieved_GBps": b / (ms * 1e-3) / 1e9,
        "frac_of_8TBps": b / (ms * 1e-3) / 8e12, "note": "k_obs_offsets + k_encode_base per repetition"}
    del buf
    # ... and with rows padded to a multiple of 256 B (TorchVecEnv's default layout since round 5): the same 74 x W floats per row
    rs = env.padded_row_stride()
    env.set_encode_row_stride(rs)
    buf = torch.zeros((cap, rs), dtype=torch.float32, device="cuda:0")
    ms = env.bench_encode_compact(buf.data_ptr(), index.data_ptr(), cap, count.data_ptr(), 40)
    assert int(count.item()) == acting
    out[f"k_encode_compact_padded_{'3p' if mode >= 3 else '4p'}"] = {
        "kernel_ms": ms, "acting_seats": acting, "bytes_per_launch": b, "row_stride_bytes": rs * 4, "achieved_GBps": b / (ms * 1e-3) / 1e9,
        "frac_of_8TBps": b / (ms * 1e-3) / 8e12, "note": "k_obs_offsets + k_encode_base per repetition; algorithmic bytes (74 x W x 4 per seat), the padding is not written"}
    env.set_encode_row_stride(0)
    del buf
    env.close()
print(json.dumps(out))
