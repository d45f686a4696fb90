#!/usr/bin/env python3
"""How fast can one row in four of a [n][4][row] float tensor be written, as a function of the row length / stride?  (DESIGN.md section 4.7:
the encoder's store pattern - the acting seat's row of every game - runs at half the rate of a dense fill.)  torch fills only."""
import time
import torch

n = 65536
dev = "cuda:0"


def rate(t, view, reps=30):
    view.fill_(1.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        view.fill_(2.0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    return view.numel() * 4 / dt / 1e12, dt * 1e6


for row in (1998, 2048, 2516, 2560, 4096):            # floats per row: 74 x 27, padded, 74 x 34, padded, 16 KB
    full = torch.zeros((n, 4, row), dtype=torch.float32, device=dev)
    seat = torch.randint(0, 4, (n,), device=dev)
    idx = torch.arange(n, device=dev)
    r_dense, us_dense = rate(full, full[: n // 4])                       # the same bytes, contiguous
    r_seat0, us0 = rate(full, full[:, 0, :])                            # row 0 of every game: regular stride
    # a random seat per game (the real pattern): index_put of ones
    src = torch.ones((n, row), dtype=torch.float32, device=dev)
    full[idx, seat] = src
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        full[idx, seat] = src
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 30
    print(f"row {row:5d} floats ({row * 4:6d} B, game stride {row * 16:6d} B): dense {r_dense:5.2f} TB/s ({us_dense:6.1f} us) | seat 0 of every game {r_seat0:5.2f} TB/s "
          f"({us0:6.1f} us) | random seat (index_put, reads src too) {n * row * 4 / dt / 1e12:5.2f} TB/s written ({dt * 1e6:6.1f} us)", flush=True)
    del full, src
