import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
x = torch.empty((65536, 4, 74 * 34), dtype=torch.float32, device="cuda:0")
ms = t(lambda: x.zero_()); print("dense zero_ 2.6 GB", ms, x.numel() * 4 / ms / 1e6, "GB/s")
ms = t(lambda: x[:, 0, :].zero_()); print("seat-0 rows (10 KB of every 40 KB) zero_", ms, x.numel() / ms / 1e6, "GB/s")
ms = t(lambda: x[:, 0, :].fill_(1.0)); print("seat-0 rows fill_", ms, x.numel() / ms / 1e6, "GB/s")
y = torch.empty((65536, 74 * 34), dtype=torch.float32, device="cuda:0")
ms = t(lambda: y.zero_()); print("compact [B, 2516] zero_", ms, y.numel() * 4 / ms / 1e6, "GB/s")
from riichienv_amd import vecenv
for mode in (2,):
    env = vecenv.VecRiichiEnv(65536, game_mode=mode, seed=0, event_ring=64); env.reset(); env.step_random(0xC0FFEE, 400, auto_reset=True)
    for oa in (2, 0):
        ms = env.bench_encode(x.data_ptr(), 40, extended=False, only_active=oa)
        print("k_encode_base only_active =", oa, ms, "ms")
