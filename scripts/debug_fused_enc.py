"""Which games / seats / cells differ between rmj_step_random_encode (one launch) and step_random(1) + rmj_encode_device per step, and since which step"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from riichienv_amd import vecenv
mode, n, PSEED, SEED = int(sys.argv[1]) if len(sys.argv) > 1 else 5, int(sys.argv[2]) if len(sys.argv) > 2 else 32768, 0xC0FFEE, 4242
total = int(sys.argv[3]) if len(sys.argv) > 3 else 120
w = 27 if mode >= 3 else 34
from tests import test_gpu_fullsize as T
PSEED, SEED = T.PSEED, T.SEED
for rep in range(int(os.environ.get("REPS", "1"))):
  a = vecenv.VecRiichiEnv(n, game_mode=mode, seed=SEED + rep, event_ring=64); b = vecenv.VecRiichiEnv(n, game_mode=mode, seed=SEED + rep, event_ring=64)
  a.reset(); b.reset()
  oa = torch.zeros((n, 4, 74, w), dtype=torch.float32, device="cuda:0"); ob = torch.zeros_like(oa)
  done = 0
  for chunk in ([int(x) for x in os.environ.get("CHUNKS", "120").split(",")]):
      if done >= total: break
      a.step_random_encode(PSEED, chunk, oa.data_ptr(), auto_reset=True, only_active=2)
      for _ in range(chunk):
          b.step_random(PSEED, 1, auto_reset=True)
          vecenv._chk(b.L.rmj_encode_device(b.h, 2, C.c_void_p(ob.data_ptr())))
      a.L.rmj_sync(a.h); b.L.rmj_sync(b.h); done += chunk
      same_state = (a.step_counts() == b.step_counts()).all() and (a.scores() == b.scores()).all()
      d = (oa != ob).flatten(2).any(2)
      gs = d.any(1).nonzero().flatten().tolist()
      if gs or not same_state or os.environ.get("VERBOSE"): print(f"rep {rep} after {done} steps: states equal {bool(same_state)}; tensors differ in {len(gs)} games {gs[:10]}")
      for g in gs[:4]:
          va, vb = a.peek(g), b.peek(g)
          act_a, ph_a, dn_a = a.status(); 
          seats = d[g].nonzero().flatten().tolist()
          print(f"   game {g}: seats {seats} active_mask a {va.active_mask} b {vb.active_mask} phase {va.phase}/{vb.phase} done {va.is_done}/{vb.is_done} hand_index {va.hand_index}/{vb.hand_index} step_count {a.step_counts()[g]}/{b.step_counts()[g]}")
          for s in seats[:2]:
              cells = (oa[g, s] != ob[g, s]).nonzero()[:6].tolist()
              print(f"      seat {s}: {int((oa[g, s] != ob[g, s]).sum())} cells differ, first {cells}: a {[float(oa[g, s][tuple(c)]) for c in cells]} b {[float(ob[g, s][tuple(c)]) for c in cells]}; row all-zero a {bool((oa[g,s]==0).all())} b {bool((ob[g,s]==0).all())}")
      if gs: break
