"""Four identical environments stepped side by side (two through rmj_step_ids_encode_device, two through step + encode): which one differs first, in which games,
and what its lists lack.  Written for the -disable-machine-licm investigation of round 5 (docs/journal_r05.md section 7); usage: python scripts/debug_first_launch.py [mode] [base|rev|allstep|allobs]"""
import sys, os
import numpy as np
import torch
from riichienv_amd.torch_env import TorchVecEnv
from tests.parity_util import diff_dict, normalize_view
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 5
VAR = sys.argv[2] if len(sys.argv) > 2 else "base"   # base: 0,1 step_obs then 2,3 step+obs | rev: 2,3 first | allstep: every env step+obs | allobs: every env step_obs
n = 4096
envs = [TorchVecEnv(n, game_mode=mode, seed=41, share_stream=True) for _ in range(4)]   # 0,1: step_obs ; 2,3: step + obs
obs = [e.obs(only_active=True) for e in envs]
prev = None
for k in range(150):
    before = [(envs[0].env.peek(g).hand_index, envs[0].env.peek(g).phase, envs[0].env.peek(g).active_mask) for g in range(3000, 3600)] if k < 4 else None
    pn = envs[0].nlegal.clone()
    ids = envs[0].sample_ids(seed=k).clone()
    o = []
    order = [2, 3, 0, 1] if VAR == "rev" else [0, 1, 2, 3]
    o = [None] * 4
    for i in order:
        e = envs[i]
        fused = (i < 2 and VAR != "allstep") or VAR == "allobs"
        if fused:
            o[i] = e.step_obs(ids).clone()
        else:
            e.step(ids)
            o[i] = e.obs(only_active=True).clone()
    for i, e in []:
        if i < 2:
            o.append(e.step_obs(ids).clone())
        else:
            e.step(ids)
            o.append(e.obs(only_active=True).clone())
    torch.cuda.synchronize()
    m = [e.mask.clone() for e in envs]
    bad = False
    for i in range(1, 4):
        if not torch.equal(m[0], m[i]) or not torch.equal(o[0], o[i]):
            dm = (m[0] != m[i]).flatten(1).any(1).nonzero().flatten().tolist()
            do = (o[0] != o[i]).flatten(1).any(1).nonzero().flatten().tolist()
            print(f"step {k}: env0 vs env{i}: mask differs in games {dm[:8]}, obs differs in games {do[:8]}")
            for g0 in sorted({x // 4 * 4 for x in (dm + do)[:6]}):
                for g in range(g0, g0 + 4):
                    v0, v1 = envs[0].env.peek(g), envs[i].env.peek(g)
                    print(f"   quad game {g}: nlegal before {pn[g].tolist()} env0 {envs[0].nlegal[g].tolist()} env{i} {envs[i].nlegal[g].tolist()} | hand_index {before[g-3000][0] if before else '?'} -> {v0.hand_index} phase {before[g-3000][1] if before else '?'} -> {v0.phase} active {before[g-3000][2] if before else '?'} -> {v0.active_mask} done {v0.is_done} ids {ids[g].tolist()} full_count {envs[0].env.total_full_path()}")
            from tests.parity_util import fmt_action
            for g in dm[:4]:
                la, ca = envs[0].env.legal(); lb, cb = envs[i].env.legal()
                for seat in range(4):
                    if ca[g, seat] != cb[g, seat]:
                        A = [fmt_action(int(x)) for x in la[g, seat, : ca[g, seat]]]; B = [fmt_action(int(x)) for x in lb[g, seat, : cb[g, seat]]]
                        v = envs[0].env.peek(g); P = v.players[seat]
                        print(f"   game {g} seat {seat}: env0 list {A}\n      env{i} list {B}\n      missing in env0: {[x for x in B if x not in A]} extra: {[x for x in A if x not in B]} | hand {list(P.hand[:P.hand_len])} drawn {v.drawn_tile} flags riichi {P.riichi_declared} stage {P.riichi_stage} forbidden {list(P.forbidden[:P.n_forbidden]) if hasattr(P, 'n_forbidden') else '?'} kita {getattr(P, 'n_kita', '?')}")
            for g in (dm + do)[:2]:
                d = diff_dict(normalize_view(envs[0].env.peek(g)), normalize_view(envs[i].env.peek(g)))
                print("   game", g, "state diff:", d[:6])
                wm = (m[0][g] != m[i][g]).nonzero()[:8].tolist()
                print("   mask bytes differing (seat, id):", wm, [int(m[0][g][tuple(x)]) for x in wm], [int(m[i][g][tuple(x)]) for x in wm], "active/phase", envs[0].env.peek(g).active_mask, envs[0].env.peek(g).phase, "nlegal", envs[0].nlegal[g].tolist(), envs[i].nlegal[g].tolist())
                if not d:
                    w = (o[0][g] != o[i][g]).nonzero()[:6].tolist()
                    print("   obs cells differing (seat, ch, col):", w, [float(o[0][g][tuple(x)]) for x in w], [float(o[i][g][tuple(x)]) for x in w])
            bad = True
    if bad:
        break
else:
    print("150 steps: all four environments equal")
