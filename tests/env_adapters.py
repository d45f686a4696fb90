"""Uniform single-game adapters over the oracle and the HIP path (n=1 VecRiichiEnv), plus a DualEnv that
drives both and asserts bit-exact agreement after every call.  Test infrastructure."""
import json

import numpy as np

from riichienv_amd import abi
from tests.parity_util import diff_dict, fmt_action, normalize_view


class OracleEnv:
    def __init__(self, game_mode=0, seed=42, rule_bits=abi.RULE_TENHOU, round_wind=0):
        from oracle import oracle

        self.g = oracle.Game(game_mode=game_mode, seed=seed, rule_bits=rule_bits, round_wind=round_wind)

    def reset(self, **kw):
        self.g.reset(**kw)

    def step(self, acts):
        self.g.step(dict(acts))

    def legal(self, seat):
        return self.g.legal(seat)

    def mask(self, seat):
        return self.g.mask(seat)

    def waits(self, seat):
        return self.g.waits(seat)

    def status(self):
        return self.g.status()

    def peek(self):
        return self.g.peek()

    def poke(self, v):
        self.g.poke(v)

    def log(self, seat=-1):
        return self.g.log(seat)

    def scores(self):
        return [p.score for p in self.g.peek().players]

    def win_results(self):
        return self.g.win_results()

    def encode(self, seat):
        return self.g.encode(seat, self.g.peek().wall_len + sum(p.hand_len for p in self.g.peek().players) <= 108)


class GpuEnv:
    def __init__(self, game_mode=0, seed=42, rule_bits=abi.RULE_TENHOU, round_wind=0):
        from riichienv_amd import vecenv

        self.e = vecenv.VecRiichiEnv(1, game_mode=game_mode, seeds=np.array([seed], np.uint64), rule_bits=rule_bits,
                                     round_wind=round_wind, event_ring=4096,
                                     reference_rng=bool(rule_bits & abi.RULE_REFERENCE_RNG))   # the twin (oracle.Game) takes its definition of seed -> wall from rule_bits alone

    def reset(self, wall=None, oya=-1, round_wind=-1, scores=None, honba=-1, kyotaku=-1):
        self.e.reset(walls=None if wall is None else np.array(wall, np.uint8)[None],
                     oya=None if oya < 0 else [oya], round_wind=None if round_wind < 0 else [round_wind],
                     scores=None if scores is None else np.array(scores, np.int32)[None],
                     honba=None if honba < 0 else [honba], kyotaku=None if kyotaku < 0 else [kyotaku])

    def step(self, acts):
        a = np.full((1, 4), abi.NO_ACTION, np.uint64)
        for k, v in dict(acts).items():
            a[0, k] = v
        self.e.step(a)

    def legal(self, seat):
        l, c = self.e.legal()
        return [int(x) for x in l[0, seat, : c[0, seat]]]

    def mask(self, seat):
        return self.e.mask()[0, seat]

    def waits(self, seat):
        return int(self.e.waits()[0, seat])

    def status(self):
        a, p, d = self.e.status()
        return int(a[0]), int(p[0]), int(d[0])

    def peek(self):
        return self.e.peek(0)

    def poke(self, v):
        self.e.poke(0, v)

    def log(self, seat=-1):
        return self.e.mjai_log(0, seat)

    def scores(self):
        return [int(x) for x in self.e.scores()[0]]

    def win_results(self):
        return self.e.win_results(0)


class DualEnv:
    """Drives the oracle and the HIP path together; every mutation is followed by a full comparison."""

    def __init__(self, **kw):
        self.np = 3 if kw.get("game_mode", 0) >= 3 else 4
        self.o = OracleEnv(**kw)
        self.g = GpuEnv(**kw)
        self.check("ctor")

    def check(self, what):
        so, sg = self.o.status(), self.g.status()
        assert so == sg, (what, "status", so, sg)
        d = diff_dict(normalize_view(self.g.peek()), normalize_view(self.o.peek()))
        assert not d, (what, d[:10])
        act, ph, dn = so
        for s in range(4):
            if (act >> s) & 1 and not dn:
                lo, lg = self.o.legal(s), self.g.legal(s)
                assert lg == lo, (what, s, [fmt_action(a) for a in lg], [fmt_action(a) for a in lo])
                assert (np.asarray(self.g.mask(s)) == np.asarray(self.o.mask(s))).all(), (what, s, "mask")
                assert self.g.waits(s) == self.o.waits(s), (what, s, "waits")
        assert self.g.log() == self.o.log(), what
        assert self.g.win_results() == self.o.win_results(), (what, "win_results", self.g.win_results(), self.o.win_results())
        for s in range(self.np):
            assert self.g.log(s) == self.o.log(s), (what, s)

    def reset(self, **kw):
        self.o.reset(**kw)
        self.g.reset(**kw)
        self.check("reset")

    def step(self, acts):
        self.o.step(acts)
        self.g.step(acts)
        self.check(("step", {k: fmt_action(v) for k, v in dict(acts).items()}))

    def poke(self, v):
        self.o.poke(v)
        self.g.poke(v)
        self.check("poke")

    def legal(self, seat):
        return self.o.legal(seat)

    def mask(self, seat):
        return self.o.mask(seat)

    def waits(self, seat):
        return self.o.waits(seat)

    def status(self):
        return self.o.status()

    def peek(self):
        return self.o.peek()

    def log(self, seat=-1):
        return self.o.log(seat)

    def scores(self):
        return self.o.scores()

    def win_results(self):
        return self.o.win_results()


def events(env, seat=-1):
    return [json.loads(s) for s in env.log(seat)]
