"""Parity at the batch sizes BASELINE.json names (configs[1..4]): the batch runs through rmj_step_random exactly like
bench.py drives it (device policy, auto-reset, the default split into four range launches on four HIP streams), then
~256 sampled games - game 0, both sides of every part boundary, the last game, and a seeded random rest - are compared
with the oracle driven by the same policy key: full state, ordered legal lists, masks, waits, status, step counts and
the tail of the MJAI log that the event ring still holds (and, for the sanma feature configuration, the 74 x 27 tensors
that rmj_encode_device wrote).  Games are independent (state/mod.rs:330-1315 touches one GameState), so the oracle only
has to replay the sampled ones."""
import ctypes as C
import os

import numpy as np
import pytest

from riichienv_amd import abi
from riichienv_amd.shard import game_seed
from tests.parity_util import diff_dict, fmt_action, normalize_view

pytestmark = pytest.mark.gpu

SEED, PSEED = 20261002, 0xC0FFEE


def sample_games(n, k=256, parts=4):
    idx = {0, 1, n - 2, n - 1}
    for i in range(1, parts):
        b = n * i // parts
        idx |= {b - 1, b, b + 1}
    for i in range(1, 8):       # boundaries of an eight-way split as well (rollouts of other stream counts, 8-GPU shards)
        b = n * i // 8
        idx |= {b - 1, b}
    rng = np.random.default_rng(n)
    idx |= set(int(x) for x in rng.integers(0, n, size=k))
    return sorted(i for i in idx if 0 <= i < n)[: k + 40]


def oracle_replay(mode, rule, g_global, steps):
    from oracle import oracle

    o = oracle.Game(game_mode=mode, seed=game_seed(SEED, g_global), rule_bits=rule)
    o.reset()
    for _ in range(steps):
        if o.status()[2]:
            o.reset()
            continue
        o.step(o.random_actions(PSEED, g_global))
    return o


def device_log_tail(env, g, ring):
    """Formatted MJAI strings of the events of game g that are still in the ring (a start_kyoku whose first record has
    been overwritten is dropped together with its continuation records)."""
    total = int(env.event_counts()[g])
    first = max(0, total - ring)
    buf, n = env.events(g, first)
    i = 0
    while i < n and buf[i].type == abi.EV_TEHAI:
        i += 1
    out = []
    s = C.create_string_buffer(2048)
    while i < n:
        used = env.L.rmj_format_event(C.cast(C.byref(buf, i * C.sizeof(abi.Event)), C.POINTER(abi.Event)), n - i, -1, s, 2048)
        assert used > 0, (g, i, buf[i].type)
        out.append(s.value.decode())
        i += used
    return out


def compare_game(env, g, o, ring, tag):
    legal, cnt, mask, waits, act, ph, dn = env.peek_outputs(g)
    oa, op, od = o.status()
    assert (act, ph, dn) == (oa, op, od), (tag, g, "status", (act, ph, dn), (oa, op, od))
    d = diff_dict(normalize_view(env.peek(g)), normalize_view(o.peek()))
    assert not d, (tag, g, d[:10])
    for s in range(4):
        if (oa >> s) & 1 and not od:
            ol = o.legal(s)
            gl = [int(x) for x in legal[s, : cnt[s]]]
            assert gl == ol, (tag, g, s, [fmt_action(a) for a in gl], [fmt_action(a) for a in ol])
            assert (mask[s] == o.mask(s)).all(), (tag, g, s, "mask")
            assert int(waits[s]) == o.waits(s), (tag, g, s, "waits")
        else:
            assert cnt[s] == 0 and mask[s].sum() == 0, (tag, g, s, "inactive seat has outputs")
    tail = device_log_tail(env, g, ring)
    olog = o.log()
    assert len(tail) >= min(len(olog), ring // 3 - 2), (tag, g, len(tail), len(olog))
    assert tail == olog[len(olog) - len(tail):], (tag, g, "mjai log tail")


def run_config(mode, rule, n, steps, offset=0, ring=256):
    from riichienv_amd import vecenv

    env = vecenv.VecRiichiEnv(n, game_mode=mode, seed=SEED, rule_bits=rule, game_offset=offset, event_ring=ring)
    env.reset()
    env.step_random(PSEED, steps, auto_reset=True)   # >= 16 384 games: four range launches per step on four streams
    counts = env.step_counts()
    sample = sample_games(n)
    total = 0
    for g in sample:
        o = oracle_replay(mode, rule, offset + g, steps)
        compare_game(env, g, o, ring, (mode, n, offset))
        assert int(counts[g]) == o.step_count, (g, int(counts[g]), o.step_count)
        total += o.step_count
    assert total > 0
    return env, sample


FUSED = os.environ.get("RMJ_STEP4", "2") == "2"      # the default kernel choice; 0 / 1 pick the per-step kernels (same parity, more launches)


def test_4096_games_4p_red_single():
    """configs[1]: 4 096 parallel 4p-red-single games (one launch per step: below the split threshold); single-kyoku
    games end after ~100 steps, so every sampled game has been restarted several times."""
    env, sample = run_config(0, abi.RULE_TENHOU, 4096, 700)
    assert env.bench_rollout(PSEED, 0, 4).launches == (1 if FUSED else 4)


@pytest.mark.parametrize("offset", [0, 7 * 65536])
def test_65536_games_4p_red_half(offset):
    """configs[2] (the headline workload), and the last of the eight shards of configs[3] (global games 458 752 ...)."""
    env, sample = run_config(2, abi.RULE_TENHOU, 65536, 700, offset=offset)
    assert not FUSED or env.bench_rollout(PSEED, 0, 4).launches == 1     # the fused rollout


@pytest.mark.parametrize("mode", [2, 5])
def test_65536_games_under_the_greedy_policy(mode):
    """The headline batch size under the policy that plays to win (rmj_step_greedy as bench.py --policy greedy drives it: one launch of
    ticket chunks, auto-reset): sampled games against the oracle playing the same policy - state, lists, masks, waits, log tail."""
    from oracle import oracle
    from riichienv_amd import vecenv

    n, steps, rate, ring = 65536, 600, 64, 256
    env = vecenv.VecRiichiEnv(n, game_mode=mode, seed=SEED, rule_bits=abi.RULE_TENHOU, event_ring=ring)
    env.reset()
    env.step_greedy(PSEED, steps, auto_reset=True, call_rate_256=rate)
    counts = env.step_counts()
    wins = 0
    for g in sample_games(n, k=96):
        o = oracle.Game(game_mode=mode, seed=game_seed(SEED, g), rule_bits=abi.RULE_TENHOU)
        o.reset()
        for _ in range(steps):
            if o.status()[2]:
                o.reset()
                continue
            o.step(o.greedy_actions(PSEED, g, rate))
        compare_game(env, g, o, ring, ("greedy", mode))
        assert int(counts[g]) == o.step_count
        wins += sum('"hora"' in e for e in o.log())
    assert wins > 20          # the sampled games end their rounds with wins
    if FUSED:
        assert int(env.bench_rollout(PSEED, 0, 100).queued) == 1


def test_65536_games_4p_mjsoul_rules_single_stream_equals_split():
    """The same batch stepped on ONE stream must end in the same sampled states as the four-stream split (different rule
    set for breadth: Mahjong Soul yakuman / pao options)."""
    from riichienv_amd import vecenv

    n, steps = 65536, 400
    env = vecenv.VecRiichiEnv(n, game_mode=1, seed=SEED, rule_bits=abi.RULE_MJSOUL, event_ring=256)
    env.set_rollout_streams(1)
    env.reset()
    env.step_random(PSEED, steps, auto_reset=True)
    for g in sample_games(n, k=96):
        compare_game(env, g, oracle_replay(1, abi.RULE_MJSOUL, g, steps), 256, "1-stream")


def test_524288_games_4p_red_half():
    """configs[3] as ONE shard on one GPU: 524 288 games, four range launches of 131 072 games per step."""
    run_config(2, abi.RULE_TENHOU, 524288, 600, ring=64)


def test_65536_games_3p_with_feature_tensor():
    """configs[4]: 65 536 sanma games, every step followed by rmj_encode_device(only_active=2) into a resident
    [B, 4, 74, 27] tensor (bench.py --mode 5 --encode); the acting seats' rows of the sampled games must be the oracle's
    Observation.encode() of the final state, byte for byte."""
    import torch

    from riichienv_amd import vecenv

    n, steps, ring, mode = 65536, 500, 256, 5
    env = vecenv.VecRiichiEnv(n, game_mode=mode, seed=SEED, event_ring=ring)
    env.reset()
    obs = torch.zeros((n, 4, 74, 27), dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()   # (torch's fill and the library's own stream are not ordered otherwise)
    env.step_random(PSEED, steps - 50, auto_reset=True)           # split rollout
    for _ in range(50):                                            # then the bench's step + encode cadence
        env.step_random(PSEED, 1, auto_reset=True)
        vecenv._chk(env.L.rmj_encode_device(env.h, 2, C.c_void_p(obs.data_ptr())))
    env.L.rmj_sync(env.h)
    sample = sample_games(n)
    rows = obs[torch.tensor(sample, device="cuda:0")].cpu().numpy()
    checked = 0
    for k, g in enumerate(sample):
        o = oracle_replay(mode, abi.RULE_TENHOU, g, steps)
        compare_game(env, g, o, ring, "3p+encode")
        oa, _, od = o.status()
        for s in range(3):
            if (oa >> s) & 1 and not od:
                ref = o.encode(s, True)
                assert rows[k, s].tobytes() == ref.tobytes(), (g, s, np.argwhere(rows[k, s] != ref)[:5])
                checked += 1
    assert checked >= len(sample) // 2


@pytest.mark.parametrize("mode,n,fused", [(5, 32768, "1"), (2, 32768, "1"), (5, 4099, "1"), (5, 32768, "0")])
def test_fused_feature_rollout_equals_step_then_encode(mode, n, fused, monkeypatch):
    """rmj_step_random_encode - round 3: ONE launch in which every wave steps its four games and writes the rows of the seats that
    are to act (k_step4_enc, as tickets k_step4_queue_enc; RMJ_ENC_FUSED=0: four parts on four streams, step + encode launches per
    part) - must leave exactly the tensor and the states of the unfused loop rmj_step_random(1) + rmj_encode_device per step."""
    import torch

    from riichienv_amd import vecenv

    steps, w = 120, (27 if mode >= 3 else 34)
    monkeypatch.setenv("RMJ_ENC_FUSED", fused)
    a = vecenv.VecRiichiEnv(n, game_mode=mode, seed=SEED, event_ring=64)
    b = vecenv.VecRiichiEnv(n, game_mode=mode, seed=SEED, event_ring=64)
    a.reset()
    b.reset()
    oa = torch.zeros((n, 4, 74, w), dtype=torch.float32, device="cuda:0")
    ob = torch.zeros((n, 4, 74, w), dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()   # (the fills run on torch's stream, the rollout on the library's own: round 6 saw the 1 GB fill of `oa` overtake the first rows once)
    a.step_random_encode(PSEED, steps, oa.data_ptr(), auto_reset=True, only_active=2)
    for _ in range(steps):
        b.step_random(PSEED, 1, auto_reset=True)
        vecenv._chk(b.L.rmj_encode_device(b.h, 2, C.c_void_p(ob.data_ptr())))
    a.L.rmj_sync(a.h)
    b.L.rmj_sync(b.h)
    assert torch.equal(oa, ob)
    assert (a.step_counts() == b.step_counts()).all() and (a.scores() == b.scores()).all()
    assert float(oa.abs().sum()) > 0


def _rollout_with(env_vars, mode, n, steps, seed=4242, pseed=PSEED):
    """a device-policy rollout in an environment created under `env_vars` (the library reads its knobs at rmj_create)"""
    from riichienv_amd import vecenv

    old = {k: os.environ.get(k) for k in env_vars}
    os.environ.update(env_vars)
    try:
        env = vecenv.VecRiichiEnv(n, game_mode=mode, seed=seed, event_ring=64)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    env.reset()
    env.step_random(pseed, steps, auto_reset=True)
    return env


def _same_batch(a, b, n):
    assert (a.step_counts() == b.step_counts()).all() and (a.scores() == b.scores()).all()
    la, ca = a.legal()
    lb, cb = b.legal()
    live = np.arange(la.shape[-1])[None, None, :] < ca[:, :, None]      # (slab entries behind a seat's count are leftovers of earlier lists)
    assert (ca == cb).all() and (np.where(live, la, 0) == np.where(live, lb, 0)).all() and (a.mask() == b.mask()).all() and (a.waits() == b.waits()).all()
    assert all((x == y).all() for x, y in zip(a.status(), b.status())) and (a.event_counts() == b.event_counts()).all()
    for g in list(range(0, n, max(1, n // 64))) + [n - 1]:
        assert not diff_dict(normalize_view(a.peek(g)), normalize_view(b.peek(g))), g
        first = max(0, int(a.event_counts()[g]) - 60)            # the last 60 event records still in the ring, byte for byte
        (ea, na), (eb, nb) = a.events(g, first, 60), b.events(g, first, 60)
        assert na == nb and bytes(ea)[: na * C.sizeof(abi.Event)] == bytes(eb)[: nb * C.sizeof(abi.Event)], g


@pytest.mark.parametrize("mode,n", [(2, 16384), (5, 16384), (2, 4099), (0, 1001)])
def test_queued_rollout_equals_the_one_quad_per_wave_rollout(mode, n):
    """k_step4_queue (the rollout handed out as (quad, chunk) tickets to a grid that fits the chip once, per-XCD queues) against
    k_step4<true> (every wave keeps its quad for the whole rollout): same outputs for EVERY game, same records and log tails on a
    sample; batches with a ragged last quad; several chunk lengths incl. one that does not divide the rollout."""
    plain = _rollout_with({"RMJ_QUEUE_CHUNK": "0"}, mode, n, 333)
    for chunk in ("64", "50", "166"):
        queued = _rollout_with({"RMJ_QUEUE_CHUNK": chunk, "RMJ_QUEUE_FORCE": "1"}, mode, n, 333)
        _same_batch(plain, queued, n)
        queued.close()
    plain.close()


def test_queued_rollout_survives_xcds_without_blocks():
    """Where blocks run is the dispatcher's business: XCDs that receive no block of the queue kernel (simulated: their waves leave at
    once) have their quads stepped by k_step4_fixup.  Same results with two, and with all but one, of the eight XCDs missing."""
    n = 8192
    plain = _rollout_with({"RMJ_QUEUE_CHUNK": "0"}, 2, n, 200)
    for skip in ("0x24", "0xFE"):
        q = _rollout_with({"RMJ_QUEUE_CHUNK": "64", "RMJ_QUEUE_FORCE": "1", "RMJ_QUEUE_TEST_SKIP_XCDS": skip}, 2, n, 200)
        _same_batch(plain, q, n)
        q.close()
    plain.close()


@pytest.mark.parametrize("mode", [2, 5])
def test_default_rollout_runs_as_tickets_and_equals_the_plain_one(mode):
    """the default library on a batch of more than one chip-full of waves (32 768 games = 8 192 quads): a 400-step rollout runs as
    k_step4_queue (RmjBenchResult.queued) and leaves every game where the one-quad-per-wave rollout leaves it; so does a short
    rollout (the driver's 20-step bench run: tickets of 5 calls of the step function); fewer than 10 steps keep one quad per wave"""
    n = 32768
    plain = _rollout_with({"RMJ_QUEUE_CHUNK": "0"}, mode, n, 400)
    dflt = _rollout_with({}, mode, n, 400)
    _same_batch(plain, dflt, n)
    if FUSED:     # (RMJ_STEP4=0 / 1 select the per-step kernels: no fused rollout, no tickets)
        assert int(dflt.bench_rollout(PSEED, 0, 200).queued) == 1 and int(plain.bench_rollout(PSEED, 0, 200).queued) == 0
        assert int(dflt.bench_rollout(PSEED, 0, 100).queued) == 1 and int(dflt.bench_rollout(PSEED, 0, 20).queued) == 1
        assert int(dflt.bench_rollout(PSEED, 0, 15).queued) == 1 and int(dflt.bench_rollout(PSEED, 0, 9).queued) == 0
        for k in (100, 20, 15, 9):     # the plain environment catches up, then a 20- and a 17-step rollout on both
            plain.step_random(PSEED, k, auto_reset=True)
        for e in (plain, dflt):
            e.step_random(PSEED, 20, auto_reset=True)
            e.step_random(PSEED, 17, auto_reset=True)
        _same_batch(plain, dflt, n)
    plain.close()
    dflt.close()
