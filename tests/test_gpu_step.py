"""GPU parity (the hot path): VecRiichiEnv through the C-ABI vs the oracle, game by game and
step by step — full state, ordered legal lists, masks, waits, MJAI event strings, scores."""
from riichienv_amd.shard import game_seed
import os
import numpy as np
import pytest

from riichienv_amd import abi
from tests.parity_util import diff_dict, fmt_action, normalize_view

pytestmark = pytest.mark.gpu


def _compare(env, games, g_ids, step_no, check_state=True):
    act, ph, dn = env.status()
    legal, cnt = env.legal()
    mask = env.mask()
    waits = env.waits()
    for g in g_ids:
        o = games[g]
        oa, op, od = o.status()
        assert dn[g] == od, (g, step_no, "done")
        assert act[g] == oa and ph[g] == op, (g, step_no, "active/phase", act[g], oa, ph[g], op)
        if check_state:
            d = diff_dict(normalize_view(env.peek(g)), normalize_view(o.peek()))
            assert not d, (g, step_no, d[:10])
        if od:
            assert cnt[g].sum() == 0
            continue
        for s in range(4):
            if (oa >> s) & 1:
                ol = o.legal(s)
                gl = [int(x) for x in legal[g, s, : cnt[g, s]]]
                assert gl == ol, (g, step_no, s, [fmt_action(a) for a in gl], [fmt_action(a) for a in ol])
                assert (mask[g, s] == o.mask(s)).all(), (g, step_no, s)
                assert int(waits[g, s]) == o.waits(s), (g, step_no, s)
            else:
                assert cnt[g, s] == 0 and mask[g, s].sum() == 0


@pytest.mark.parametrize("mode,rule", [(2, abi.RULE_TENHOU), (1, abi.RULE_MJSOUL)])
def test_random_rollout_parity(mode, rule):
    """configs[1]-style: random-agent self-play, every step compared with the oracle."""
    from oracle import oracle
    from riichienv_amd import vecenv

    n, seed, pseed = 48, 1000, 77
    env = vecenv.VecRiichiEnv(n, game_mode=mode, seed=seed, rule_bits=rule, event_ring=8192)
    games = [oracle.Game(game_mode=mode, seed=game_seed(seed, g), rule_bits=rule) for g in range(n)]
    _compare(env, games, range(n), -1)  # constructor state (shuffle #0)
    env.reset()
    for o in games:
        o.reset()
    _compare(env, games, range(n), 0)
    for step in range(1, 2500):
        acts = np.array([games[g].random_actions(pseed, g) for g in range(n)], dtype=np.uint64)
        dev = env.random_actions(pseed)
        assert (dev == acts).all(), step
        env.step(acts)
        for g in range(n):
            games[g].step([int(x) for x in acts[g]])
        # full state on a rotating subset (peek is a D2H copy per game), outputs on all games
        _compare(env, games, range(n), step, check_state=False)
        _compare(env, games, [step % n, (step * 7) % n], step, check_state=True)
        if all(o.status()[2] for o in games):
            break
    assert all(o.status()[2] for o in games)
    sc = env.scores()
    for g in range(n):
        _compare(env, games, [g], 99999)
        assert env.mjai_log(g) == games[g].log(), g
        for seat in range(4):
            assert env.mjai_log(g, seat) == games[g].log(seat), (g, seat)
        assert list(sc[g]) == [p.score for p in games[g].peek().players]


def test_step_random_device_policy_matches_oracle():
    """rmj_step_random (policy fused on device, auto-reset) against the oracle driven by the same policy."""
    from oracle import oracle
    from riichienv_amd import vecenv

    n, seed, pseed, steps = 32, 5, 4242, 1500
    env = vecenv.VecRiichiEnv(n, game_mode=0, seed=seed, event_ring=64)
    games = [oracle.Game(game_mode=0, seed=game_seed(seed, g)) for g in range(n)]
    env.reset()
    for o in games:
        o.reset()
    env.step_random(pseed, steps, auto_reset=True)
    for g, o in enumerate(games):
        for _ in range(steps):
            if o.status()[2]:
                o.reset()
                continue
            o.step(o.random_actions(pseed, g))
    _compare(env, games, range(n), steps)
    assert list(env.step_counts()) == [o.step_count for o in games]
    assert env.total_steps() == sum(o.step_count for o in games)


@pytest.mark.parametrize("mode,rule", [(2, abi.RULE_TENHOU), (5, abi.RULE_TENHOU), (1, abi.RULE_MJSOUL)])
def test_device_policy_long_rollout(mode, rule):
    """Several hanchan per game through the fused fast/full-path kernel (device policy, auto-reset): final state, legal
    lists, masks, waits, step counts and the whole MJAI log of the last game must equal the oracle's."""
    from oracle import oracle
    from riichienv_amd import vecenv

    n, seed, pseed, steps = 96, 900 + mode, 0xBEEF, 4000
    env = vecenv.VecRiichiEnv(n, game_mode=mode, seed=seed, rule_bits=rule, event_ring=4096)
    games = [oracle.Game(game_mode=mode, seed=game_seed(seed, g), rule_bits=rule) for g in range(n)]
    env.reset()
    for o in games:
        o.reset()
    env.step_random(pseed, steps, auto_reset=True)
    for g, o in enumerate(games):
        for _ in range(steps):
            if o.status()[2]:
                o.reset()
                continue
            o.step(o.random_actions(pseed, g))
    _compare(env, games, range(n), steps)
    assert list(env.step_counts()) == [o.step_count for o in games]
    for g in (0, n // 2, n - 1):
        assert env.mjai_log(g) == games[g].log(), g


@pytest.mark.parametrize("mode", [2, 5])
def test_encode_parity_along_rollout(mode):
    """Row A14: rmj_encode (74 x 34 f32 in 4P, 74 x 27 in 3P, every seat) bit-exact vs the oracle along a rollout."""
    from oracle import oracle
    from riichienv_amd import vecenv

    n, seed, pseed = 24, 321, 5
    sanma = mode >= 3
    env = vecenv.VecRiichiEnv(n, game_mode=mode, seed=seed)
    games = [oracle.Game(game_mode=mode, seed=game_seed(seed, g)) for g in range(n)]
    env.reset()
    for o in games:
        o.reset()
    for step in range(400):
        acts = np.array([games[g].random_actions(pseed, g) for g in range(n)], dtype=np.uint64)
        env.step(acts)
        for g in range(n):
            games[g].step([int(x) for x in acts[g]])
        if step % 7 == 0:
            enc = env.encode()
            act, ph, dn = env.status()
            enc_act = env.encode(only_active=True)
            for g in range(n):
                for s in range(3 if sanma else 4):
                    ref = games[g].encode(s, sanma)
                    assert enc[g, s].tobytes() == ref.tobytes(), (step, g, s, np.argwhere(enc[g, s] != ref)[:5])
                    if (act[g] >> s) & 1 and not dn[g]:
                        assert enc_act[g, s].tobytes() == ref.tobytes()
                    else:
                        assert not enc_act[g, s].any()


@pytest.mark.parametrize("mode", [2, 5])
def test_encode_extended_parity_along_rollout(mode):
    """Row N3: rmj_encode_extended (215 x 34 / 215 x 27 f32, every seat) bit-exact vs the oracle along a rollout
    (oracle shanten = enumeration, so fewer samples than the base encoder test)."""
    from oracle import oracle
    from riichienv_amd import vecenv

    n, seed, pseed = 8, 4321, 9
    sanma = mode >= 3
    env = vecenv.VecRiichiEnv(n, game_mode=mode, seed=seed)
    games = [oracle.Game(game_mode=mode, seed=game_seed(seed, g)) for g in range(n)]
    env.reset()
    for o in games:
        o.reset()
    for step in range(330):
        acts = np.array([games[g].random_actions(pseed, g) for g in range(n)], dtype=np.uint64)
        env.step(acts)
        for g in range(n):
            games[g].step([int(x) for x in acts[g]])
        if step % 82 == 0 or step == 329:
            enc = env.encode_extended()
            for g in range(n):
                for s in range(3 if sanma else 4):
                    ref = games[g].encode_extended(s)
                    bad = np.argwhere(enc[g, s] != ref)
                    assert enc[g, s].tobytes() == ref.tobytes(), (step, g, s, bad[:6], enc[g, s][tuple(bad[0])], ref[tuple(bad[0])])
                assert not enc[g, 3].any() or not sanma


@pytest.mark.parametrize("mode,rule", [(5, abi.RULE_TENHOU), (4, abi.RULE_MJSOUL)])
def test_sanma_random_rollout_parity(mode, rule):
    """configs[4]-style 3P games (108-tile wall, kita, no chi, 35000 start): every step compared with the oracle."""
    from oracle import oracle
    from riichienv_amd import vecenv

    n, seed, pseed = 48, 4000, 13
    env = vecenv.VecRiichiEnv(n, game_mode=mode, seed=seed, rule_bits=rule, event_ring=8192)
    games = [oracle.Game(game_mode=mode, seed=game_seed(seed, g), rule_bits=rule) for g in range(n)]
    _compare(env, games, range(n), -1)
    env.reset()
    for o in games:
        o.reset()
    _compare(env, games, range(n), 0)
    for step in range(1, 2500):
        acts = np.array([games[g].random_actions(pseed, g) for g in range(n)], dtype=np.uint64)
        assert (env.random_actions(pseed) == acts).all(), step
        env.step(acts)
        for g in range(n):
            games[g].step([int(x) for x in acts[g]])
        _compare(env, games, range(n), step, check_state=False)
        _compare(env, games, [step % n, (step * 7) % n], step, check_state=True)
        if all(o.status()[2] for o in games):
            break
    assert all(o.status()[2] for o in games)
    for g in range(n):
        _compare(env, games, [g], 99999)
        assert env.mjai_log(g) == games[g].log(), g
        for seat in range(3):
            assert env.mjai_log(g, seat) == games[g].log(seat), (g, seat)
    assert (env.ranks()[:, 3] == 0).all() and (np.sort(env.ranks()[:, :3], axis=1) == [1, 2, 3]).all()


@pytest.mark.parametrize("mode", [2, 5])
def test_shards_equal_whole_batch(mode):
    """Multi-GPU sharding on one device: two shard environments with game_offset 0 and B hold exactly
    the games of one 2B environment — seeds and policy keys are functions of the global game index."""
    from riichienv_amd import shard, vecenv

    B, K = 256, 600
    whole = vecenv.VecRiichiEnv(2 * B, game_mode=mode, seed=31, event_ring=4096)
    parts = [vecenv.VecRiichiEnv(B, game_mode=mode, seed=31, game_offset=shard.shard_offset(r, B), event_ring=4096)
             for r in range(2)]
    for e in [whole] + parts:
        e.reset()
        e.step_random(0xFEED, K, auto_reset=True)
    assert (np.concatenate([p.step_counts() for p in parts]) == whole.step_counts()).all()
    assert (np.concatenate([p.scores() for p in parts]) == whole.scores()).all()
    lw, cw = whole.legal()
    lp = np.concatenate([p.legal()[0] for p in parts])
    cp = np.concatenate([p.legal()[1] for p in parts])
    assert (cw == cp).all() and (lw == lp).all()
    assert whole.total_steps() == sum(p.total_steps() for p in parts)
    for g in (0, B - 1, B, 2 * B - 1):
        r, l = shard.owner_of(g, B)
        assert whole.mjai_log(g) == parts[r].mjai_log(l)


@pytest.mark.parametrize("mode", [2, 5])
def test_two_stream_rollout_equals_single_stream(mode):
    """rmj_step_random issues a multi-step rollout as ONE launch in which every wave steps its four games n_steps times
    (header); the result must be the one of stepping launch by launch on the handle's stream (n_steps = 1 is always a
    launch of its own)."""
    from riichienv_amd import vecenv

    B, K = 32768, 400
    a = vecenv.VecRiichiEnv(B, game_mode=mode, seed=5, event_ring=1024)
    b = vecenv.VecRiichiEnv(B, game_mode=mode, seed=5, event_ring=1024)
    a.reset()
    b.reset()
    a.step_random(0xBEEF, K, auto_reset=True)          # split
    for _ in range(K):
        b.step_random(0xBEEF, 1, auto_reset=True)      # one stream
    assert a.total_steps() == b.total_steps()
    assert (a.step_counts() == b.step_counts()).all()
    assert (a.scores() == b.scores()).all()
    assert (a.status()[0] == b.status()[0]).all()
    la, ca = a.legal()
    lb, cb = b.legal()
    live = np.arange(la.shape[-1])[None, None, :] < ca[:, :, None]      # (slab entries behind a seat's count are leftovers of earlier lists)
    assert (ca == cb).all() and (np.where(live, la, 0) == np.where(live, lb, 0)).all()
    assert (a.mask() == b.mask()).all()
    for g in (0, B // 4 - 1, B // 4, B // 2, 3 * B // 4 - 1, B - 1):
        assert a.mjai_log(g) == b.mjai_log(g)
    r = a.bench_rollout(0xBEEF, 0, 10)
    if os.environ.get("RMJ_STEP4", "2") == "2":                   # the default: the fused rollout, one launch, every wave loops
        assert r.launches == 1 and r.launches_in_flight == 1


@pytest.mark.parametrize("mode,policy", [(2, "random"), (5, "random"), (2, "greedy"), (5, "greedy")])
def test_noisy_host_actions_parity(mode, policy):
    """GameState::step validates what it is given (state/mod.rs:339-402): a seeded fraction of the host's actions is replaced
    by actions the reference rejects or ignores - a discard of a tile the seat does not hold, another seat's legal action, a
    Pass outside WaitResponse, an action from a seat that is not to act, a missing action - and every step (illegal-action
    penalty rounds included) must leave device and oracle in the same state."""
    from oracle import oracle
    from riichienv_amd import vecenv

    n, seed, pseed = 64, 31337 + mode, 99
    npl = 3 if mode >= 3 else 4
    env = vecenv.VecRiichiEnv(n, game_mode=mode, seed=seed, event_ring=8192)
    games = [oracle.Game(game_mode=mode, seed=game_seed(seed, g)) for g in range(n)]
    env.reset()
    for o in games:
        o.reset()
    rng = np.random.default_rng(seed)
    noisy = 0
    for step in range(1, 900):
        if policy == "greedy":   # the base play declares riichi, holds tenpai hands and wins: the noise then lands in those states
            acts = np.array([[abi.NO_ACTION] * 4 if games[g].status()[2] else [int(x) for x in games[g].greedy_actions(pseed, g, 64)]
                             for g in range(n)], dtype=np.uint64)
        else:
            acts = np.array([games[g].random_actions(pseed, g) for g in range(n)], dtype=np.uint64)
        for g in range(n):
            oa, _, od = games[g].status()
            if od or rng.random() > 0.02:
                continue
            s = int(rng.integers(npl))
            kind = int(rng.integers(5))
            if kind == 0:
                acts[g, s] = abi.pack_action(abi.DISCARD, int(rng.integers(136)))
            elif kind == 1:
                other = [a for p in range(npl) if p != s for a in games[g].legal(p)]
                if other:
                    acts[g, s] = other[int(rng.integers(len(other)))]
            elif kind == 2:
                acts[g, s] = abi.pack_action(abi.PASS)
            elif kind == 3:
                acts[g, s] = abi.NO_ACTION
            else:
                acts[g, s] = abi.pack_action(abi.RIICHI, int(rng.integers(136)))
            noisy += 1
        env.step(acts)
        for g in range(n):
            if not games[g].status()[2]:
                games[g].step([int(x) for x in acts[g]])
        _compare(env, games, range(n), step, check_state=False)
        _compare(env, games, [step % n, (step * 5) % n, (step * 11) % n], step, check_state=True)
        if all(o.status()[2] for o in games):
            break
    assert noisy > 100
    for g in range(n):
        _compare(env, games, [g], 99999)
        assert env.mjai_log(g) == games[g].log(), g


@pytest.mark.parametrize("mode,rule", [(2, abi.RULE_TENHOU), (5, abi.RULE_MJSOUL)])
def test_greedy_play_parity(mode, rule):
    """Parity where rounds END IN WINS: the oracle plays with the tenpai-seeking policy of tests/mjsoul_util (every win /
    riichi / kan / kita taken, shanten-greedy discards), the same actions go through rmj_step, and every step must leave device
    and oracle with the same status, legal lists, masks and waits (full state on a rotating sample and at the end, whole MJAI
    logs at the end).  The uniform RandomAgent wins about once in 250 rounds; here most rounds end with a Ron or a Tsumo."""
    import json

    from oracle import oracle
    from riichienv_amd import vecenv
    from tests.mjsoul_util import greedy_actions

    n, seed = 24, 4711 + mode
    sanma = mode >= 3
    env = vecenv.VecRiichiEnv(n, game_mode=mode, seed=seed, rule_bits=rule, event_ring=16384)
    games = [oracle.Game(game_mode=mode, seed=game_seed(seed, g), rule_bits=rule) for g in range(n)]
    env.reset()
    for o in games:
        o.reset()
    rng = np.random.default_rng(seed)
    for step in range(1, 700):
        acts = np.full((n, 4), abi.NO_ACTION, dtype=np.uint64)
        for g, o in enumerate(games):
            if not o.status()[2]:
                a = greedy_actions(o, rng, sanma)
                acts[g] = a
                o.step(a)
        env.step(acts)
        _compare(env, games, range(n), step, check_state=False)
        _compare(env, games, [step % n], step, check_state=True)
    _compare(env, games, range(n), -1, check_state=True)
    kinds = {}
    for g, o in enumerate(games):
        log = o.log()
        assert env.mjai_log(g) == log, g
        for s in log:
            t = json.loads(s)["type"]
            kinds[t] = kinds.get(t, 0) + 1
    assert kinds.get("hora", 0) >= 40 and kinds.get("reach_accepted", 0) >= 30, kinds


@pytest.mark.parametrize("n", [1, 2, 3, 5, 7, 9])
def test_ragged_batch_sizes(n):
    """Batches that do not fill their last wave (four games per wave): the fused rollout, single device-policy steps, host actions
    and the encoders all handle 1 .. 3 live rows in a quad; every game still equals its oracle."""
    from oracle import oracle
    from riichienv_amd import vecenv

    seed, pseed = 900 + n, 5
    for mode in (2, 5):
        env = vecenv.VecRiichiEnv(n, game_mode=mode, seed=seed, event_ring=8192)
        games = [oracle.Game(game_mode=mode, seed=game_seed(seed, g)) for g in range(n)]
        env.reset()
        for o in games:
            o.reset()
        def advance(k):
            for g, o in enumerate(games):
                for _ in range(k):
                    if o.status()[2]:
                        o.reset()
                    else:
                        o.step(o.random_actions(pseed, g))
        env.step_random(pseed, 200, auto_reset=True)          # one launch, every wave loops
        advance(200)
        _compare(env, games, range(n), 200)
        for _ in range(50):                                   # a launch per step
            env.step_random(pseed, 1, auto_reset=True)
        advance(50)
        _compare(env, games, range(n), 250)
        for step in range(40):                                # host actions
            acts = np.array([games[g].random_actions(pseed, g) for g in range(n)], dtype=np.uint64)
            env.step(acts)
            for g, o in enumerate(games):
                if not o.status()[2]:
                    o.step([int(x) for x in acts[g]])
        _compare(env, games, range(n), 290)
        enc = env.encode()
        for g, o in enumerate(games):
            act = o.status()[0]
            for s in range(3 if mode >= 3 else 4):
                if (act >> s) & 1:
                    assert enc[g, s].tobytes() == o.encode(s, mode >= 3).tobytes(), (n, mode, g, s)
            assert env.mjai_log(g) == o.log(), (n, mode, g)
        env.close()
