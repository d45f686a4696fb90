"""GPU parity of the sequence features (rmj_encode_seq) against the Python restatement of observation/sequence_features.rs
(oracle/seq_features.py) fed with the oracle game's state and its seat log of the current round."""
from riichienv_amd.shard import game_seed
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _check(env_out, g, game, step, all_seats):
    from oracle import seq_features as sf

    a, _, done = game.status()
    full = sf.round_events(game.log(-1))
    want_prog = sf.progression(full, cap=256)
    n = int(env_out["n_progression"][g])
    assert n == len(want_prog), (step, g, n, len(want_prog))
    got = [tuple(int(x) for x in r) for r in env_out["progression"][g, :n]]
    assert got == want_prog, (step, g, [(i, x, y) for i, (x, y) in enumerate(zip(got, want_prog)) if x != y][:4])
    assert (env_out["progression"][g, n:] == np.array(sf.PROG_PAD, np.uint16)).all()
    for p in range(4):
        acts = bool((a >> p) & 1) and not done
        if not (acts or all_seats):
            continue
        ev = sf.round_events(game.log(p))
        obs = sf.observation_of(game, p)
        tok = sf.sparse(obs, ev, 1)
        ns = int(env_out["n_sparse"][g, p])
        assert [int(x) for x in env_out["sparse"][g, p, :ns]] == tok, (step, g, p, list(env_out["sparse"][g, p]), tok)
        assert (env_out["sparse"][g, p, ns:] == sf.SPARSE_PAD).all()
        assert [float(x) for x in env_out["numeric"][g, p]] == sf.numeric(obs, ev), (step, g, p)
        cand = sf.candidates(obs, ev, game.legal(p) if acts else [])
        nc = int(env_out["n_candidates"][g, p])
        got_c = [tuple(int(x) for x in r) for r in env_out["candidates"][g, p, :nc]]
        assert got_c == cand, (step, g, p, got_c, cand)
        assert (env_out["candidates"][g, p, nc:] == np.array(sf.CAND_PAD, np.uint16)).all()


@pytest.mark.parametrize("mode,seed", [(2, 11), (0, 100)])
def test_seq_features_along_rollout(mode, seed):
    from oracle import oracle
    from riichienv_amd import vecenv

    n, pseed = 12, 5
    env = vecenv.VecRiichiEnv(n, game_mode=mode, seed=seed, event_ring=1024)
    games = [oracle.Game(game_mode=mode, seed=game_seed(seed, g)) for g in range(n)]
    env.reset()
    for o in games:
        o.reset()
    kinds = set()
    for step in range(900):
        acts = np.array([games[g].random_actions(pseed, g) for g in range(n)], dtype=np.uint64)
        env.step(acts)
        for g in range(n):
            games[g].step([int(x) for x in acts[g]])
        if step % 9 == 0:
            out = env.encode_seq(1)
            for g in range(n):
                _check(out, g, games[g], step, all_seats=True)
                kinds |= {int(t) for t in out["progression"][g, : out["n_progression"][g], 1]}
    # the rollout exercised discards, chi, pon and at least one kind of kan
    assert any(1 <= t <= 37 for t in kinds) and any(38 <= t <= 127 for t in kinds) and any(128 <= t <= 167 for t in kinds)
    assert any(168 <= t <= 275 for t in kinds)


def test_seq_features_small_ring_and_sanma():
    from riichienv_amd import vecenv

    env = vecenv.VecRiichiEnv(4, game_mode=2, seed=1, event_ring=64)
    env.reset()
    env.step_random(7, 60, auto_reset=True)
    out = env.encode_seq(0)
    assert (out["n_progression"] == 0xFFFF).all()          # the round's start_kyoku is no longer in the ring (header)
    assert (out["sparse"][:, :, 0] == 0).all()              # game_style 0
    e3 = vecenv.VecRiichiEnv(2, game_mode=5, seed=1)
    e3.reset()
    with pytest.raises(vecenv.RmjError):
        e3.encode_seq()


def test_compat_observation_seq_features():
    """reference-named accessors (src/riichienv/_riichienv.pyi:429-449) on the scalar env: first observation of a game"""
    from riichienv_amd.compat import RiichiEnv

    env = RiichiEnv(game_mode="4p-red-half", seed=3)
    obs = env.reset()
    pid, o = next(iter(obs.items()))
    sp = np.frombuffer(o.encode_seq_sparse(game_style=1), np.uint16)
    assert len(sp) == 5 + 1 + 14 + 1 and sp[0] == 1 and sp[1] == 2 + pid and sp[4] == 13 + 69   # 136-14-(14+1) = 107 -> capped at 69
    assert 404 <= sp[-1] <= 440 and all(268 <= t <= 403 for t in sp[6:20])
    nu = np.frombuffer(o.encode_seq_numeric(), np.float32)
    assert list(nu) == [0, 0, 25000, 25000, 25000, 25000, 0, 0, 25000, 25000, 25000, 25000]
    pr = np.frombuffer(o.encode_seq_progression(), np.uint16).reshape(-1, 5)
    assert pr.tolist() == [[4, 0, 2, 2, 4]]
    ca = np.frombuffer(o.encode_seq_candidates(), np.uint16).reshape(-1, 4)
    assert len(ca) == len([a for a in o.legal_actions()]) and (ca[:, 3] == 3).all() and ca[:, 1].sum() <= 1  # 1 only if the drawn copy is the canonical one
    env3 = RiichiEnv(game_mode="3p-red-half", seed=3)
    o3 = next(iter(env3.reset().values()))
    with pytest.raises(AttributeError):
        o3.encode_seq_numeric()


@pytest.mark.parametrize("mode,seed,policy", [(2, 34, "random"), (0, 100, "random"), (2, 31, "greedy"), (1, 32, "greedy")])   # (seed 34: a RandomAgent game that declares riichi within the window - searched again in round 6, the policy key changed)
def test_seq_features_per_observation_delta(mode, seed, policy):
    """rmj_encode_seq_delta: the features over Observation.events as the reference's LIVE environment hands them out - the
    seat's log since its previous observation (state/mod.rs:211-218).  The harness keeps the oracle seats' cursors exactly
    like get_observation does (advanced for every acting seat after reset and after every step) and feeds the Python
    restatement of sequence_features.rs with that delta."""
    from oracle import oracle
    from oracle import seq_features as sf
    from riichienv_amd import vecenv

    n, pseed = 12, 5
    env = vecenv.VecRiichiEnv(n, game_mode=mode, seed=seed, event_ring=256)
    games = [oracle.Game(game_mode=mode, seed=game_seed(seed, g)) for g in range(n)]
    env.reset()
    cursor = [[0] * 4 for _ in range(n)]
    delta = [[[] for _ in range(4)] for _ in range(n)]

    def observe(g):
        a, _, dn = games[g].status()
        for p in range(4):
            if (a >> p) & 1 and not dn:
                log = games[g].log(p)
                delta[g][p] = log[cursor[g][p]:]
                cursor[g][p] = len(log)

    for g, o in enumerate(games):
        o.reset()
        observe(g)
    seen_reach_without_draw = seen_start = 0
    for step in range(700):
        # every fifth step, and whenever some acting seat's delta is just its own reach (the live-environment special case)
        after_reach = any(len(delta[g][p]) == 1 and '"reach"' in delta[g][p][0] and (games[g].status()[0] >> p) & 1
                          for g in range(n) for p in range(4))
        if step % 5 == 0 or after_reach:
            out = env.encode_seq_delta(1)
            for g, o in enumerate(games):
                a, _, dn = o.status()
                for p in range(4):
                    acts = bool((a >> p) & 1) and not dn
                    npg = int(out["n_progression"][g, p])
                    if not acts:
                        assert npg == 0 and int(out["n_candidates"][g, p]) == 0
                        continue
                    ev = delta[g][p]
                    want = sf.progression(ev, cap=64)
                    got = [tuple(int(x) for x in r) for r in out["progression"][g, p, :npg]]
                    assert got == want, (step, g, p, got, want)
                    assert (out["progression"][g, p, npg:] == np.array(sf.PROG_PAD, np.uint16)).all()
                    obs = sf.observation_of(o, p)
                    tok = sf.sparse(obs, ev, 1)
                    ns = int(out["n_sparse"][g, p])
                    assert [int(x) for x in out["sparse"][g, p, :ns]] == tok, (step, g, p)
                    assert [float(x) for x in out["numeric"][g, p]] == sf.numeric(obs, ev), (step, g, p)
                    cand = sf.candidates(obs, ev, o.legal(p))
                    nc = int(out["n_candidates"][g, p])
                    assert [tuple(int(x) for x in r) for r in out["candidates"][g, p, :nc]] == cand, (step, g, p)
                    # the live-environment effects the round-based variant cannot show
                    v = o.peek()
                    if v.drawn_tile >= 0 and v.current_player == p and sf.get_drawn_tile(ev, p) is None:
                        seen_reach_without_draw += 1
                    seen_start += any('"start_kyoku"' in s for s in ev)
        if policy == "greedy":   # play that declares riichi, calls kans and wins (orc_game_greedy_actions); finished games idle
            from riichienv_amd import abi
            acts = np.array([[abi.NO_ACTION] * 4 if games[g].status()[2] else [int(x) for x in games[g].greedy_actions(pseed, g, 64)]
                             for g in range(n)], dtype=np.uint64)
        else:
            acts = np.array([games[g].random_actions(pseed, g) for g in range(n)], dtype=np.uint64)
        env.step(acts)
        for g in range(n):
            if policy == "random" or not games[g].status()[2]:
                games[g].step([int(x) for x in acts[g]])
            observe(g)
    assert seen_start > 0
    if mode == 2:
        assert seen_reach_without_draw > 0   # after its own reach the seat's delta no longer holds the tsumo
