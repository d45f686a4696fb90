"""Edge cases of the C-ABI path: ragged batch sizes (grid tail), empty inputs, the event ring, argument errors."""
import ctypes as C

import numpy as np
import pytest

from riichienv_amd import abi

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [1, 3, 5, 63, 257])
def test_ragged_batch_sizes_equal_prefix_of_larger_batch(n):
    """4 games per block: every batch size must behave like the first n games of a bigger batch."""
    from riichienv_amd import vecenv

    big = vecenv.VecRiichiEnv(260, game_mode=2, seed=9)
    small = vecenv.VecRiichiEnv(n, game_mode=2, seed=9)
    for e in (big, small):
        e.reset()
        e.step_random(0xABC, 300, auto_reset=True)
    assert (small.step_counts() == big.step_counts()[:n]).all()
    assert (small.scores() == big.scores()[:n]).all()
    ls, cs = small.legal()
    lb, cb = big.legal()
    assert (cs == cb[:n]).all() and (ls == lb[:n]).all()
    assert (small.mask() == big.mask()[:n]).all()
    assert (small.encode() == big.encode()[:n]).all()


def test_empty_hand_math_batches():
    from riichienv_amd import vecenv

    assert len(vecenv.eval_hands([])) == 0
    assert vecenv.shanten(np.zeros((0, 34), np.uint8)).shape == (0,)
    assert vecenv.effective_tiles(np.zeros((0, 34), np.uint8)).shape == (0,)
    assert vecenv.best_ukeire(np.zeros((0, 34), np.uint8), np.zeros((0, 34), np.uint8)).shape == (0,)
    empty = np.zeros((1, 34), np.uint8)
    assert vecenv.shanten(empty)[0] == vecenv.shanten(empty, sanma=True)[0]          # 0 tiles: len_div3 = 0
    with pytest.raises(ValueError):
        vecenv.effective_tiles(empty)                                                # 3n hand: the reference asserts


def test_event_ring_wraps_and_reports_overwritten_range():
    from riichienv_amd import vecenv

    env = vecenv.VecRiichiEnv(4, game_mode=2, seed=3, event_ring=64)
    env.reset()
    env.step_random(5, 400, auto_reset=False)
    cnt = env.event_counts()
    assert (cnt > 64).all()
    with pytest.raises(vecenv.RmjError):
        env.events(0, first=0)                                   # already overwritten
    buf, n = env.events(0, first=int(cnt[0]) - 64)
    assert n == 64 and all(buf[i].type != abi.EV_NONE for i in range(n))
    big = vecenv.VecRiichiEnv(4, game_mode=2, seed=3, event_ring=4096)
    big.reset()
    big.step_random(5, 400, auto_reset=False)
    b2, n2 = big.events(0, first=int(cnt[0]) - 64)
    assert n2 == 64 and bytes(buf)[: 64 * 32] == bytes(b2)[: 64 * 32]      # the ring holds exactly the last records


def test_argument_errors_are_reported_not_crashed():
    from riichienv_amd import vecenv

    L = vecenv.load_lib()
    h = C.c_void_p()
    cfg = abi.Config()
    cfg.n_games = 0
    assert L.rmj_create(C.byref(cfg), C.byref(h)) != 0 and b"" != L.rmj_last_error()
    cfg.n_games = 4
    cfg.device = 99
    assert L.rmj_create(C.byref(cfg), C.byref(h)) != 0
    assert L.rmj_step(None, None) != 0 and L.rmj_get_legal(None, None, None) != 0
    env = vecenv.VecRiichiEnv(2, game_mode=0, seed=1)
    with pytest.raises(vecenv.RmjError):
        env.peek(2)                                              # game index out of range
    with pytest.raises(vecenv.RmjError):
        env.events(7)
    v = env.peek(0)
    v.players[0].hand_len = 15
    with pytest.raises(vecenv.RmjError):
        env.poke(0, v)                                           # player view out of range


def test_longest_legal_lists_fit():
    """A hand with three ankan options plus riichi-free discards: list lengths stay within RMJ_MAX_LEGAL and the lists
    equal the oracle's (DualEnv compares after every mutation)."""
    from tests.env_adapters import DualEnv
    from tests.scenarios import setup, tiles

    env = DualEnv(game_mode=2, seed=1)
    setup(env, hands=[tiles("1111m2222m3333m44m"), None, None, None], drawn_tile=None)
    env.check("many kans")
    n = len(env.g.legal(0))
    assert 14 <= n <= abi.MAX_LEGAL


@pytest.mark.parametrize("mode,n", [(2, 1000), (5, 777), (0, 3)])
def test_legal_compact_equals_the_list_slab(mode, n):
    """rmj_get_legal_compact: one row per seat that is to act, (game, seat) order, entries = that seat's ordered list - the same
    content as the [n][4][64] slab of rmj_get_legal, incl. finished games (no rows) and a ragged last block"""
    from riichienv_amd import vecenv

    env = vecenv.VecRiichiEnv(n, game_mode=mode, seed=77, event_ring=64)
    env.reset()
    for k in range(12):
        env.step_random(9, 37 if k else 1, auto_reset=(k % 3 != 2))
        act, ph, dn = env.status()
        legal, cnt = env.legal()
        idx, off, ent = env.legal_compact()
        rows = [(g, s) for g in range(n) for s in range(4) if (act[g] >> s) & 1 and not dn[g] and cnt[g, s] > 0]
        assert [int(x) for x in idx] == [g * 4 + s for g, s in rows]
        assert off[0] == 0 and int(off[-1]) == len(ent) == sum(int(cnt[g, s]) for g, s in rows)
        for i, (g, s) in enumerate(rows):
            assert (ent[off[i]: off[i + 1]] == legal[g, s, : cnt[g, s]]).all(), (k, g, s)
    env.close()
