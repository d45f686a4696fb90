""""per-GPU handle/stream; host thread per GPU".  The header promises a thread-compatible library (one handle per
host thread): two handles driven CONCURRENTLY from two host threads (ctypes releases the GIL for the duration of every call) end
exactly where the same two call sequences end when they run one after the other."""
import threading

import numpy as np
import pytest

from riichienv_amd import vecenv

pytestmark = pytest.mark.gpu


def _drive(env, pseed, rounds, out, key):
    try:
        acc = 0
        for k in range(rounds):
            env.step_random(pseed, 7, auto_reset=True)            # fused rollout (one launch)
            acts = env.random_actions(pseed + 1)                   # device policy -> host -> validated host-buffer step
            env.step(acts)
            legal, cnt = env.legal()
            acc += int(cnt.sum()) + int(env.total_steps())
            if k % 5 == 0:
                env.mask(); env.waits(); env.scores()
        out[key] = acc
    except Exception as e:  # noqa: BLE001
        out[key] = e


def _snapshot(env):
    legal, cnt = env.legal()
    return (env.step_counts().copy(), env.scores().copy(), legal.copy(), cnt.copy(), env.mask().copy(), env.waits().copy(),
            [np.asarray(x).copy() for x in env.status()], env.event_counts().copy(), [env.mjai_log(g) for g in (0, 1, env.n - 1)])


def _equal(a, b):
    for x, y in zip(a[:4] + a[4:6], b[:4] + b[4:6]):
        assert (x == y).all()
    assert all((x == y).all() for x, y in zip(a[6], b[6])) and (a[7] == b[7]).all() and a[8] == b[8]


@pytest.mark.parametrize("modes", [(2, 2), (2, 5)])
def test_two_handles_on_two_host_threads_equal_the_serial_runs(modes):
    n, rounds = 2048, 40
    mk = lambda i: vecenv.VecRiichiEnv(n, game_mode=modes[i], seed=900 + i, event_ring=4096)   # noqa: E731
    serial, res = [mk(0), mk(1)], {}
    for i, e in enumerate(serial):
        e.reset()
        _drive(e, 50 + i, rounds, res, ("serial", i))
    conc = [mk(0), mk(1)]
    for e in conc:
        e.reset()
    ths = [threading.Thread(target=_drive, args=(conc[i], 50 + i, rounds, res, ("conc", i))) for i in range(2)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    for i in range(2):
        assert not isinstance(res[("conc", i)], Exception), res[("conc", i)]
        assert res[("conc", i)] == res[("serial", i)]
        _equal(_snapshot(serial[i]), _snapshot(conc[i]))
    for e in serial + conc:
        e.close()
