"""Pins the ORACLE's shanten / effective tiles / best ukeire on the REFERENCE's own tables: tests/golden/shanten_vectors.json
holds the answers of the nyanten lookup (riichienv-core/src/shanten.rs:163-261, 407-484 over src/data/nyanten_*.bin,
evaluated in the build container by scripts/gen_shanten_vectors.py) for 10^5 sampled 4P and 10^5 sampled 3P hands of every
len/3 class; the hands are re-created here by the same integer sampler (tests/shanten_sampler.py)."""
import json
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from oracle import oracle
from tests.shanten_sampler import sample_hand, sample_hands, sample_visible

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "shanten_vectors.json")) as f:
    GOLD = json.load(f)


def expected_shanten(tag):
    return np.frombuffer(GOLD[f"shanten_{tag}"].encode(), dtype=np.uint8).astype(np.int16) - ord("0") - 1


def _threads(fn, arrays, parts=8):
    n = len(arrays[0])
    cuts = [n * i // parts for i in range(parts + 1)]
    with ThreadPoolExecutor(parts) as ex:   # the ctypes calls release the GIL
        outs = list(ex.map(lambda k: fn(*[a[cuts[k]:cuts[k + 1]] for a in arrays]), range(parts)))
    return np.concatenate(outs)


def test_sampler_is_the_generators():
    assert GOLD["n_shanten"] == 100000 and len(GOLD["shanten_4p"]) == 100000 and len(GOLD["shanten_3p"]) == 100000
    h = sample_hands(GOLD["seed"], 64, False)
    assert h.shape == (64, 34) and h.max() <= 4
    sizes = sorted(set(int(x) for x in sample_hands(GOLD["seed"], 200, True).sum(axis=1)))
    assert sizes == [1, 2, 4, 5, 7, 8, 10, 11, 13, 14]     # every len/3 class, 3n+1 and 3n+2
    assert not sample_hands(GOLD["seed"], 500, True)[:, 1:8].any()   # sanma hands hold no 2m-8m


def test_oracle_shanten_equals_nyanten_4p_and_3p():
    for tag, sanma in (("4p", False), ("3p", True)):
        n = 40000   # the GPU test covers all 10^5; the oracle enumerates suit by suit (memoised), sized for the CPU suite
        hands = sample_hands(GOLD["seed"], n, sanma)
        got = _threads(lambda h: oracle.shanten(h, sanma=sanma), [hands])
        exp = expected_shanten(tag)[:n]
        bad = np.nonzero(got != exp)[0]
        assert bad.size == 0, (tag, bad[:5], hands[bad[:1]], got[bad[:5]], exp[bad[:5]])
        assert (exp == -1).sum() > 500 and (exp == 0).sum() > 5000     # complete and tenpai hands are well represented


def test_oracle_ukeire_equals_nyanten_4p_and_3p():
    for tag, sanma in (("4p", False), ("3p", True)):
        n = GOLD["n_ukeire"]
        hands = np.array([sample_hand(GOLD["seed"] + 1, i, sanma) for i in range(n)], dtype=np.uint8)
        vis = np.array([sample_visible(GOLD["seed"] + 1, i, hands[i]) for i in range(n)], dtype=np.uint8)
        eff_exp = np.array(GOLD[f"effective_tiles_{tag}"])
        uke_exp = np.array(GOLD[f"best_ukeire_{tag}"])
        k = eff_exp >= 0
        got = _threads(lambda h: oracle.effective_tiles(h, sanma=sanma), [hands[k]])
        assert (got == eff_exp[k]).all(), (tag, np.nonzero(got != eff_exp[k])[0][:5])
        k = uke_exp >= 0
        got = _threads(lambda h, v: oracle.best_ukeire(h, v, sanma=sanma), [hands[k], vis[k]])
        assert (got == uke_exp[k]).all(), (tag, np.nonzero(got != uke_exp[k])[0][:5])
        assert k.sum() > 800
