"""Deterministic hand sampler shared by scripts/gen_shanten_vectors.py (which evaluates the hands with the reference's
nyanten tables in the build container) and the tests (which re-create the same hands and compare the oracle and the
HIP kernels with the committed answers, tests/golden/shanten_vectors.json).  Pure integer arithmetic (splitmix64): the
hands do not depend on a numpy / Python version.  A fixture therefore only needs to carry the expected numbers."""
import numpy as np

M64 = 0xFFFFFFFFFFFFFFFF
SIZES = [13, 14, 13, 14, 13, 14, 10, 11, 7, 8, 4, 5, 1, 2, 13, 14]   # hand sizes cycled over the cases (every len/3 class)
SANMA_TYPES = [0, 8] + list(range(9, 34))


class Rng:
    def __init__(self, seed):
        self.s = seed & M64

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & M64
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
        return z ^ (z >> 31)

    def below(self, n):
        return self.next() % n


def _draw(r, pool, k, counts):
    """k tiles without replacement from `pool` (a list of tile types, one entry per physical tile)."""
    pool = list(pool)
    for _ in range(min(k, len(pool))):
        j = r.below(len(pool))
        counts[pool[j]] += 1
        pool[j] = pool[-1]
        pool.pop()


def sample_hand(seed, i, sanma):
    """Case i: a 34-histogram with at most four copies per type.  Styles: 0 wall draw, 1 one suit dominant, 2 terminals and
    honors dominant (kokushi / chiitoi territory), 3 sets + pair with a few tiles swapped (low shanten, complete hands)."""
    r = Rng(seed * 1000003 + i)
    types = SANMA_TYPES if sanma else list(range(34))
    n = SIZES[i % len(SIZES)]
    style = (i // len(SIZES)) % 4
    c = [0] * 34
    wall = [t for t in types for _ in range(4)]
    if style == 0:
        _draw(r, wall, n, c)
    elif style == 1:
        suit = 1 + r.below(2) if sanma else r.below(3)
        own = [t for t in wall if 9 * suit <= t < 9 * suit + 9]
        k_own = min(n, n - r.below(4))
        _draw(r, own, k_own, c)
        rest = [t for t in wall if not (9 * suit <= t < 9 * suit + 9)]
        _draw(r, rest, n - sum(c), c)
    elif style == 2:
        yao = [t for t in wall if t >= 27 or t % 9 in (0, 8)]
        k_y = min(n, n - r.below(5))
        _draw(r, yao, k_y, c)
        rest = [t for t in wall if not (t >= 27 or t % 9 in (0, 8))]
        _draw(r, rest, n - sum(c), c)
    else:
        # sets (triplet or run) + a pair up to n tiles, then swap up to three tiles for wall draws
        left = n
        guard = 0
        while left >= 3 and guard < 64:
            guard += 1
            if r.below(2):
                t = types[r.below(len(types))]
                if c[t] <= 1:
                    c[t] += 3
                    left -= 3
            else:
                s = (1 + r.below(2)) if sanma else r.below(3)
                a = 9 * s + r.below(7)
                if all(c[a + d] < 4 for d in range(3)):
                    for d in range(3):
                        c[a + d] += 1
                    left -= 3
        guard = 0
        while left >= 2 and guard < 64:
            guard += 1
            t = types[r.below(len(types))]
            if c[t] <= 2:
                c[t] += 2
                left -= 2
        swaps = r.below(4)
        held = [t for t in range(34) for _ in range(c[t])]
        for _ in range(min(swaps, len(held))):
            j = r.below(len(held))
            c[held[j]] -= 1
            held[j] = held[-1]
            held.pop()
        pool = [t for t in types for _ in range(4 - c[t])]
        _draw(r, pool, n - sum(c), c)
    assert sum(c) == n and max(c) <= 4, (i, c)
    return c


def sample_hands(seed, n, sanma):
    return np.array([sample_hand(seed, i, sanma) for i in range(n)], dtype=np.uint8)


def sample_visible(seed, i, hand):
    """A `visible` histogram for calculate_best_ukeire: up to 30 further tiles drawn from what the hand leaves."""
    r = Rng(seed * 7919 + i)
    pool = [t for t in range(34) for _ in range(4 - hand[t])]
    v = [0] * 34
    _draw(r, pool, r.below(31), v)
    return v
