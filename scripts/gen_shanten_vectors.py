#!/usr/bin/env python3
"""Build container only (reads /root/reference, never shipped to the GPU box): evaluates the sampled hands of
tests/shanten_sampler.py with the REFERENCE's shanten tables and writes the answers to tests/golden/shanten_vectors.json.

What is read from the reference, at generation time: the five nyanten key blobs riichienv-core/src/data/nyanten_*.bin and
the two hash tables SHUPAI_TABLE / ZIPAI_TABLE (numeric constants of shanten.rs:6-153).  The lookup chain is restated
from shanten.rs:155-261 (4P) and :407-484 (3P), the derived quantities from :265-405 / :486-626.  Nothing of it is copied
into the repository: the fixture holds only the expected numbers, the hands are re-created by the sampler."""
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.shanten_sampler import SANMA_TYPES, sample_hand, sample_visible  # noqa: E402

REF = "/root/reference/riichienv-core/src"
SEED, N_SHANTEN, N_UKEIRE = 20261002, 100000, 3000


def load_tables():
    src = open(os.path.join(REF, "shanten.rs")).read()

    def table(name, rows):
        body = src[src.index(f"const {name}"):]
        body = body[body.index("= [") + 2:]
        body = body[: body.index("];")]
        nums = [int(x) for x in re.findall(r"\d+", re.sub(r"//[^\n]*", "", body))]
        assert len(nums) == rows * 15 * 5, (name, len(nums))
        return [[nums[(i * 15 + n) * 5:(i * 15 + n) * 5 + 5] for n in range(15)] for i in range(rows)]

    blobs = {}
    for k in ("shupai_keys", "zipai_keys", "keys1", "keys2", "keys3"):
        blobs[k] = open(os.path.join(REF, "data", f"nyanten_{k}.bin"), "rb").read()
    return table("SHUPAI_TABLE", 9), table("ZIPAI_TABLE", 7), blobs


SHUPAI, ZIPAI, B = load_tables()


def _hash(tab, tiles):
    n = h = 0
    for i, c in enumerate(tiles):
        n += c
        h += tab[i][n][c]
    return h


def calc_normal(t, m):  # shanten.rs:186-196
    k0_m = B["shupai_keys"][_hash(SHUPAI, t[0:9])]
    k0_p = B["shupai_keys"][_hash(SHUPAI, t[9:18])]
    k1 = B["keys1"][k0_m * 126 + k0_p]
    k0_s = B["shupai_keys"][_hash(SHUPAI, t[18:27])]
    k2 = B["keys2"][k1 * 126 + k0_s]
    k0_z = B["zipai_keys"][_hash(ZIPAI, t[27:34])]
    return B["keys3"][(k2 * 55 + k0_z) * 5 + m] - 1


def calc_chitoi(t, sanma):  # :198-211, 3P :437-454
    kinds = pairs = 0
    for i, c in enumerate(t):
        if sanma and 1 <= i <= 7:
            continue
        if c > 0:
            kinds += 1
            pairs += c >= 2
    return 7 - pairs + max(0, 7 - kinds) - 1


def calc_kokushi(t):  # :213-226
    term = [0, 8, 9, 17, 18, 26, 27, 28, 29, 30, 31, 32, 33]
    kinds = sum(1 for i in term if t[i] > 0)
    pair = any(t[i] >= 2 for i in term)
    return 14 - kinds - int(pair) - 1


def calc_normal_3p(t, m):  # :407-435
    t = list(t)
    mc = [t[0], t[8]]
    t[0] = t[8] = 0
    slot = 27
    for i, pos in enumerate((0, 8)):
        if mc[i] == 0:
            continue
        while slot < 34 and t[slot] != 0:
            slot += 1
        if slot < 34:
            t[slot] = mc[i]
            slot += 1
        else:
            t[pos] = mc[i]
    return calc_normal(t, m)


def shanten(t, sanma):  # :228-242 / :456-468 with len_div3 = tiles / 3 (:244-261, :470-484)
    m = sum(t) // 3
    s = calc_normal_3p(t, m) if sanma else calc_normal(t, m)
    if s <= 0 or m < 4:
        return s
    s = min(s, calc_chitoi(t, sanma))
    return min(s, calc_kokushi(t)) if s > 0 else s


def effective_tiles(t, sanma):  # :265-297 / :486-520
    cur = shanten(t, sanma)
    n = 0
    for ty in (SANMA_TYPES if sanma else range(34)):
        if t[ty] >= 4:
            continue
        t[ty] += 1
        n += shanten(t, sanma) < cur
        t[ty] -= 1
    return n


def effective_tiles_with_discard(t, sanma):  # :304-327 / :525-548
    if sum(t) % 3 == 1:
        return effective_tiles(t, sanma)
    cur = shanten(t, sanma)
    best = 0
    for ty in range(34):
        if t[ty] == 0:
            continue
        t[ty] -= 1
        if shanten(t, sanma) <= cur:
            best = max(best, effective_tiles(t, sanma))
        t[ty] += 1
    return best


def best_ukeire(t, vis, sanma):  # :331-405 / :552-626
    cur = shanten(t, sanma)
    best = 0
    for ty in range(34):
        if t[ty] == 0:
            continue
        t[ty] -= 1
        ns = shanten(t, sanma)
        if ns <= cur:
            u = 0
            for d in (SANMA_TYPES if sanma else range(34)):
                if t[d] >= 4:
                    continue
                t[d] += 1
                if shanten(t, sanma) < ns:
                    u += max(0, max(0, 4 - vis[d]) - (t[d] - 1))
                t[d] -= 1
            best = max(best, u)
        t[ty] += 1
    return best


def main():
    doc = {"what": "answers of the reference's nyanten lookup (riichienv-core/src/shanten.rs + data/nyanten_*.bin) for the hands "
                   "of tests/shanten_sampler.py; written by scripts/gen_shanten_vectors.py in the build container",
           "seed": SEED, "n_shanten": N_SHANTEN, "n_ukeire": N_UKEIRE,
           "blob_sha256": {k: hashlib.sha256(v).hexdigest() for k, v in B.items()}}
    for sanma in (False, True):
        tag = "3p" if sanma else "4p"
        digits = []
        for i in range(N_SHANTEN):
            s = shanten(sample_hand(SEED, i, sanma), sanma)
            assert -1 <= s <= 8
            digits.append(str(s + 1))
        doc[f"shanten_{tag}"] = "".join(digits)   # one digit per hand: shanten + 1
        eff, uke = [], []
        for i in range(N_UKEIRE):
            h = sample_hand(SEED + 1, i, sanma)
            if sum(h) % 3 == 0:
                eff.append(-1)
                uke.append(-1)
                continue
            eff.append(effective_tiles_with_discard(list(h), sanma))
            uke.append(best_ukeire(list(h), sample_visible(SEED + 1, i, h), sanma) if sum(h) % 3 == 2 else -1)
        doc[f"effective_tiles_{tag}"] = eff
        doc[f"best_ukeire_{tag}"] = uke
        print(tag, "shanten histogram:", {d: digits.count(str(d)) for d in range(10)}, file=sys.stderr)
    with open(os.path.join(ROOT, "tests", "golden", "shanten_vectors.json"), "w") as f:
        json.dump(doc, f, separators=(",", ":"))
    print("written", file=sys.stderr)


if __name__ == "__main__":
    main()
