#!/usr/bin/env python3
"""Generates tests/golden/spec_tables.json from the reference's SPECIFICATION TEXT (run in the build container, where
/root/reference exists; the fixture travels, the reference does not):

    docs/FEATURE_ENCODING.md            the 74 channels of Observation.encode(): index ranges, names, normalisers ("/ 24.0"),
                                        relative-seat order, the 21 yaku of encode_yaku_possibility, the shanten-efficiency
                                        normalisers, the decay example
    docs/SEQUENCE_FEATURE_ENCODING.md   sparse offsets, numeric indices, progression / candidate type ranges, tuple vocabularies,
                                        padding tuples, the MJAI event -> tuple table, the wrapper's constants

Every entry keeps the line of the document it was read from.  The tests that use the fixture (tests/test_oracle_spec_tables.py)
check the ORACLE's encoders against these numbers on real game states, so that the expectations are the reference's published
tables, parsed mechanically, not constants typed from memory.  A table is data (offsets, counts, divisors), not source text."""
import json
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "spec_tables.json")


def rows(lines, start):
    """the cells of the markdown table whose header row is at index `start` (0-based): list of (line number, [cells])"""
    out = []
    i = start + 2
    while i < len(lines) and lines[i].lstrip().startswith("|"):
        cells = [c.strip() for c in lines[i].strip().strip("|").split("|")]
        out.append((i + 1, cells))
        i += 1
    return out


def find_table(lines, *header_words, after=0):
    for i in range(after, len(lines)):
        ln = lines[i]
        if ln.lstrip().startswith("|") and all(w.lower() in ln.lower() for w in header_words):
            return i
    raise SystemExit(f"table with header {header_words} not found")


def strip_md(s):
    return re.sub(r"[*`]", "", s).strip()


def span(s):
    m = re.match(r"(\d+)\s*-\s*(\d+)$", strip_md(s))
    if m:
        return int(m.group(1)), int(m.group(2))
    v = int(strip_md(s))
    return v, v


def feature_encoding():
    path = os.path.join(REF, "docs", "FEATURE_ENCODING.md")
    lines = open(path).read().split("\n")
    chans = []
    i = 0
    while True:
        try:
            i = find_table(lines, "Channel Index", "Description", after=i)
        except SystemExit:
            break
        for ln, c in rows(lines, i):
            lo, hi = span(c[0])
            det = strip_md(c[2])
            norm = re.search(r"/\s*([0-9]+(?:\.[0-9]+)?)\s*\)", det)
            formula = re.search(r"\(([^()]*round_wind[^()]*)\)\s*/\s*([0-9]+(?:\.[0-9]+)?)", c[2].replace("`", ""))   # (keeps the `*` of the formula)
            chans.append({"lo": lo, "hi": hi, "name": strip_md(c[1]), "line": ln,
                          "divisor": float(formula.group(2)) if formula else (float(norm.group(1)) if norm else None),
                          "formula": formula.group(1).strip() if formula else None,
                          "relative_seat_order": "relative seat order" in det,
                          "broadcast": "broadcast" in det.lower()})
        i += 1
    total = [int(m.group(1)) for ln in lines for m in [re.search(r"\*\*Total Channels:\s*(\d+)\*\*", ln)] if m]
    covered = sorted(ch for c in chans for ch in range(c["lo"], c["hi"] + 1))
    assert covered == list(range(total[0])), ("the channel tables do not tile 0..%d" % (total[0] - 1), covered)
    yi = find_table(lines, "Index", "Yaku", "Detection Logic")
    yaku = []
    for ln, c in rows(lines, yi):
        lo, hi = span(c[0])
        yaku.append({"lo": lo, "hi": hi, "name": strip_md(c[1]), "logic": strip_md(c[2]), "line": ln})
    assert sorted(k for y in yaku for k in range(y["lo"], y["hi"] + 1)) == list(range(21))
    text = "\n".join(lines)
    shapes = {name: [int(x) for x in m.group(1).split(",")] for name, m in (
        ("encode_discard_history_decay", re.search(r"encode_discard_history_decay\(decay_rate=0\.2\)`\s*\n\s*\nReturns a \*\*\(([\d, ]+)\)\*\*", text)),
        ("encode_yaku_possibility", re.search(r"encode_yaku_possibility\(\)`\s*\n\s*\nReturns a \*\*\(([\d, ]+)\)\*\*", text)),
        ("encode_furiten_ron_possibility", re.search(r"encode_furiten_ron_possibility\(\)`\s*\n\s*\nReturns a \*\*\(([\d, ]+)\)\*\*", text)),
        ("encode_shanten_efficiency", re.search(r"encode_shanten_efficiency\(\)`\s*\n\s*\nReturns a \*\*\(([\d, ]+)\)\*\*", text)))}
    sh = {k: float(m.group(1)) for k, m in (
        ("shanten", re.search(r"\*\*Shanten \(normalized /(\d+)\)", text)), ("effective_tiles", re.search(r"\*\*Effective Tiles \(normalized /(\d+)\)", text)),
        ("best_ukeire", re.search(r"\*\*Best Ukeire \(normalized /(\d+)\)", text)), ("turn_progress", re.search(r"\*\*Turn Progress \(normalized /(\d+)\)", text)))}
    unknown = float(re.search(r"are set to ([0-9.]+) \(unknown\)", text).group(1))
    ex = re.search(r"discarded tiles in this order: \[([^\]]+)\]\s*\n\s*\nWith decay_rate = ([0-9.]+):\s*\n```\n(.*?)```", text, re.S)
    example = {"order": [t.strip() for t in ex.group(1).split(",")], "decay_rate": float(ex.group(2)),
               "values": {m.group(1): float(m.group(2)) for m in re.finditer(r"^(\w+):.*=\s*([0-9.]+)\s*$", ex.group(3), re.M)}}
    return {"source": "docs/FEATURE_ENCODING.md", "total_channels": total[0], "channels": chans, "yaku": yaku, "shapes": shapes,
            "shanten_efficiency_divisors": sh, "shanten_efficiency_unknown": unknown, "decay_example": example}


def sequence_encoding():
    path = os.path.join(REF, "docs", "SEQUENCE_FEATURE_ENCODING.md")
    lines = open(path).read().split("\n")
    text = "\n".join(lines)
    groups = [{"name": strip_md(c[0]), "shape": [int(x) for x in re.findall(r"\d+", c[1])], "dtype": strip_md(c[2]), "line": ln}
              for ln, c in rows(lines, find_table(lines, "Feature Group", "Shape", "Type"))]
    kan37 = []
    for ln, c in rows(lines, find_table(lines, "Range", "Tiles")):
        lo, hi = span(c[0])
        kan37.append({"lo": lo, "hi": hi, "tiles": strip_md(c[1]), "line": ln})
    sparse = []
    for ln, c in rows(lines, find_table(lines, "Offset", "Count", "Feature", "Source")):
        lo, hi = span(c[0])
        sparse.append({"lo": lo, "hi": hi, "count": int(c[1]), "feature": strip_md(c[2]), "source": strip_md(c[3]), "line": ln})
    assert all(s["hi"] - s["lo"] + 1 == s["count"] for s in sparse)
    m = re.search(r"\*\*Vocabulary size: (\d+), max tokens: (\d+), padding index: (\d+)\*\*", text)
    sparse_meta = {"vocab": int(m.group(1)), "max_tokens": int(m.group(2)), "padding": int(m.group(3))}
    numeric = []
    for ln, c in rows(lines, find_table(lines, "Index", "Feature", "Source", after=find_table(lines, "Offset", "Count", "Feature", "Source") + 1)):
        lo, hi = span(c[0])
        numeric.append({"lo": lo, "hi": hi, "feature": strip_md(c[1]), "source": strip_md(c[2]), "line": ln})

    def fields(after):
        i = find_table(lines, "Field", "Vocab", "Values", after=after)
        return i, [{"field": strip_md(c[0]), "vocab": int(c[1]), "values": strip_md(c[2]), "line": ln} for ln, c in rows(lines, i)]

    def types(after):
        i = find_table(lines, "Range", "Count", "Action", "Encoding", after=after)
        out = []
        for ln, c in rows(lines, i):
            lo, hi = span(c[0])
            out.append({"lo": lo, "hi": hi, "count": int(c[1]), "action": strip_md(c[2]), "encoding": strip_md(c[3]), "line": ln})
        assert all(t["hi"] - t["lo"] + 1 == t["count"] for t in out)
        return i, out

    p0 = text.index("## 3. Progression Features")
    p_line = text[:p0].count("\n")
    fi, prog_fields = fields(p_line)
    ti, prog_types = types(fi)
    prog_pad = [int(x) for x in re.search(r"\*\*Padding tuple:\*\* `\(([\d, ]+)\)`", "\n".join(lines[p_line:])).group(1).split(",")]
    events = []
    for ln, c in rows(lines, find_table(lines, "Event", "Tuple", after=ti)):
        events.append({"event": strip_md(c[0]), "tuple": strip_md(c[1]), "line": ln})
    c0 = text.index("## 4. Candidate Features")
    c_line = text[:c0].count("\n")
    cfi, cand_fields = fields(c_line)
    _, cand_types = types(cfi)
    cand_pad = [int(x) for x in re.search(r"\*\*Padding tuple:\*\* `\(([\d, ]+)\)`", "\n".join(lines[c_line:])).group(1).split(",")]
    consts = {}
    for m in re.finditer(r"SequenceFeatureEncoder\.(\w+)\s+#\s*(\([\d, ]+\)|\d+)", text):
        v = m.group(2)
        consts[m.group(1)] = [int(x) for x in re.findall(r"\d+", v)] if v.startswith("(") else int(v)
    chi_per_suit = [int(m.group(1)) for m in re.finditer(r"^Rank \d-\d-\d \((?:no|has) 5\): \d patterns\s*->\s*(\d+)\s*$", text, re.M)]
    rel = re.search(r"`\(target - actor \+ (\d+)\) % (\d+)`", text)
    return {"source": "docs/SEQUENCE_FEATURE_ENCODING.md", "groups": groups, "kan37": kan37, "relative_seat": {"add": int(rel.group(1)), "mod": int(rel.group(2))},
            "sparse": sparse, "sparse_meta": sparse_meta, "numeric": numeric, "progression": {"fields": prog_fields, "types": prog_types, "padding": prog_pad, "events": events},
            "candidates": {"fields": cand_fields, "types": cand_types, "padding": cand_pad}, "constants": consts, "chi_patterns_per_suit": chi_per_suit,
            "pon_patterns": {"per_suit": int(re.search(r"Per suit \((\d+) patterns\)", text).group(1)), "honors": int(re.search(r"Honors: (\d+) patterns", text).group(1))}}


def main():
    out = {"generator": "scripts/gen_spec_tables.py", "reference": "smly/RiichiEnv docs/ (specification text: tables of offsets, counts and divisors)",
           "feature_encoding": feature_encoding(), "sequence_encoding": sequence_encoding()}
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
        f.write("\n")
    fe, se = out["feature_encoding"], out["sequence_encoding"]
    print(f"{OUT}: {len(fe['channels'])} channel rows ({fe['total_channels']} channels), {len(fe['yaku'])} yaku rows, {len(se['sparse'])} sparse rows, "
          f"{len(se['numeric'])} numeric rows, {len(se['progression']['types'])} + {len(se['candidates']['types'])} type rows, {len(se['progression']['events'])} event rows")


if __name__ == "__main__":
    main()
