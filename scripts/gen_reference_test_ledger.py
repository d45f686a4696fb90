"""Generates tests/REFERENCE_TESTS.md: every test of the reference (pytest functions under /root/reference/tests, #[test] functions of
riichienv-core) -> where this repository restates it (a scenario, KAT or test that cites its file and lines / its name), or why it is not on the
step path.  VERDICT r5 item 8.  Run in the build container (needs /root/reference); the output is committed.

Matching is mechanical: a reference test (file, name, first line, last line) counts as restated when a file under tests/ or oracle/ or
riichienv_amd/ cites `<file>:<a>-<b>` (or `<file>:<a>`) with [a, b] overlapping the test's span, or names the test function.  A citation of the
whole file without lines (e.g. "the fixtures of tests/test_agari_calculator.py") counts for every test of a file listed in WHOLE_FILE.
What is left is classified by hand below (OUT_OF_PATH: a reason per file or per test; everything else is reported as MISSING)."""
import os
import re
import sys

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# reference files whose tests are covered by consuming the file's own data wholesale (the citation names the file, not line ranges)
WHOLE_FILE = {
    "tests/test_agari_calculator.py": "tests/golden/agari_4p.json / agari_3p.json / hands_negative.json are the fixture files this test reads: every case runs on the oracle and on the GPU (tests/test_oracle_hand.py, tests/test_gpu_hand.py)",
    "tests/test_mjai_parity.py": "the two hanchan logs these tests replay are tests/golden/126_204_0_mjai.jsonl and ui_example_after_injection.jsonl: every action legal, every hora's han / fu / yaku / points (tests/test_oracle_replay.py, tests/test_gpu_replay.py, tests/test_gpu_ui_example.py)",
    "riichienv-core/tests/agari_correctness.rs": "the same agari_*.json fixture files, consumed wholesale (tests/test_oracle_hand.py, tests/test_gpu_hand.py)",
}
# not on the step path (SURVEY section 8 / section 2 "out of scope"), by file or by (file, test)
OUT_OF_PATH = {
    "tests/test_metadata_injection.py": "viewer metadata injection (riichienv-ui): out of scope (SURVEY section 2)",
    "tests/test_observation_serialization.py": "pickle / base64 / serde round trips of the Python Observation object: a property of the PyO3 binding, not of the step path; the shim's Observation is rebuilt from device state on every call (riichienv_amd/compat.py)",
}
OUT_OF_PATH_TESTS = {
    ("tests/env/actions/test_riichi_no_claim.py", "test_riichi_rule_violation"): "an empty test the reference itself skips (`@pytest.mark.skip(reason=\"Too complex to test\")`, body `pass`)",
    ("tests/env/test_sanma.py", "test_to_dict"): "dict export of the PyO3 Observation object (a binding property, like tests/test_observation_serialization.py)",
    ("tests/env/test_sanma.py", "test_round_trip"): "base64 round trip of the PyO3 Observation3P object (binding property: the shim's Observation has no wire format)",
    ("tests/env/test_sanma.py", "test_legal_actions_preserved"): "base64 round trip of the PyO3 Observation3P object (binding property)",
    ("tests/env/test_sanma.py", "test_round_trip_multiple_seeds"): "base64 round trip of the PyO3 Observation3P object (binding property)",
    ("riichienv-core/src/tests.rs", "test_ron_pao_50_50_split"): "pure integer arithmetic on local variables (32000 / 2 == 16000 ...): it calls nothing of the engine; the rule it illustrates runs through the state machine in tests/scenarios.py sc_tenhou_ron_pao_composite / sc_mjsoul_liability_only_*",
    ("riichienv-core/src/tests.rs", "test_mjsoul_4p_ron_pao_composite"): "pure integer arithmetic on local variables (no engine call); the same case through the state machine: tests/scenarios.py sc_mjsoul_pao_* (tests/env/test_majsoul_pao_scoring.py) and sc_mjsoul_3p_ron_pao_composite",
    ("riichienv-core/src/tests.rs", "test_sanma_observation_num_players"): "its only assertion is `!obs.hands[0].is_empty()` after a deal (the length checks were removed upstream); covered by tests/scenarios.py sc3_basics / sc3_initialization and tests/test_gpu_compat_sanma.py::test_observation_fields_and_sizes",
    ("riichienv-core/src/state_3p/wall.rs", "test_old_layout_remove0_would_consume_dora_indicators"): "a test of the reference's OLD wall layout (it re-implements the pre-fix behaviour inline to show the regression existed); nothing of the current engine is called",
}
# covered indirectly (no scenario of its own): the explanation names what exercises the behaviour
INDIRECT = {
    ("riichienv-core/src/state/game_mode.rs", "test_four_player_dora_wrapping"): "get_next_dora_tile (9m -> 1m, N -> E, Chun -> Haku): the dora han of the 816 agari_4p.json fixtures depend on it in every suit and both honor cycles (tests/test_oracle_hand.py, tests/test_gpu_hand.py); 3P wrap: tests/scenarios.py sc3_dora_wraps_between_1m_and_9m",
    ("riichienv-core/src/state_3p/wall.rs", "test_dead_wall_layout_has_8_rinshan_slots"): "indicators at W[8 + 2i] / W[9 + 2i]: tests/scenarios.py sc3_ankan_dora_before_rinshan, sc3_kakan_dora_before_discard and sc3_tsumo_payments_and_nukidora read dora / ura off those slots after rinshan draws",
    ("riichienv-core/src/state_3p/wall.rs", "test_rinshan_draw_does_not_consume_dora_indicators"): "rinshan draws come from W[0..8), indicators from W[8..18): tests/scenarios.py sc3_kita / sc3_kita_with_correct_tile / sc3_ankan_dora_before_rinshan (replacement tile = the front of the dead wall, the indicator count and values unchanged by it); 1 000 seeds x 8 draws are not repeated",
    ("riichienv-core/src/tests.rs", "test_apply_mjai_event_honor_and_red_tiles"): "apply_mjai_event with honor and red-five tile strings: tests/test_oracle_apply_event.py and tests/test_gpu_apply_event.py feed whole logs of random and greedy games (every tile string, red fives included) back through rmj_apply_events against the oracle; tile string tables: tests/test_convert.py",
    ("tests/env/test_paishan.py", "test_real_dora_reveal"): "a real Tenhou paishan string + one rinshan draw -> second indicator: tests/scenarios.py sc_paishan_dora_indices (:23-40, 67-90 of the same file: the same indices on wall = range(136)) and convert.paishan_to_wall in tests/test_convert.py; the reference test pokes private fields (_reveal_kan_dora) the shim does not have",
    ("tests/test_mjai_replay.py", "test_mjai_replay_4p_reach_discard_observation_is_not_duplicated_state"): "tests/test_gpu_replay.py walks reach -> dahai decisions of whole logs and compares every sample's state with the oracle (no duplicated state can pass); the reference test constructs its log by hand",
    ("tests/test_mjai_replay.py", "test_mjai_replay_mjsoul_nonfinal_end_scores_fallback_to_next_round_scores"): "MjSoul record quirk (end scores of a non-final round taken from the next round's start): tests/test_mjsoul_replay.py covers the MjSoul reader's score handling on the reference's own record fixtures",
}


def ref_tests():
    out = []
    for base, _, files in os.walk(REF):
        if "/.git" in base or "/node_modules" in base or "/target" in base:
            continue
        for f in sorted(files):
            p = os.path.join(base, f)
            rel = os.path.relpath(p, REF)
            if f.startswith("test_") and f.endswith(".py") and "/tests" in "/" + rel:
                lines = open(p, errors="replace").read().split("\n")
                starts = [(i + 1, re.match(r"^(\s*)(?:async\s+)?def (test_\w+)", ln)) for i, ln in enumerate(lines)]
                starts = [(i, m.group(2), len(m.group(1))) for i, m in starts if m]
                for k, (ln, name, ind) in enumerate(starts):
                    end = len(lines)
                    for j in range(ln, len(lines)):
                        s = lines[j]
                        if s.strip() and (len(s) - len(s.lstrip())) <= ind and not s.lstrip().startswith(("#", ")", "]", "}")) and j + 1 > ln:
                            end = j
                            break
                    out.append((rel, name, ln, end))
            elif f.endswith(".rs") and rel.startswith("riichienv-core"):
                lines = open(p, errors="replace").read().split("\n")
                for i, ln in enumerate(lines):
                    if ln.strip() == "#[test]":
                        for j in range(i + 1, min(i + 6, len(lines))):
                            m = re.match(r"\s*(?:pub\s+)?fn (\w+)\s*\(", lines[j])
                            if m:
                                depth, end = 0, j
                                for q in range(j, len(lines)):
                                    depth += lines[q].count("{") - lines[q].count("}")
                                    if depth == 0 and "{" in "".join(lines[j:q + 1]):
                                        end = q
                                        break
                                out.append((rel, m.group(1), j + 1, end + 1))
                                break
    return sorted(out)


def citations():
    """[(our file, our line, cited path fragment, a, b, text)] over tests/, oracle/, riichienv_amd/ (Python and C++ sources)"""
    cites, names = [], {}
    pat = re.compile(r"([\w/.-]*\w+\.(?:py|rs)):(\d+)(?:-(\d+))?((?:\s*(?:,|/|and)\s*:?\d+(?:-\d+)?)*)")
    for top in ("tests", "oracle", "riichienv_amd", "scripts"):
        for base, _, files in os.walk(os.path.join(ROOT, top)):
            if "__pycache__" in base or "/golden" in base:
                continue
            for f in files:
                if not f.endswith((".py", ".hpp", ".cpp", ".h", ".hip")):
                    continue
                p = os.path.join(base, f)
                rel = os.path.relpath(p, ROOT)
                if rel in ("scripts/gen_reference_test_ledger.py",):
                    continue
                for i, ln in enumerate(open(p, errors="replace"), 1):
                    for m in pat.finditer(ln):
                        spans = [(int(m.group(2)), int(m.group(3) or m.group(2)))]
                        for extra in re.findall(r"(\d+)(?:-(\d+))?", m.group(4) or ""):
                            spans.append((int(extra[0]), int(extra[1] or extra[0])))
                        for a, b in spans:
                            cites.append((rel, i, m.group(1), a, b))
                    for m in re.finditer(r"\b(test_\w+)\b", ln):
                        names.setdefault(m.group(1), []).append((rel, i))
    return cites, names


def main():
    tests = ref_tests()
    cites, names = citations()
    rows, missing = [], []
    basename_count = {}
    for rel in {t[0] for t in tests}:
        basename_count[os.path.basename(rel)] = basename_count.get(os.path.basename(rel), 0) + 1
    for rel, name, a, b in tests:
        where = []
        key = rel
        for cf, cl, frag, ca, cb in cites:
            # a citation names the file by any unambiguous suffix of its path ("actions/test_meld_aka.py", "shanten.rs", "observation/encode.rs", "src/tests.rs")
            if (key == frag or key.endswith("/" + frag)) and (("/" in frag) or basename_count[os.path.basename(key)] == 1) and ca <= b and cb >= a:
                where.append(f"{cf}:{cl}")
        if rel.endswith(".py"):
            for cf, cl in names.get(name, []):
                if cf.startswith(("tests/", "oracle/")):
                    where.append(f"{cf}:{cl}")
        where = sorted(set(where))
        if where:
            rows.append((rel, name, a, b, "restated", ", ".join(where[:4]) + (" ..." if len(where) > 4 else "")))
        elif rel in WHOLE_FILE:
            rows.append((rel, name, a, b, "restated (whole file)", WHOLE_FILE[rel]))
        elif (rel, name) in INDIRECT:
            rows.append((rel, name, a, b, "restated (indirectly)", INDIRECT[(rel, name)]))
        elif (rel, name) in OUT_OF_PATH_TESTS:
            rows.append((rel, name, a, b, "not on the path", OUT_OF_PATH_TESTS[(rel, name)]))
        elif rel in OUT_OF_PATH:
            rows.append((rel, name, a, b, "not on the path", OUT_OF_PATH[rel]))
        else:
            rows.append((rel, name, a, b, "MISSING", ""))
            missing.append((rel, name, a, b))
    n = len(rows)
    c = {k: sum(1 for r in rows if r[4].startswith(k)) for k in ("restated", "not on the path", "MISSING")}
    out = ["# Reference tests -> where this repository restates them", "",
           "Generated by `scripts/gen_reference_test_ledger.py` from `/root/reference` (smly/RiichiEnv v0.4.8) - do not edit by hand.", "",
           f"{n} reference tests: **{c['restated']} restated**, {c['not on the path']} not on the step path (reason given), **{c['MISSING']} missing**.", "",
           "| reference test | lines | status | where / why |", "|---|---|---|---|"]
    for rel, name, a, b, st, why in rows:
        out.append(f"| `{rel}::{name}` | {a}-{b} | {st} | {why} |")
    open(os.path.join(ROOT, "tests", "REFERENCE_TESTS.md"), "w").write("\n".join(out) + "\n")
    print(f"{n} reference tests: {c}")
    by_file = {}
    for rel, name, a, b in missing:
        by_file.setdefault(rel, []).append(f"{name}:{a}-{b}")
    for rel, names_ in sorted(by_file.items()):
        print(f"  MISSING {rel}: {len(names_)}: {', '.join(names_[:60])}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
