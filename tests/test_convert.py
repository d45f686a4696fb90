"""riichienv_amd.convert (the reference's src/riichienv/convert.py under the same names): the reference's tests/test_convert.py
and the string half of tests/env/test_paishan.py transcribed, and every output of the reference module recorded in
tests/golden/convert_vectors.json (scripts/gen_convert_vectors.py: all 136 ids, all names, list cases, walls, rejected inputs)."""
import json
import os

import pytest

from riichienv_amd import abi, convert, mjai

VEC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "convert_vectors.json")


def test_reference_unit_tests():
    """tests/test_convert.py:5-66"""
    assert [convert.tid_to_mpsz(t) for t in (0, 16, 17, 32, 108, 124)] == ["1m", "0m", "5m", "9m", "1z", "5z"]
    assert [convert.tid_to_mjai(t) for t in (0, 16, 17, 108, 112, 124, 128, 132)] == ["1m", "5mr", "5m", "E", "S", "P", "F", "C"]
    assert [convert.mpsz_to_tid(s) for s in ("1m", "0m", "5m", "1z", "5z")] == [0, 16, 17, 108, 124]
    assert [convert.mjai_to_tid(s) for s in ("1m", "5mr", "5m", "E", "P", "C")] == [0, 16, 17, 108, 124, 132]
    assert (convert.mjai_to_mpsz("E"), convert.mjai_to_mpsz("5mr")) == ("1z", "0m")
    assert (convert.mpsz_to_mjai("1z"), convert.mpsz_to_mjai("0p")) == ("E", "5pr")
    assert convert.tid_to_mpsz_list([0, 16, 124]) == ["1m", "0m", "5z"]
    assert convert.tid_to_mjai_list([0, 16, 124]) == ["1m", "5mr", "P"]


def test_paishan_to_wall_kats():
    """tests/env/test_paishan.py:7-22"""
    wall = convert.paishan_to_wall("1m2m3m")
    assert len(wall) == 3 and wall[0] // 4 == 0 and wall[1] // 4 == 1
    base = convert.mpsz_to_tid("1m")
    assert convert.paishan_to_wall("1m1m") == [base, base + 1]
    with pytest.raises(ValueError):
        convert.paishan_to_wall("1m2")


def test_every_recorded_output_of_the_reference_module():
    with open(VEC) as f:
        v = json.load(f)
    assert [convert.tid_to_mpsz(t) for t in range(136)] == v["tid_to_mpsz"]
    assert [convert.tid_to_mjai(t) for t in range(136)] == v["tid_to_mjai"]
    for name in ("mpsz_to_tid", "mjai_to_tid", "mpsz_to_mjai", "mjai_to_mpsz"):
        for s, want in v[name].items():
            assert getattr(convert, name)(s) == want, (name, s)
    for c in v["lists"]:
        assert convert.tid_to_mpsz_list(c["tids"]) == c["mpsz"] and convert.tid_to_mjai_list(c["tids"]) == c["mjai"]
        assert convert.mpsz_to_tid_list(c["mpsz"]) == c["mpsz_back"] and convert.mjai_to_tid_list(c["mjai"]) == c["mjai_back"]
        assert convert.mpsz_to_mjai_list(c["mpsz"]) == c["mpsz_mjai"] and convert.mjai_to_mpsz_list(c["mjai"]) == c["mjai_mpsz"]
    for w in v["walls"]:
        wall = convert.paishan_to_wall(w["paishan"])
        assert wall == w["wall"] and sorted(wall) == list(range(136))
    for fn, key in ((convert.mpsz_to_tid, "bad_mpsz"), (convert.mjai_to_tid, "bad_mjai")):
        for s, want in v[key].items():
            if want == "ValueError":
                with pytest.raises(ValueError):
                    fn(s)
            else:
                fn(s)
    for t in (-1, 136):
        with pytest.raises(ValueError):
            convert.tid_to_mpsz(t)
        with pytest.raises(ValueError):
            convert.tid_to_mjai(t)


def test_the_other_name_tables_of_the_package_agree():
    """abi.mjai_to_tid (parser.rs:336-385) and mjai.tid_to_mjai (parser.rs:301-334) are the Rust side's tables; the Python
    module's must be the same maps"""
    for t in range(136):
        assert mjai.tid_to_mjai(t) == convert.tid_to_mjai(t)
    for s in [f"{n}{x}" for x in "mps" for n in range(1, 10)] + ["5mr", "5pr", "5sr"] + list("ESWNPFC"):
        assert abi.mjai_to_tid(s) == convert.mjai_to_tid(s)
