"""Host logic of row N1: select_action_from_mjai / tid_to_mjai against the reference's tests
(tests/env/actions/test_action_to_mjai.py:25-88) and the branches of observation/mjai_select.rs:88-194."""
from riichienv_amd import abi, mjai

P = abi.pack_action


def test_tid_to_mjai_roundtrip():
    names = [mjai.tid_to_mjai(t) for t in (0, 16, 17, 52, 53, 88, 104, 108, 124, 132, 135)]
    assert names == ["1m", "5mr", "5m", "5pr", "5p", "5sr", "9s", "E", "P", "C", "C"]
    for s in ("1m", "5mr", "5m", "9p", "E", "C"):
        assert mjai.tid_to_mjai(abi.mjai_to_tid(s)) == s


def test_reference_select_cases():
    legal = [P(abi.DISCARD, 53), P(abi.RIICHI)]
    a = mjai.select_action_from_mjai(legal, {"type": "dahai", "pai": "5p"})
    assert abi.unpack_action(a)[:2] == (abi.DISCARD, 53)
    assert abi.unpack_action(mjai.select_action_from_mjai(legal, {"type": "reach"}))[0] == abi.RIICHI
    assert mjai.select_action_from_mjai(legal, {"type": "dahai", "pai": "1z"}) is None
    loose = {"type": "dahai", "pai": "5p", "tsumogiri": True, "meta": {"foo": "bar"}}
    assert abi.unpack_action(mjai.select_action_from_mjai(legal, loose))[:2] == (abi.DISCARD, 53)
    assert abi.unpack_action(mjai.select_action_from_mjai(legal, '{"type":"dahai","pai":"5p"}'))[1] == 53


def test_select_branches():
    legal = [P(abi.DISCARD, 17), P(abi.DISCARD, 18), P(abi.CHI, 53, (49, 57)), P(abi.CHI, 53, (57, 61)), P(abi.PON, 53, (54, 55)),
             P(abi.RON, 53), P(abi.PASS), P(abi.KYUSHU), P(abi.KITA, 120)]
    sel = lambda m, **kw: mjai.select_action_from_mjai(legal, m, **kw)  # noqa: E731
    assert sel({"type": "hora"}) == P(abi.RON, 53)
    assert sel({"type": "none"}) == P(abi.PASS)
    assert sel({"type": "ryukyoku"}) == P(abi.KYUSHU)
    assert sel({"type": "chi", "pai": "5p", "consumed": ["6p", "7p"]}) == P(abi.CHI, 53, (57, 61))
    assert sel({"type": "chi", "pai": "5p", "consumed": ["6p", "4p"]}) == P(abi.CHI, 53, (49, 57))
    assert sel({"type": "chi", "pai": "4p", "consumed": ["6p", "7p"]}) is None
    assert sel({"type": "chi", "pai": "5p", "consumed": ["6p", "7p"]}, three_player=True) is None
    assert sel({"type": "pon"}) == P(abi.PON, 53, (54, 55))
    assert sel({"type": "kita"}) is None and sel({"type": "kita"}, three_player=True) == P(abi.KITA, 120)
    # two 5m discards (ids 17, 18): tsumogiri picks the drawn one, tedashi the other; no hint -> the first
    assert sel({"type": "dahai", "pai": "5m", "tsumogiri": True}, drawn_tile=18) == P(abi.DISCARD, 18)
    assert sel({"type": "dahai", "pai": "5m", "tsumogiri": False}, drawn_tile=17) == P(abi.DISCARD, 18)
    assert sel({"type": "dahai", "pai": "5m"}) == P(abi.DISCARD, 17)
    assert sel({"type": "dahai"}) == P(abi.DISCARD, 17)          # malformed but lenient (mjai_select.rs:121-128)
    assert sel({"type": "bogus"}) is None and sel(12345) is None and sel("not json") is None
