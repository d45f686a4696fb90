"""Action ids of the 82-way (4P) and 60-way (3P) spaces: riichienv-core/src/tests.rs:1258-1469 (test_action_encode_*,
test_action_space_size, test_3p_encode_all_discard_ids_contiguous) on the host-side Action of riichienv_amd.compat and on the
oracle's encoders (the device's a_encode / a_encode_3p are compared with the oracle through every mask in the GPU suite)."""
import pytest

from riichienv_amd import abi
from riichienv_amd.compat import Action, ActionType


def _orc(sanma):
    from oracle import oracle

    L = oracle.lib()
    return L.orc_action_encode_3p if sanma else L.orc_action_encode


def both(sanma, atype, tile=None, consume=()):
    a = Action(ActionType(atype), tile, list(consume))
    host = a.encode_3p() if sanma else a.encode()
    assert _orc(sanma)(abi.pack_action(atype, tile, list(consume))) == host
    return host


def test_4p_discard_and_special_ids():
    assert [both(False, abi.DISCARD, t) for t in (0, 4, 132)] == [0, 1, 33]
    assert both(False, abi.RIICHI) == 37 and both(False, abi.PON) == 41 and both(False, abi.DAIMINKAN, 0) == 42
    assert both(False, abi.RON) == 79 and both(False, abi.KYUSHU) == 80 and both(False, abi.PASS) == 81
    with pytest.raises(ValueError):
        Action(ActionType.KITA).encode()               # no Kita in the 82-way space
    assert _orc(False)(abi.pack_action(abi.KITA)) < 0


def test_3p_discard_and_special_ids():
    assert [both(True, abi.DISCARD, t) for t in (0, 32, 36, 68, 72, 132)] == [0, 1, 2, 10, 11, 26]
    assert both(True, abi.RIICHI) == 27 and both(True, abi.PON) == 28
    assert both(True, abi.DAIMINKAN, 0) == 29 and both(True, abi.DAIMINKAN, 32) == 30
    assert both(True, abi.ANKAN, None, [132]) == 55
    assert both(True, abi.RON) == 56 and both(True, abi.KYUSHU) == 57 and both(True, abi.PASS) == 58 and both(True, abi.KITA) == 59


def test_3p_rejects_manzu_2_to_8_and_chi():
    for t in (4, 16, 28):
        with pytest.raises(ValueError):
            Action(ActionType.DISCARD, t).encode_3p()
        assert _orc(True)(abi.pack_action(abi.DISCARD, t)) < 0
    with pytest.raises(ValueError):
        Action(ActionType.CHI, 36, [40, 44]).encode_3p()
    assert _orc(True)(abi.pack_action(abi.CHI, 36, [40, 44])) < 0


def test_action_space_sizes_and_contiguous_3p_discards():
    assert abi.ACTION_SPACE_4P == 82 and abi.ACTION_SPACE_3P == 60
    valid = [0] + list(range(8, 34))
    assert len(valid) == 27
    assert sorted(both(True, abi.DISCARD, 4 * t) for t in valid) == list(range(27))


def test_action_to_mjai_like_the_reference():
    """tests/env/actions/test_action_to_mjai.py:7-24 (dahai / chi / reach), tests/test_core.py:190-217 and src/tests.rs:835-854 (a reach carries its actor only
    when one is set; the attribute can be re-assigned) on the shim's host-side Action (action.rs:107-149)."""
    import json

    assert json.loads(Action(ActionType.DISCARD, tile=53).to_mjai()) == {"type": "dahai", "pai": "5p"}
    assert json.loads(Action(ActionType.CHI, tile=53, consume_tiles=[49, 57]).to_mjai()) == {"type": "chi", "pai": "5p", "consumed": ["4p", "6p"]}
    assert json.loads(Action(ActionType.RIICHI).to_mjai()) == {"type": "reach"}
    a = Action(type=ActionType.RIICHI, actor=2)
    assert json.loads(a.to_mjai()) == {"type": "reach", "actor": 2}
    assert "actor" not in json.loads(Action(type=ActionType.RIICHI).to_mjai())
    a.actor = 0
    assert json.loads(a.to_mjai())["actor"] == 0
    a.actor = None
    assert "actor" not in json.loads(a.to_mjai())
