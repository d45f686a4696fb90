"""riichienv_amd — MI355X-native batched Riichi Mahjong step path (drop-in for smly/RiichiEnv's hot path).

`VecRiichiEnv` (riichienv_amd.vecenv) is the batched environment over the C-ABI library.  The names of the reference's Python
package (`import riichienv as rv`: src/riichienv/__init__.py) resolve lazily to their counterparts here - RiichiEnv, Action,
ActionType, Observation, Meld, MeldType, Phase, GameRule, GameType, RandomAgent (riichienv_amd.compat); Conditions, HandEvaluator,
HandEvaluator3P, Wind, WinResult, Score, calculate_score, calculate_shanten, calculate_shanten_3p, check_riichi_candidates,
parse_hand, parse_tile (riichienv_amd.hand); MjaiReplay, MjSoulReplay, Kyoku, WinResultContext (riichienv_amd.replay); Yaku, get_yaku_by_id, get_all_yaku
(riichienv_amd.yaku_table); the
`convert` and `consts` modules.
"""
from .vecenv import VecRiichiEnv, RmjError, load_lib  # noqa: F401

_LAZY = {
    "compat": ("RiichiEnv", "Action", "ActionType", "Observation", "Meld", "MeldType", "Phase", "GameRule", "RandomAgent", "GameType"),
    "hand": ("Conditions", "HandEvaluator", "HandEvaluator3P", "Wind", "WinResult", "Score", "calculate_score", "calculate_shanten",
             "calculate_shanten_3p", "check_riichi_candidates", "parse_hand", "parse_tile"),
    "replay": ("MjaiReplay", "MjSoulReplay", "Kyoku", "WinResultContext"),
    "yaku_table": ("Yaku", "get_yaku_by_id", "get_all_yaku"),
}
__all__ = ["VecRiichiEnv", "RmjError", "load_lib", "convert", "consts"] + [n for names in _LAZY.values() for n in names]


def __getattr__(name):
    import importlib

    if name in ("convert", "consts"):
        return importlib.import_module("." + name, __name__)
    for mod, names in _LAZY.items():
        if name in names:
            return getattr(importlib.import_module("." + mod, __name__), name)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
