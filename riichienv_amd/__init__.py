"""riichienv_amd — MI355X-native batched Riichi Mahjong step path (drop-in for smly/RiichiEnv's hot path).

`VecRiichiEnv` (riichienv_amd.vecenv) is the batched environment over the C-ABI library;
`riichienv_amd.compat` mirrors the reference's scalar names (RiichiEnv, Action, Observation, ...).
"""
from .vecenv import VecRiichiEnv, RmjError, load_lib  # noqa: F401

__all__ = ["VecRiichiEnv", "RmjError", "load_lib"]
