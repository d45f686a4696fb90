#!/usr/bin/env python3
"""The four ways in, each a few lines (needs an MI355X; run from the repo root after `python -c "import __graft_entry__ as g; g.build()"`)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import riichienv_amd as rv  # noqa: E402


def reference_loop():
    """the reference's README loop, unchanged (one game, reference names)"""
    agent = rv.RandomAgent(seed=0)
    env = rv.RiichiEnv(game_mode="4p-red-half", seed=42)
    obs = env.reset()
    while not env.done():
        obs = env.step({pid: agent.act(o) for pid, o in obs.items()})
    return env.scores(), env.ranks()


def batched_rollout(n=4096, steps=300):
    """many games in lock-step with the device's RandomAgent: one launch for the whole rollout"""
    env = rv.VecRiichiEnv(n, game_mode=2, seed=0, skip_mjai_logging=True)
    env.reset()
    env.step_random(policy_seed=1, n_steps=steps, auto_reset=True)
    # ... or with the device policy that plays to win (shanten-greedy discards, every win / riichi / kan taken, a quarter of the calls)
    env.step_greedy(policy_seed=1, n_steps=steps, auto_reset=True, call_rate_256=64)
    idx, off, ent = env.legal_compact()     # what a host agent loop reads: the ordered lists of the seats that are to act
    return int(env.total_steps()), env.scores()[:2], env.points("basic")[:1].round(1).tolist(), len(idx)


def policy_loop(n=2048, iters=50):
    """a policy on the same GPU: zero-copy masks, dense feature batch of the acting seats, action ids back"""
    import torch

    from riichienv_amd.torch_env import TorchVecEnv

    env = TorchVecEnv(n, game_mode=2, seed=0)
    for it in range(iters):
        obs, index = env.obs_compact()                 # [k, 74, 34] f32 + game * 4 + seat of every row
        logits = obs.mean(dim=(1, 2))[:, None].expand(-1, 82).contiguous()                     # [k, 82] (stand-in for a network)
        env.step(env.sample_ids(seed=it, logits=logits, index=index))                           # compact logits + the index they belong to
    return tuple(obs.shape[1:]), int(env.env.total_steps())


def hands_and_logs():
    """hand math under the reference's names, and logs into training samples"""
    res = rv.HandEvaluator.hand_from_text("123m456p789s111z2z").calc(rv.parse_tile("2z"), conditions=rv.Conditions(tsumo=True, player_wind=rv.Wind.South))
    here = os.path.dirname(os.path.abspath(__file__))
    log = os.path.join(here, "..", "tests", "golden", "126_204_0_mjai.jsonl")
    kyoku = next(iter(rv.MjaiReplay.from_jsonl(log).take_kyokus()))
    decisions = [(seat, act.action_type.name) for seat, obs, act in kyoku.steps(skip_single_action=True)]
    return (res.han, res.fu, res.tsumo_agari_oya, res.tsumo_agari_ko, [y.name_en for y in res.yaku_list()]), len(decisions), decisions[:3]


if __name__ == "__main__":
    print("reference loop:", reference_loop())
    print("batched rollout:", batched_rollout())
    print("policy loop:", policy_loop())
    print("hands and logs:", hands_and_logs())
    print(np.__name__, "ok")
