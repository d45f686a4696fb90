"""GPU: round boundaries, per-round score deltas and rank rewards for a trainer on the same GPU (rmj_round_track_device,
TorchVecEnv.step_rl, GymVectorAdapter) against the oracle's score history - what riichienv-ml's PPO worker computes on the host
(trainers/_ppo_worker.py:100-116, 240-266, 283-291)."""
import numpy as np
import pytest

from riichienv_amd.shard import game_seed

pytestmark = pytest.mark.gpu


def _facts(o):
    v = o.peek()
    return int(v.hand_index), bool(v.is_done), [int(v.players[p].score) for p in range(4)], [int(v.round_wind), int(v.oya), int(v.honba), int(v.riichi_sticks)]


@pytest.mark.parametrize("mode", [2, 4])
def test_round_tracker_equals_the_oracles_score_history_under_winning_play(mode):
    from oracle import oracle
    from riichienv_amd.torch_env import TorchVecEnv

    n, seed, pseed, steps, rate = 96, 100 + mode, 9, 700, 64
    tv = TorchVecEnv(n, game_mode=mode, seed=seed, skip_mjai_logging=True)
    games = [oracle.Game(game_mode=mode, seed=game_seed(seed, g), skip_log=True) for g in range(n)]
    for o in games:
        o.reset()
    track = [dict(zip(("hi", "done", "start", "meta"), _facts(o))) for o in games]
    tv.round_track()                                      # baseline
    seen = {1: 0, 2: 0, "restart": 0}
    for _ in range(steps):
        tv.env.step_greedy(pseed, 1, auto_reset=True, call_rate_256=rate)
        ended, delta, meta, kidx = (x.cpu().numpy() for x in tv.round_track())
        for g, o in enumerate(games):
            if o.status()[2]:
                o.reset()
            else:
                o.step(o.greedy_actions(pseed, g, rate))
            hi, done, sc, mt = _facts(o)
            t = track[g]
            want, rebase = 0, False
            if t["done"] and not done:
                rebase = True
                seen["restart"] += 1
            elif done and not t["done"]:
                want = 2
            elif not done and hi != t["hi"]:
                want, rebase = 1, True
            assert ended[g] == want, (g, ended[g], want)
            if want:
                # the worker's reward inputs at a kyoku boundary: cur_scores - kyoku_start_scores and the round's opening facts
                assert list(delta[g]) == [sc[p] - t["start"][p] for p in range(4)], g
                assert list(meta[g]) == t["meta"], g
                seen[want] += 1
            else:
                assert not delta[g].any() and not meta[g].any()
            assert kidx[g] == o.peek().kyoku_idx
            if rebase:
                t["hi"], t["start"], t["meta"] = hi, sc, mt
            t["done"] = done
    assert seen[1] > 200 and seen[2] > 10 and seen["restart"] > 5, seen


def test_step_rl_pays_score_deltas_and_rank_rewards():
    import torch

    from riichienv_amd.torch_env import TorchVecEnv

    n = 256
    tv = TorchVecEnv(n, game_mode=0, seed=5)              # single-round games: every game ends with its first round
    total = torch.zeros((n, 4), dtype=torch.float32, device=tv.device)
    finished = torch.zeros((n,), dtype=torch.bool, device=tv.device)
    start = tv.scores().clone()
    final_scores = torch.zeros_like(start)
    final_ranks = torch.zeros((n, 4), dtype=torch.int64, device=tv.device)
    for it in range(400):
        ids = tv.sample_ids(None, seed=it).clone()
        ids[finished] = -1
        obs, reward, terminated, info = tv.step_rl(ids, auto_reset=False)
        assert obs.shape == (n, 4, 74, 34) and reward.shape == (n, 4) and terminated.dtype == torch.bool
        total += torch.where(finished[:, None], torch.zeros_like(reward), reward)
        newly = terminated & ~finished
        final_scores[newly] = info["scores"][newly]
        final_ranks[newly] = info["ranks"][newly]
        finished |= terminated
        if bool(finished.all()):
            break
    assert bool(finished.all())
    table = torch.tensor([0.0, 10.0, 4.0, -4.0, -10.0], device=tv.device)
    want = (final_scores - start).to(torch.float32) / 1000.0 + table[final_ranks]
    assert torch.allclose(total, want), (total - want).abs().max()


def test_gym_vector_adapter_plays_hero_against_the_sampler():
    import torch

    from riichienv_amd.torch_env import GymVectorAdapter, TorchVecEnv

    n = 64
    tv = TorchVecEnv(n, game_mode=0, seed=21)
    hero = torch.arange(n, device=tv.device) % 4
    env = GymVectorAdapter(tv, hero=hero)
    obs, info = env.reset(seed=1)
    assert obs["features"].shape == (n, 74, 34) and obs["mask"].shape == (n, 82)
    assert bool(obs["mask"].any(-1).all())               # every env waits at its hero's decision
    g = torch.Generator(device="cpu").manual_seed(0)
    ended = 0
    ret = torch.zeros((n,), device=tv.device)
    for _ in range(60):
        m = obs["mask"].to(torch.float32).cpu()
        has = m.sum(-1) > 0
        a = torch.multinomial(torch.where(has[:, None], m, torch.ones_like(m)), 1, generator=g)[:, 0].to(tv.device)
        obs, reward, terminated, truncated, info = env.step(a)
        assert reward.shape == (n,) and terminated.shape == (n,) and not bool(truncated.any())
        live = ~terminated
        assert bool(obs["mask"][live].any(-1).all())     # not over: the hero is to act again
        ret += reward
        ended += int(terminated.sum())
    assert ended >= n                                     # single-round games: all of them ended at least once
    assert float(ret.abs().sum()) > 0


def test_sample_ids_from_compact_logits_without_a_host_count():
    """rows of `index` behind the device count are stale and may repeat live (game, seat) pairs"""
    import torch

    from riichienv_amd.torch_env import TorchVecEnv

    n = 512
    tv = TorchVecEnv(n, game_mode=2, seed=8)
    for it in range(40):
        tv.step(tv.sample_ids(None, seed=it))
    obs, idx = tv.obs_compact(sync_count=True)
    k = idx.shape[0]
    logits = torch.randn((k, 82), device=tv.device)
    want = tv.sample_ids(logits, seed=123, index=idx).clone()
    cobs, cidx, ccnt = tv.obs_compact(sync_count=False)
    assert int(ccnt.item()) == k and cidx.shape[0] > k
    cidx[k:] = cidx[0]                                     # the worst case: every stale row names a live pair
    full = torch.randn((cidx.shape[0], 82), device=tv.device) * 50.0
    full[:k] = logits
    got = tv.sample_ids(full, seed=123, index=cidx, count=ccnt)
    assert torch.equal(got, want)


def test_round_tracker_takes_a_manual_reset_or_poke_as_a_new_baseline():
    """resetting (or poking) a LIVE game used to look like a round end (hand_index moved) and paid out
    new start scores - old start scores.  rmj_reset / rmj_poke_state mark the games they touch; the tracker rebases on them."""
    from riichienv_amd.torch_env import TorchVecEnv

    n = 64
    tv = TorchVecEnv(n, game_mode=2, seed=3, skip_mjai_logging=True)
    tv.round_track()
    tv.env.step_greedy(5, 40, auto_reset=True, call_rate_256=64)
    tv.round_track()
    sel = np.zeros(n, np.uint8)
    sel[::2] = 1
    sc = np.tile(np.array([40000, 30000, 20000, 10000], np.int32), (n, 1))
    tv.env.reset(select=sel, scores=sc)                     # live games, other scores, hand_index + 1
    v = tv.env.peek(1)
    v.players[0].score += 5000                              # a poke of a game that was not reset
    tv.env.poke(1, v)
    ended, delta, meta, _ = (x.cpu().numpy() for x in tv.round_track())
    assert not ended[::2].any() and not delta[::2].any() and not ended[1] and not delta[1].any()
    # the next real round end of a reset game pays against the scores it was reset to
    for _ in range(200):
        tv.env.step_greedy(5, 1, auto_reset=False, call_rate_256=64)
        ended, delta, meta, _ = (x.cpu().numpy() for x in tv.round_track())
        hit = [g for g in range(0, n, 2) if ended[g]]
        if hit:
            g = hit[0]
            now = [int(tv.env.peek(g).players[p].score) for p in range(4)]
            assert list(delta[g]) == [now[p] - int(sc[g][p]) for p in range(4)]
            break
    else:
        raise AssertionError("no reset game ended a round in 200 steps")
