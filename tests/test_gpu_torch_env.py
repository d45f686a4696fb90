"""Row N4: the trainer-side loop (torch tensors on the GPU, zero-copy masks, id-based stepping) against the oracle
driven with the same action ids through find_action semantics."""
from riichienv_amd.shard import game_seed
import numpy as np
import pytest

from riichienv_amd import abi

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode,shared", [(2, True), (5, True), (2, False)])
def test_torch_env_matches_oracle(mode, shared):
    torch = pytest.importorskip("torch")
    from oracle import oracle
    from riichienv_amd.torch_env import TorchVecEnv

    n, seed = 48, 700 + mode
    sanma = mode >= 3
    env = TorchVecEnv(n, game_mode=mode, seed=seed, extended=False, skip_mjai_logging=False, share_stream=shared)
    games = [oracle.Game(game_mode=mode, seed=game_seed(seed, g)) for g in range(n)]
    for o in games:
        o.reset()
    gen = torch.Generator(device=env.device)
    gen.manual_seed(1234)
    enc_fn = games[0].L.orc_action_encode_3p if sanma else games[0].L.orc_action_encode
    for step in range(700):
        act = env.active().cpu().numpy()
        mask = env.mask.cpu().numpy()
        ids = env.sample_random_ids(gen)
        ids_h = ids.cpu().numpy()
        for g, o in enumerate(games):
            oa, _, od = o.status()
            assert bool(env.done()[g]) == bool(od), (step, g)
            if od:
                o.reset()
                continue
            acts = [abi.NO_ACTION] * 4
            for s in range(4):
                assert act[g, s] == bool((oa >> s) & 1), (step, g, s)
                if act[g, s]:
                    assert (mask[g, s] == o.mask(s)).all(), (step, g, s)
                    cand = [a for a in o.legal(s) if enc_fn(a) == int(ids_h[g, s])]
                    assert cand, (step, g, s, int(ids_h[g, s]))
                    acts[s] = cand[0]
            o.step(acts)
        env.step(ids, auto_reset=True)
        if step % 100 == 0:
            obs = env.obs(only_active=False).cpu().numpy()
            for g in (0, n - 1):
                for s in range(3 if sanma else 4):
                    assert obs[g, s].tobytes() == games[g].encode(s, sanma).tobytes(), (step, g, s)
    sc = env.scores().cpu().numpy()
    rk = env.ranks().cpu().numpy()
    for g, o in enumerate(games):
        v = o.peek()
        assert list(sc[g]) == [v.players[p].score for p in range(4)]
    assert (rk == env.env.ranks()).all()
