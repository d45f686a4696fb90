import hashlib, ctypes as C, numpy as np
from riichienv_amd import vecenv, abi
from riichienv_amd.shard import game_seed
from oracle import oracle
R = abi.RULE_TENHOU | abi.RULE_REFERENCE_RNG
env = vecenv.VecRiichiEnv(4, game_mode=2, seed=42, rule_bits=R)
g = oracle.Game(game_mode=2, seed=game_seed(42, 0), rule_bits=R)
print("dev", env.wall_digest(0)); print("orc", g.wall_meta())
w, salt, dg, _ = oracle.reference_wall(game_seed(game_seed(42, 0), 0))
print("ref", salt, dg, hashlib.sha256(salt.encode() + w.tobytes()).hexdigest())
v = env.peek(0)
print("dev wall", list(v.wall[:v.wall_len])[:20], v.wall_len)
print("ref wall", list(w[::-1][:20]))
walls = np.tile(np.arange(136, dtype=np.uint8), (4, 1))
env.reset(walls=walls)
s2, d2 = env.wall_digest(0)
print("inj", s2, d2, hashlib.sha256(s2.encode() + bytes(range(136))).hexdigest())
for k in range(0, 153, 8):
    print(k, hashlib.sha256(s2.encode() + bytes(range(136))[:k]).hexdigest()[:16])
