import sys, os
sys.path.insert(0, "/root/repo")
from riichienv_amd import vecenv
vecenv.LIB_PATH = os.path.abspath(sys.argv[1])
n = int(sys.argv[2])
env = vecenv.VecRiichiEnv(n, game_mode=int(sys.argv[3]), seed=0)
env.reset()
print("reset ok", flush=True)
for i in range(3000):
    env.step_random(0xC0FFEE, 1, auto_reset=True)
    t = env.total_steps()
    if i % 100 == 0 or i < 5:
        print(i, t, flush=True)
print("done")
