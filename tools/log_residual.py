"""Residual of the device's fp64 log / sqrt in the polar-gaussian target jitter of reset (flight_env_easy.py:107-108
through NumPy's legacy randn) against glibc (the C oracle): max |target coordinate difference| and how many of the
jittered coordinates differ at all, over B envs x several resets.  Integer outcomes are compared in the parity tests."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cooperative_search_amd as cs
from oracle import oracle as orc
B, n = 4096, 3
seeds = (np.arange(B) + 777).astype(np.uint32)
env = cs.BatchedFlightEnv(cs.make_env_args("flight_easy", n_agents=n), batch=B, seeds=seeds)
env.seed(seeds)
ob = orc.OracleBatch(orc.make_config(variant="flight_easy", n_agents=n), B, seeds)
worst, differing, total = 0.0, 0, 0
for r in range(5):
    env.reset(init=False)
    ob.reset(init=False, threads=8)
    tg = env.raw()["tgt"][:, :15].cpu().numpy()
    ref = np.stack([ob.env(b).targets()[0] for b in range(B)])
    d = np.abs(tg - ref)
    worst = max(worst, float(d.max()))
    differing += int((d > 0).sum())
    total += d.size
print(f"max |dtarget| = {worst:.3e}; {differing} of {total} coordinates differ ({100.0 * differing / total:.3f} %)")
