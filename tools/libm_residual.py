"""How often does the reference's arithmetic (libm sin/cos/pow: oracle default mode, bit-pinned to the golden traces)
give a different integer outcome than the HIP path's arithmetic (correctly rounded trig, x*x: oracle HIP-equivalent
mode, bit-identical to the kernels)?  Runs both on the same seeds/actions and counts episodes whose reward or
terminated sequence differs.  CPU only.   python tools/libm_residual.py [episodes] [n_agents]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as orc


def count_divergent(n_agents, episodes, steps=200, seed0=123456, threads=8, agent_mode=0):
    cfg = orc.make_config(n_agents=n_agents, agent_mode=agent_mode)
    seeds = (np.arange(episodes, dtype=np.uint64) + seed0).astype(np.uint32)
    a, b = orc.OracleBatch(cfg, episodes, seeds), orc.OracleBatch(cfg, episodes, seeds)
    try:
        orc.set_trig_mode(0); orc.set_exact_pow(True); a.reset(init=True, threads=threads)
        orc.set_trig_mode(1); orc.set_exact_pow(False); b.reset(init=True, threads=threads)
        rng = np.random.RandomState(n_agents + 17 * agent_mode)
        diverged = np.zeros(episodes, dtype=bool)
        for _ in range(steps):
            act = rng.randint(0, 3, size=(episodes, n_agents)).astype(np.int32)
            orc.set_trig_mode(0); orc.set_exact_pow(True)
            ra, ta, _ = a.step(act, freeze_done=True, threads=threads, emit=False)
            ra, ta = ra.copy(), ta.copy()
            orc.set_trig_mode(1); orc.set_exact_pow(False)
            rb, tb, _ = b.step(act, freeze_done=True, threads=threads, emit=False)
            diverged |= (ra != rb) | (ta != tb)
    finally:
        orc.set_trig_mode(0); orc.set_exact_pow(True)
    return int(diverged.sum())


if __name__ == "__main__":
    episodes = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    for n in ([int(sys.argv[2])] if len(sys.argv) > 2 else [3, 5]):
        t0 = time.time()
        d = count_divergent(n, episodes)
        print(f"n_agents={n}: {d} of {episodes} episodes (x200 steps) diverge between libm and correctly-rounded arithmetic "
              f"({time.time()-t0:.0f} s)")
