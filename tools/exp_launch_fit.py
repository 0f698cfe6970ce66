"""Fixed cost of a rollout launch: time per launch (10 back-to-back launches per region) against steps per launch.
    python tools/exp_launch_fit.py [kernel=auto] [n=3] [B=4096]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import cooperative_search_amd as cs
kernel = sys.argv[1] if len(sys.argv) > 1 else "auto"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
B = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
env = cs.BatchedFlightEnv(cs.make_env_args("flight_easy", n_agents=n), batch=B, freeze_done=False, auto_reset=True, kernel=kernel)
Ts, us = [1, 2, 5, 10, 20, 50, 100], []
for T in Ts:
    acts = torch.randint(0, 3, (T, B, n), dtype=torch.int32, device="cuda")
    out = env.rollout(acts)
    for _ in range(3):
        env.rollout(acts, out=out, update_views=False)
    xs = []
    for _ in range(60):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            env.rollout(acts, out=out, update_views=False)
        e1.record()
        torch.cuda.synchronize()
        xs.append(e0.elapsed_time(e1) * 100)
    us.append(float(np.median(xs)))
    print(f"T={T}: {us[-1]:.1f} us per launch", flush=True)
k, c = np.polyfit(Ts[2:], us[2:], 1)
print(f"fit (T >= 5): {k:.3f} us per step + {c:.1f} us per launch")
