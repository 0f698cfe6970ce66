"""What a fused auto-reset costs inside the rollout kernels: env-steps/s with time_limit = 1 (every env resets before
every step) against the default episode length.   python tools/exp_reset_cost.py [kernel=auto] [n=3] [B=4096]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import cooperative_search_amd as cs
kernel = sys.argv[1] if len(sys.argv) > 1 else "auto"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
B = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
T = 100
for tl in (200, 10, 2, 1):
    args = cs.make_env_args("flight_easy", n_agents=n)
    args.time_limit = tl
    env = cs.BatchedFlightEnv(args, batch=B, freeze_done=False, auto_reset=True, kernel=kernel)
    acts = torch.randint(0, 3, (T, B, n), dtype=torch.int32, device="cuda")
    out = env.rollout(acts)
    for _ in range(3):
        env.rollout(acts, out=out, update_views=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        env.rollout(acts, out=out, update_views=False)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (10 * T)
    print(f"time_limit {tl}: {us:.2f} us per step of {B} envs", flush=True)
