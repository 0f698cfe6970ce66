#!/usr/bin/env python3
"""flight closed loop (conv front end + recurrent network + step + map update per step), B envs: env-steps/s; run it under
`rocprofv3 --kernel-trace --stats` for the per-kernel split.   python tools/exp_flight_closed.py [B=8192] [steps=400]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cooperative_search_amd as cs
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
dev = torch.device("cuda", 0)
args = cs.make_env_args("flight", n_agents=3)
env = cs.BatchedFlightEnv(args, batch=B, device=dev, freeze_done=False, auto_reset=True)
cs.apply_env_info(args, env)
torch.manual_seed(0)
agents = cs.FusedAgents(args, B, device=dev)
out = env.rollout_policy(agents, 100, emit=False)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps // 100):
        env.rollout_policy(agents, 100, emit=False, out=out, update_views=False)
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e-3)
ts.sort()
t = ts[len(ts) // 2]
print(f"flight closed loop B={B}: {t / steps * 1e6:.1f} us per step, {B * steps / t:.3e} env-steps/s")
