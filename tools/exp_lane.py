"""A few launches of one rollout kernel for profiling: python tools/exp_lane.py n kernel B [launches]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import cooperative_search_amd as cs
n, kernel, B = int(sys.argv[1]), sys.argv[2], int(sys.argv[3])
launches = int(sys.argv[4]) if len(sys.argv) > 4 else 4
T = 100
env = cs.BatchedFlightEnv(cs.make_env_args("flight_easy", n_agents=n), batch=B, freeze_done=False, auto_reset=True, kernel=kernel)
acts = torch.randint(0, 3, (T, B, n), dtype=torch.int32, device="cuda")
out = env.rollout(acts)
for _ in range(launches - 1):
    env.rollout(acts, out=out, update_views=False)
torch.cuda.synchronize()
