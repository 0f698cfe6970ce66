"""Kernel time of k_policy alone: python tools/exp_policy.py [N_AGENTS BATCH] (default: two sizes)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cooperative_search_amd as cs
from tools.bench_policy import timed
sizes = ((3, 4096), (3, 65536)) if len(sys.argv) < 3 else ((int(sys.argv[1]), int(sys.argv[2])),)
out = {}
for n, B in sizes:
    args = cs.make_env_args("flight_easy", n_agents=n)
    env = cs.BatchedFlightEnv(args, batch=B, freeze_done=True)
    cs.apply_env_info(args, env)
    fused = cs.FusedAgents(args, B)
    obs = env.get_obs()
    out[B * n] = round(timed(lambda: fused.choose_action(obs), 300), 2)
print(json.dumps(out))
