import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cooperative_search_amd as cs
from tools.bench_policy import timed
out = {}
for n, B in ((3, 4096), (3, 65536)):
    args = cs.make_env_args("flight_easy", n_agents=n)
    env = cs.BatchedFlightEnv(args, batch=B, freeze_done=True)
    cs.apply_env_info(args, env)
    fused = cs.FusedAgents(args, B)
    obs = env.get_obs()
    out[B * n] = round(timed(lambda: fused.choose_action(obs), 300), 2)
print(os.environ.get("COOPSEARCH_LIB", "default").split("/")[-1], os.environ.get("CS_POLICY_M", "auto"), json.dumps(out))
