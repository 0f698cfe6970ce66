"""A known number of env-steps of one workload, for profiling passes (rocprofv3 --pmc / --kernel-trace):
    python tools/exp_workload.py <flight_easy|flight> <n_agents> <kernel> <B> <rollout|step> [launches=4] [T=100]
Every "launch" is T steps of all B envs (one cs_rollout call, or T cs_step calls); prints the env-steps executed."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import cooperative_search_amd as cs
env_name, n, kernel, B, mode = sys.argv[1], int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), sys.argv[5]
launches = int(sys.argv[6]) if len(sys.argv) > 6 else 4
T = int(sys.argv[7]) if len(sys.argv) > 7 else 100
env = cs.BatchedFlightEnv(cs.make_env_args(env_name, n_agents=n), batch=B, freeze_done=False, auto_reset=True, kernel=kernel)
acts = torch.randint(0, 3, (T, B, n), dtype=torch.int32, device="cuda")
out = None
for _ in range(launches):
    if mode == "rollout":
        out = env.rollout(acts, out=out, update_views=False)
    else:
        for t in range(T):
            env.step(acts[t])
torch.cuda.synchronize()
print("env_steps", launches * T * B)
