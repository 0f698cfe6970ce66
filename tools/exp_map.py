import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cooperative_search_amd as cs
B, n = 8192, 3
env = cs.BatchedFlightEnv(cs.make_env_args("flight", n_agents=n), batch=B, freeze_done=False, auto_reset=True)
acts = torch.randint(0, 3, (50, B, n), dtype=torch.int32, device="cuda")
for s in range(50): env.step(acts[s])
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for it in range(4):
    for s in range(50): env.step(acts[s])
e1.record(); torch.cuda.synchronize()
print("%.1f us/step" % (e0.elapsed_time(e1) * 1e3 / 200))
