"""Debug: per-phase cycle counts of block 0 of k_policy_split (library built with -DPOL_TIMELINE)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cooperative_search_amd as cs
n, B = 3, int(os.environ.get("B", 65536))
args = cs.make_env_args("flight_easy", n_agents=n)
env = cs.BatchedFlightEnv(args, batch=B, freeze_done=True)
cs.apply_env_info(args, env)
fused = cs.FusedAgents(args, B)
obs = env.get_obs()
for _ in range(3):
    fused.choose_action(obs)
torch.cuda.synchronize()
L = cs.lib.load()
buf = (C.c_ulonglong * (32 * 8))()
L.cs_policy_debug_read_stamps.argtypes = [C.c_void_p]
assert L.cs_policy_debug_read_stamps(buf) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(32, 8).astype(np.int64)
k = 20 if B >= 65536 else 1
d = np.diff(st[:k, :7], axis=1)
names = ["stage->b1", "fc1->b2", "gru->b3", "fc2a->b4", "fc2b->b5", "epilogue"]
print("median cycles per phase:", {nm: int(np.median(d[1:, i])) if k > 2 else int(d[0, i]) for i, nm in enumerate(names)})
if k > 2:
    print("iteration to iteration:", int(np.median(np.diff(st[1:k, 0]))))
