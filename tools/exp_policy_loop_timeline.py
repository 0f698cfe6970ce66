"""Debug: per-phase cycle counts of wavefront 0 / workgroup 0 of the fused closed-loop kernel k_rollout_policy (library built with
-DCS_TIMELINE, tools/build_var.sh <name> 3 -DCS_TIMELINE; COOPSEARCH_LIB must point at that build)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cooperative_search_amd as cs
from cooperative_search_amd.agents import FusedAgents
n, B, T = int(os.environ.get("N", 3)), int(os.environ.get("B", 4096)), 64
args = cs.make_env_args("flight_easy", n_agents=n)
env = cs.BatchedFlightEnv(args, batch=B, freeze_done=False, auto_reset=True)
cs.apply_env_info(args, env)
fused = FusedAgents(args, B)
fused.init_hidden()
for _ in range(3):
    env.rollout_policy(fused, T, epsilon=0.05, evaluate=False)
torch.cuda.synchronize()
L = cs.lib.load()
buf = (C.c_ulonglong * (64 * 16))()
L.cs_debug_read_stamps.argtypes = [C.c_void_p]
assert L.cs_debug_read_stamps(buf) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(64, 16).astype(np.int64)
order = [6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 0, 1, 2, 3, 4, 5]
names = ["wait barrier 1 (loop top)", "fc1 + store h1", "wait barrier 2", "GRU + stores", "wait barrier 3", "fc2 + store f",
         "wait barrier 4", "fc3 + select (wavefront 0: tile 0)", "wait barrier 5", "actions from LDS, flush loads", "reset check",
         "kinematics", "detect + draws", "bookkeeping, prefetch", "flush stores, deposit", "obs columns -> x (to next loop top)"]
seq = st[:, order]
d = np.diff(seq, axis=1)
last = np.concatenate([seq[1:, 0] - seq[:-1, -1], [0]])
print("k_rollout_policy n", n, "B", B, ": median / mean / max cycles per phase (wavefront 0 of workgroup 0, steps 5..60)")
tot = 0
for i, nm in enumerate(names):
    col = d[5:60, i] if i < 15 else last[5:60]
    tot += np.median(col)
    print(f"  {nm:40s} {int(np.median(col)):7d} {int(col.mean()):7d} {int(col.max()):7d}")
step = seq[1:, 0] - seq[:-1, 0]
print("median step-to-step:", int(np.median(step[5:60])), "mean", int(step[5:60].mean()), " sum of medians", int(tot))
