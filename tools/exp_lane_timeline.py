"""Debug: per-stage cycle counts of wavefront 0 of k_rollout_lane (library built with -DCS_TIMELINE; see
tools/build_timeline.sh).  COOPSEARCH_LIB must point at that build."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cooperative_search_amd as cs
n, B, T = int(os.environ.get("N", 3)), int(os.environ.get("B", 262144)), 64
env = cs.BatchedFlightEnv(cs.make_env_args("flight_easy", n_agents=n), batch=B, freeze_done=False, auto_reset=True, kernel=os.environ.get("KERNEL", "lane"))
acts = torch.randint(0, 3, (T, B, n), dtype=torch.int32, device="cuda")
out = env.rollout(acts); out = env.rollout(acts, out=out, update_views=False)
torch.cuda.synchronize()
L = cs.lib.load()
buf = (C.c_ulonglong * (64 * 16))()
L.cs_debug_read_stamps.argtypes = [C.c_void_p]
assert L.cs_debug_read_stamps(buf) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(64, 16).astype(np.int64)
d = np.diff(st[:, :8], axis=1)
step = st[1:, 0] - st[:-1, 0]
names = ["reset", "kinematics", "emit-f+sensor tests", "draws", "reward+row", "refill+prefetch", "outputs(stores)"]
print("kernel", os.environ.get("KERNEL", "lane"), "n", n, "B", B)
print("median / mean cycles per stage (wave 0, steps 5..60):")
for i, nm in enumerate(names):
    print(f"  {nm:22s} {int(np.median(d[5:60, i])):7d} {int(d[5:60, i].mean()):7d}  max {int(d[5:60, i].max()):7d}")
print("median step-to-step:", int(np.median(step[5:60])), "mean", int(step[5:60].mean()))
