"""Debug: per-stage cycle counts of one wavefront of k_rollout (library built with -DCS_TIMELINE)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cooperative_search_amd as cs
n, B, T = int(os.environ.get("N", 3)), int(os.environ.get("B", 4096)), 64
args = cs.make_env_args("flight_easy", n_agents=n); args.time_limit = 10**9
env = cs.BatchedFlightEnv(args, batch=B, freeze_done=False, auto_reset=True, kernel="solo")
acts = torch.randint(0, 3, (T, B, n), dtype=torch.int32, device="cuda")
out = env.rollout(acts); out = env.rollout(acts, out=out, update_views=False)
torch.cuda.synchronize()
L = cs.lib.load()
buf = (C.c_ulonglong * (64 * 16))()
L.cs_debug_read_stamps.argtypes = [C.c_void_p]
assert L.cs_debug_read_stamps(buf) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(64, 16).astype(np.int64)
d = np.diff(st[:, :6], axis=1)
step = st[1:, 0] - st[:-1, 0]
names = ["entry->pre-kin", "kinematics", "detect", "post(prefetch)", "emit_wave"]
print("median cycles per stage:", {nm: int(np.median(d[5:60, i])) for i, nm in enumerate(names)})
print("median step-to-step:", int(np.median(step[5:60])), " (loop overhead = step - sum =", int(np.median(step[5:60]) - np.median(d[5:60].sum(1))), ")")
