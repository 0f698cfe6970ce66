"""Debug: per-stage cycle counts of the K / D wavefront pair 0 of block 0 of k_rollout_duo (-DCS_TIMELINE build)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cooperative_search_amd as cs
n, B, T = int(os.environ.get("N", 3)), int(os.environ.get("B", 4096)), 64
env = cs.BatchedFlightEnv(cs.make_env_args("flight_easy", n_agents=n), batch=B, freeze_done=False, auto_reset=True, kernel="group")
acts = torch.randint(0, 3, (T, B, n), dtype=torch.int32, device="cuda")
out = env.rollout(acts); out = env.rollout(acts, out=out, update_views=False)
torch.cuda.synchronize()
L = cs.lib.load()
buf = (C.c_ulonglong * (64 * 16))()
L.cs_debug_read_stamps.argtypes = [C.c_void_p]
assert L.cs_debug_read_stamps(buf) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(64, 16).astype(np.int64)
r = slice(5, 60)
med = lambda x: int(np.median(x[r]))
print("K: produce", med(st[:, 1] - st[:, 0]), " barrier wait", med(st[:, 2] - st[:, 1]), " step-to-step", med(np.diff(st[:, 0])[4:59]))
print("D: slot/reset", med(st[:, 9] - st[:, 8]), " detect", med(st[:, 10] - st[:, 9]), " emit", med(st[:, 11] - st[:, 10]),
      " barrier wait", med(st[:, 12] - st[:, 11]), " step-to-step", med(np.diff(st[:, 8])[4:59]))
print("K start - D start (same step):", med(st[:, 0] - st[:, 8]))
