"""Debug: per-stage cycle counts of wavefront 0 of k_rollout_oct (library built with -DCS_TIMELINE; see
tools/build_timeline.sh).  COOPSEARCH_LIB must point at that build.  env: N (agents), B (envs)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cooperative_search_amd as cs
n, B, T = int(os.environ.get("N", 3)), int(os.environ.get("B", 8192)), 64
env = cs.BatchedFlightEnv(cs.make_env_args("flight_easy", n_agents=n), batch=B, freeze_done=False, auto_reset=True, kernel="oct")
acts = torch.randint(0, 3, (T, B, n), dtype=torch.int32, device="cuda")
out = env.rollout(acts); out = env.rollout(acts, out=out, update_views=False)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); env.rollout(acts, out=out, update_views=False); e1.record(); torch.cuda.synchronize()
call_us = e0.elapsed_time(e1) * 1e3
L = cs.lib.load()
buf = (C.c_ulonglong * (64 * 16))()
L.cs_debug_read_stamps.argtypes = [C.c_void_p]
assert L.cs_debug_read_stamps(buf) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(64, 16).astype(np.int64)
d = np.diff(st[:, :8], axis=1)
step = st[1:, 0] - st[:-1, 0]
names = ["prefetch + reset", "kinematics", "pos/obs to LDS", "detection", "flags to LDS", "top-up", "outputs (stores)"]
print(f"k_rollout_oct<{n}> B={B}: median / mean cycles per stage (wave 0, steps 5..60):")
for i, nm in enumerate(names):
    print(f"  {nm:22s} {int(np.median(d[5:60, i])):7d} {int(d[5:60, i].mean()):7d}  max {int(d[5:60, i].max()):7d}")
print("median step-to-step:", int(np.median(step[5:60])), "mean", int(step[5:60].mean()))
span = int(st[63, 7] - st[0, 0])
print(f"wave 0: {span} counter ticks over the {T} steps; the call took {call_us:.1f} us by HIP events -> "
      f"{span / call_us:.0f} ticks per us if wave 0 spans the call (s_memtime tick rate)")
