"""Debug (-DCS_TIMELINE build): where a T-step k_rollout_duo launch spends its time outside the steady-state loop."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cooperative_search_amd as cs
n, B, T = int(os.environ.get("N", 3)), int(os.environ.get("B", 4096)), int(os.environ.get("T", 20))
env = cs.BatchedFlightEnv(cs.make_env_args("flight_easy", n_agents=n), batch=B, freeze_done=False, auto_reset=True, kernel="group")
acts = torch.randint(0, 3, (T, B, n), dtype=torch.int32, device="cuda")
out = env.rollout(acts)
for _ in range(3):
    out = env.rollout(acts, out=out, update_views=False)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); env.rollout(acts, out=out, update_views=False); e1.record(); torch.cuda.synchronize()
L = cs.lib.load()
buf = (C.c_ulonglong * (64 * 16))()
L.cs_debug_read_stamps.argtypes = [C.c_void_p]
assert L.cs_debug_read_stamps(buf) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(64, 16).astype(np.int64)
t0 = min(st[63, 3], st[63, 13])
print(f"launch (events) {e0.elapsed_time(e1) * 1e3:.1f} us; cycle counter units below (100 MHz = 10 ns?)")
print("K: entry", st[63, 3] - t0, "trig ready", st[63, 4] - t0, "loop s=0", st[0, 0] - t0, "s=1", st[1, 0] - t0, f"s={T-1}", st[T - 1, 0] - t0, "exit", st[63, 5] - t0)
print("D: entry", st[63, 13] - t0, "trig ready", st[63, 14] - t0, "loop s=0", st[0, 8] - t0, "s=1", st[1, 8] - t0, f"s={T-1}", st[T - 1, 8] - t0, "after last barrier", st[T - 1, 12] - t0, "exit", st[63, 15] - t0)
print("K step starts:", (st[:T, 0] - t0).tolist())
print("step: K produce, K wait | D slot, detect, emit, wait")
for s in range(min(T, 12)):
    print(s, st[s, 1] - st[s, 0], st[s, 2] - st[s, 1], "|", st[s, 9] - st[s, 8], st[s, 10] - st[s, 9], st[s, 11] - st[s, 10], st[s, 12] - st[s, 11])
