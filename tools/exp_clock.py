"""The shader clock a rollout kernel actually runs at: wavefront 0 stamps s_memtime (shader clock) and s_memrealtime (constant
100 MHz) at the top of every step (library built with -DCS_TIMELINE, tools/build_timeline.sh; COOPSEARCH_LIB points at it).
    N=5 B=262144 KERNEL=lane python tools/exp_clock.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cooperative_search_amd as cs
n, B, T = int(os.environ.get("N", 3)), int(os.environ.get("B", 262144)), 64
kernel = os.environ.get("KERNEL", "lane")
env = cs.BatchedFlightEnv(cs.make_env_args("flight_easy", n_agents=n), batch=B, freeze_done=False, auto_reset=True, kernel=kernel)
acts = torch.randint(0, 3, (T, B, n), dtype=torch.int32, device="cuda")
out = env.rollout(acts)
for _ in range(3):
    out = env.rollout(acts, out=out, update_views=False)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); env.rollout(acts, out=out, update_views=False); e1.record(); torch.cuda.synchronize()
call_us = e0.elapsed_time(e1) * 1e3
L = cs.lib.load()
buf = (C.c_ulonglong * (64 * 16))()
L.cs_debug_read_stamps.argtypes = [C.c_void_p]
assert L.cs_debug_read_stamps(buf) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(64, 16).astype(np.int64)
cyc = st[60, 0] - st[4, 0]
real = st[60, 8] - st[4, 8]
print(f"{kernel}<{n}> B={B}: {call_us / T:.2f} us per step (instrumented build); wave 0, steps 4..60: {cyc} shader cycles in "
      f"{real} ticks of the 100 MHz counter -> {cyc / max(real, 1) * 100:.0f} MHz; {cyc / 56:.0f} cycles per wave-step")
