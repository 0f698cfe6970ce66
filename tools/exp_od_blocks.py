#!/usr/bin/env python3
"""Where a T-step k_rollout_od launch spends its time OUTSIDE the step loop (timeline build, -DCS_TIMELINE): per workgroup,
cycle stamps at entry / loop start / loop end / exit of both wavefronts.  COOPSEARCH_LIB=<timeline build>."""
import argparse, ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import cooperative_search_amd as cs
from cooperative_search_amd import _lib
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=3)
ap.add_argument("--B", type=int, default=4096)
ap.add_argument("--T", type=int, default=20)
ap.add_argument("--reps", type=int, default=12)
a = ap.parse_args()
L = _lib.load()
L.cs_debug_read_blk.argtypes = [C.c_void_p]
dev = torch.device("cuda", 0)
env = cs.BatchedFlightEnv(cs.make_env_args("flight_easy", n_agents=a.n), batch=a.B, device=dev, freeze_done=False, auto_reset=True, kernel=os.environ.get("KERNEL", "ode"))
acts = torch.randint(0, 3, (a.T, a.B, a.n), dtype=torch.int32, device=dev)
out = env.rollout(acts, update_views=False)
nb = a.B // 8
rows = []
for rep in range(a.reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    env.rollout(acts, out=out, update_views=False)
    e1.record()
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * (1024 * 8))()
    assert L.cs_debug_read_blk(buf) == 0
    st = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 8).astype(np.int64)[:nb]
    t0 = st[:, [0, 4]].min()
    end = st[:, [3, 7]].max()
    k0, k1, k2, k3, d0, d1, d2, d3 = [st[:, i] - t0 for i in range(8)]
    q = lambda v: "%6d %6d %6d" % (np.percentile(v, 10), np.median(v), v.max())
    print(f"rep {rep}: event {e0.elapsed_time(e1)*1e3:.1f} us, first entry -> last exit {end - t0} cycles")
    print("   entry after first (p10 med max):      K", q(k0), "  D", q(d0))
    print("   prologue (entry -> loop):             K", q(k1 - k0), "  D", q(d1 - d0))
    print("   loop:                                 K", q(k2 - k1), "  D", q(d2 - d1))
    print("   epilogue (loop end -> exit):          K", q(k3 - k2), "  D", q(d3 - d2))
    print("   exit after first entry:               K", q(k3), "  D", q(d3))
    sp = (C.c_uint * (1024 * 4))()
    L.cs_debug_read_spin.argtypes = [C.c_void_p]
    assert L.cs_debug_read_spin(sp) == 0
    spn = np.frombuffer(sp, dtype=np.uint32).reshape(1024, 4)[:nb]
    print("   polls spent waiting per launch (p10 med max):  K for a slot", q(spn[:, 0]), "  D for K", q(spn[:, 1]), "  E for D", q(spn[:, 2]))
    print("   D loop of the slowest / median workgroup per step:", (d2 - d1).max() // a.T, int(np.median(d2 - d1)) // a.T)
