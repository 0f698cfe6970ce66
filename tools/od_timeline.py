#!/usr/bin/env python3
"""Per-phase cycles of the pair kernels' K and D wavefronts (workgroup 0) from a -DCS_TIMELINE build: stamps at the phase boundaries
of both roles' step loops (DUO_STAMP / KIN_STAMP in csrc/rollout_od.h, rollout_oct.h), first 64 steps of a launch, median over launches.
    tools/build_var.sh tlo_n5 5 -DCS_TIMELINE;  COOPSEARCH_LIB=build/var/tlo_n5.so python tools/od_timeline.py 5 od 16384
K: 0 loop top | 2 flow control passed | 3 positions read, yaw updated | 4 trig done | 5 stages done | 6 kinematics returned | 1 published
D: 8 loop top | 9 reset block / refresh adoption done | (wait for K's step) | 10 ring read + detection + reward done | 11 outputs read from
LDS, requests issued, stores issued | 12 d_steps posted."""
import ctypes as C
import os
import statistics
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import cooperative_search_amd as cs

n = int(sys.argv[1]); kernel = sys.argv[2]; B = int(sys.argv[3])
T = 100
env = cs.BatchedFlightEnv(cs.make_env_args("flight_easy", n_agents=n), batch=B, freeze_done=False, auto_reset=True, kernel=kernel)
acts = torch.randint(0, 3, (T, B, n), dtype=torch.int32, device="cuda")
out = env.rollout(acts)
for _ in range(3):
    env.rollout(acts, out=out, update_views=False)
torch.cuda.synchronize()
L = cs.lib.load()
buf = (C.c_ulonglong * (64 * 16))()
rows = []
for rep in range(8):
    env.rollout(acts, out=out, update_views=False)
    torch.cuda.synchronize()
    assert L.cs_debug_read_stamps(buf) == 0
    rows.append(np.array(buf[:], dtype=np.uint64).reshape(64, 16).astype(np.int64))
segs = {"K: flow control": (0, 2), "K: read positions, yaw": (2, 3), "K: trig": (3, 4), "K: stages + wall": (4, 5), "K: rest of kinematics": (5, 6),
        "K: publish": (6, 1), "D: reset / refresh adoption": (8, 9), "D: wait for K + ring read + detection + reward": (9, 10),
        "D: record / outputs / requests / stores": (10, 11), "D: post": (11, 12)}
val = {k: [] for k in segs}
kstep, dstep = [], []
for st in rows:
    for s in range(3, 60):
        for k, (a, b) in segs.items():
            d = int(st[s, b] - st[s, a])
            if 0 <= d < 200000:
                val[k].append(d)
        kstep.append(int(st[s + 1, 0] - st[s, 0]))
        dstep.append(int(st[s + 1, 8] - st[s, 8]))
rs = {"D reset: placement (oct_place_targets)": [], "D reset: state + top-up + barrier": [], "D reset: near test / reset-time pass / top-up": []}
nres = 0
for st in rows:
    for s in range(3, 60):
        if st[s, 13] > st[s, 8] and st[s, 14] > st[s, 13] and st[s, 9] > st[s, 15] > st[s, 14]:   # a reset ran in this step
            nres += 1
            rs["D reset: placement (oct_place_targets)"].append(int(st[s, 14] - st[s, 13]))
            rs["D reset: state + top-up + barrier"].append(int(st[s, 15] - st[s, 14]))
            rs["D reset: near test / reset-time pass / top-up"].append(int(st[s, 9] - st[s, 15]))
print(f"{kernel}<{n}>, {B} envs: K step {statistics.median(kstep)} cycles, D step {statistics.median(dstep)} cycles (medians, workgroup 0)")
for k, v in val.items():
    if v:
        print(f"  {k:52s} {statistics.median(v):7.0f}  (mean {statistics.fmean(v):.0f})")
print(f"  steps with a reset: {nres} of {len(dstep)}")
for k, v in rs.items():
    if v:
        print(f"  {k:52s} {statistics.median(v):7.0f}  (mean {statistics.fmean(v):.0f})")
