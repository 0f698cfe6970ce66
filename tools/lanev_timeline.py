#!/usr/bin/env python3
"""Per-phase cycles of one wavefront-step of k_rollout_lanev<N> (workgroup 0, wavefront 0): a -DCS_TIMELINE build stamps the cycle
counter at the phase boundaries of the step loop (LANE_STAMP 0..6 in csrc/rollout_lanev.h); this reads the stamps of the first 64 steps
of a launch back (cs_debug_read_stamps) and prints the median length of every phase.
    tools/build_var.sh tl_n5 5 -DCS_TIMELINE;  COOPSEARCH_LIB=build/var/tl_n5.so python tools/lanev_timeline.py 5 65536
Phases: 0-1 reset block (target placement of the envs whose episode ended) | 1-2 kinematics (+ the refresh row's request) | 2-3 agents'
floats + sensor tests | 3-4 draws (miss walk) | 4-5 found flags, reward, termination | 5-6 MT19937 row refresh (twist + tape) + next
actions | 6-0' state deposit, copy-out, stores."""
import ctypes as C
import os
import statistics
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import cooperative_search_amd as cs

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
T = 100
env = cs.BatchedFlightEnv(cs.make_env_args("flight_easy", n_agents=n), batch=B, freeze_done=False, auto_reset=True, kernel="lanev")
acts = torch.randint(0, 3, (T, B, n), dtype=torch.int32, device="cuda")
out = env.rollout(acts)
for _ in range(3):
    env.rollout(acts, out=out, update_views=False)
torch.cuda.synchronize()
L = cs.lib.load()
buf = (C.c_ulonglong * (64 * 16))()
rows = []
for rep in range(5):
    env.rollout(acts, out=out, update_views=False)
    torch.cuda.synchronize()
    assert L.cs_debug_read_stamps(buf) == 0
    rows.append(np.array(buf[:], dtype=np.uint64).reshape(64, 16).astype(np.int64))
names = ["reset block", "kinematics (+ row request)", "agents' floats + sensor tests", "draws (miss walk)", "found / reward / termination",
         "row refresh + next actions", "state deposit, copy-out, stores"]
per = {k: [] for k in range(7)}
step = []
for st in rows:
    for s in range(2, 62):
        for k in range(6):
            per[k].append(int(st[s, k + 1] - st[s, k]))
        per[6].append(int(st[s + 1, 0] - st[s, 6]))
        step.append(int(st[s + 1, 0] - st[s, 0]))
sub = {k: [] for k in ("wait for the row + to LDS", "twist", "312 hit bits + tape", "on-the-spot top-ups + next actions' request")}
for st in rows:
    for s in range(2, 62):
        if st[s, 9] > st[s, 5] and st[s, 12] > st[s, 9]:   # a refresh ran in this step
            sub["wait for the row + to LDS"].append(int(st[s, 10] - st[s, 9]))
            sub["twist"].append(int(st[s, 11] - st[s, 10]))
            sub["312 hit bits + tape"].append(int(st[s, 12] - st[s, 11]))
            sub["on-the-spot top-ups + next actions' request"].append(int(st[s, 6] - st[s, 13]))
tot = statistics.median(step)
print(f"k_rollout_lanev<{n}>, {B} envs: median wavefront-step {tot} cycles ({len(step)} steps of wavefront 0)")
for k in range(7):
    m = statistics.median(per[k])
    print(f"  {names[k]:34s} {m:7.0f} cycles  {100.0 * m / tot:5.1f} %   (mean {statistics.fmean(per[k]):.0f})")
print("  inside the row refresh (steps in which one ran: %d of %d):" % (len(sub["twist"]), len(step)))
for k, v in sub.items():
    if v:
        print(f"    {k:44s} {statistics.median(v):7.0f} cycles  (mean {statistics.fmean(v):.0f})")
