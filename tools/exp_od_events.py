"""Debug: what D's rare events cost in the c2 pair kernel (library built with -DCS_TIMELINE; COOPSEARCH_LIB must point at that build).
Ten launches of 64 steps; workgroup 0's stamps after each.  A step whose top phase (stamp 8 -> 9) is long is a reset when stamps
13..15 (reset begin / placement done / barrier passed) lie inside it, otherwise a row top-up."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cooperative_search_amd as cs
n, B, T = int(os.environ.get("N", 3)), int(os.environ.get("B", 4096)), 64
env = cs.BatchedFlightEnv(cs.make_env_args("flight_easy", n_agents=n), batch=B, freeze_done=False, auto_reset=True, kernel=os.environ.get("KERNEL", "ode"))
L = cs.lib.load()
L.cs_debug_read_stamps.argtypes = [C.c_void_p]
resets, tops, plain, kfix = [], [], [], []
out = None
for rep in range(12):
    acts = torch.randint(0, 3, (T, B, n), dtype=torch.int32, device="cuda")
    out = env.rollout(acts, out=out, update_views=False) if out is not None else env.rollout(acts)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * (64 * 16))()
    assert L.cs_debug_read_stamps(buf) == 0
    st = np.frombuffer(buf, dtype=np.uint64).reshape(64, 16).astype(np.int64)
    if rep < 2:
        continue
    for i in range(3, 62):
        top = st[i, 9] - st[i, 8]
        if top > 1500:
            if st[i, 8] <= st[i, 13] <= st[i, 14] <= st[i, 15] <= st[i, 9]:
                resets.append((st[i, 13] - st[i, 8], st[i, 14] - st[i, 13], st[i, 15] - st[i, 14], st[i, 9] - st[i, 15], top))
            else:
                tops.append(top)
        else:
            plain.append(top)
        kw = st[i, 2] - st[i, 0]
        if kw > 1500:
            kfix.append(kw)
r = np.array(resets) if resets else np.zeros((0, 5))
print(f"k_rollout_od<{n}> B={B}: workgroup 0, {10 * 59} steps")
print(f"  plain steps: top phase median {int(np.median(plain))} cycles")
print(f"  row top-ups: {len(tops)}  median {int(np.median(tops)) if tops else 0}  max {max(tops) if tops else 0}")
print(f"  resets: {len(r)}")
if len(r):
    for k, nm in enumerate(["before (pending top-up, flags)", "placement (+ wait for E before the tile)", "bookkeeping + barrier", "near test / detection pass", "total"]):
        print(f"    {nm:42s} median {int(np.median(r[:, k])):6d}  mean {int(r[:, k].mean()):6d}  max {int(r[:, k].max()):6d}")
print(f"  K waits > 1500 cycles (fix + redo, or a full ring): {len(kfix)}  median {int(np.median(kfix)) if kfix else 0}  max {max(kfix) if kfix else 0}")
