"""Debug: the rare events of k_rollout_od's D role (workgroup 0, -DCS_TIMELINE build): cycles of the step-top phase (row
refresh finish, urgent top-up, reset) wherever it is not the plain check.  env: N, B."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cooperative_search_amd as cs
n, B, T = int(os.environ.get("N", 3)), int(os.environ.get("B", 4096)), 64
args = cs.make_env_args("flight_easy", n_agents=n)
args.time_limit = int(os.environ.get("TL", 200))
env = cs.BatchedFlightEnv(args, batch=B, freeze_done=False, auto_reset=True, kernel="od")
acts = torch.randint(0, 3, (T, B, n), dtype=torch.int32, device="cuda")
out = env.rollout(acts)
L = cs.lib.load()
L.cs_debug_read_stamps.argtypes = [C.c_void_p]
ev = []
for rep in range(12):
    out = env.rollout(acts, out=out, update_views=False)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * (64 * 16))()
    assert L.cs_debug_read_stamps(buf) == 0
    st = np.frombuffer(buf, dtype=np.uint64).reshape(64, 16).astype(np.int64)
    top = st[:, 9] - st[:, 8]
    ev += [int(v) for v in top if v > 700]
    r = st[63, 3:8]
    if r[4] > r[0]:
        print("  last reset_targets batch: tables", int(r[1] - r[0]), " words", int(r[2] - r[1]), " polar", int(r[3] - r[2]), " select+commit", int(r[4] - r[3]))
    for i in range(64):
        if top[i] > 7000 and st[i, 13] > st[i, 8]:
            print("  reset at step", i, ": before", int(st[i, 13] - st[i, 8]), " rounds", int(st[i, 14] - st[i, 13]), " top-up check", int(st[i, 15] - st[i, 14]),
                  " reset-time pass + rest", int(st[i, 9] - st[i, 15]))
    base = int(np.median(top))
print(f"k_rollout_od<{n}> time_limit {args.time_limit}: plain step-top check {base} cycles; events (cycles): {sorted(ev)}")
