"""Debug: per-stage cycle counts of workgroup 0 of k_rollout_od (library built with -DCS_TIMELINE).  env: N, B."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cooperative_search_amd as cs
n, B, T = int(os.environ.get("N", 3)), int(os.environ.get("B", 4096)), 64
env = cs.BatchedFlightEnv(cs.make_env_args("flight_easy", n_agents=n), batch=B, freeze_done=False, auto_reset=True, kernel=os.environ.get("KERNEL", "ode"))
acts = torch.randint(0, 3, (T, B, n), dtype=torch.int32, device="cuda")
out = env.rollout(acts); out = env.rollout(acts, out=out, update_views=False)
torch.cuda.synchronize()
L = cs.lib.load()
buf = (C.c_ulonglong * (64 * 16))()
L.cs_debug_read_stamps.argtypes = [C.c_void_p]
assert L.cs_debug_read_stamps(buf) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(64, 16).astype(np.int64)
med = lambda a: int(np.median(a[5:60]))
print(f"k_rollout_od<{n}> B={B}, workgroup 0, median cycles over steps 5..60")
print("  K: wait for a free slot", med(st[:, 2] - st[:, 0]), " produce(s)", med(st[:, 1] - st[:, 2]), " step-to-step", med(st[1:, 0] - st[:-1, 0]))
print("  K inside produce: entry->trig", med(st[:, 3] - st[:, 2]), " trig pair", med(st[:, 4] - st[:, 3]), " fast path / stages", med(st[:, 5] - st[:, 4]),
      " selects+ballot", med(st[:, 6] - st[:, 5]), " publish (LDS)", med(st[:, 1] - st[:, 6]))
print("  D: top-up/reset", med(st[:, 9] - st[:, 8]), " ring+detect", med(st[:, 10] - st[:, 9]), " flags+rows+stores", med(st[:, 11] - st[:, 10]),
      " publish", med(st[:, 12] - st[:, 11]), " step-to-step", med(st[1:, 8] - st[:-1, 8]))
mean = lambda a: int(np.mean(a[5:60]))
print("  MEANS  K: wait+fix", mean(st[:, 2] - st[:, 0]), " produce", mean(st[:, 1] - st[:, 2]), " step-to-step", mean(st[1:, 0] - st[:-1, 0]),
      "| D: top", mean(st[:, 9] - st[:, 8]), " ring+detect", mean(st[:, 10] - st[:, 9]), " rows+stores", mean(st[:, 11] - st[:, 10]),
      " publish(+ack wait)", mean(st[:, 12] - st[:, 11]), " step-to-step", mean(st[1:, 8] - st[:-1, 8]))
kw = st[5:60, 2] - st[5:60, 0]
print("  K wait+fix per step:", [int(v) for v in kw])
dw = st[5:60, 10] - st[5:60, 9]
print("  D ring+detect per step:", [int(v) for v in dw])
dt = st[5:60, 9] - st[5:60, 8]
print("  D top per step:", [int(v) for v in dt])
dp = st[5:60, 12] - st[5:60, 11]
print("  D publish per step:", [int(v) for v in dp])
# resets (steps whose top-up / reset phase is long): placement (incl. E3's wait for E before the tile is touched) / bookkeeping + barrier /
# near test + reset-time detection pass
ev = [i for i in range(5, 60) if st[i, 9] - st[i, 8] > 2000 and st[i, 13] >= st[i, 8] and st[i, 15] <= st[i, 9]]
for i in ev:
    print(f"  reset at D step {i}: before {st[i, 13] - st[i, 8]}  placement {st[i, 14] - st[i, 13]}  top-up + barrier {st[i, 15] - st[i, 14]}"
          f"  near test / detection pass {st[i, 9] - st[i, 15]}  total {st[i, 9] - st[i, 8]}")
