import ctypes as C, os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np, torch
import cooperative_search_amd as cs
n, B, T = 3, 4096, 64
env = cs.BatchedFlightEnv(cs.make_env_args("flight_easy", n_agents=n), batch=B, freeze_done=False, auto_reset=True, kernel="od")
acts = torch.randint(0, 3, (T, B, n), dtype=torch.int32, device="cuda")
out = env.rollout(acts)
for rep in range(3):
    out = env.rollout(acts, out=out, update_views=False)
    torch.cuda.synchronize()
    L = cs.lib.load()
    buf = (C.c_ulonglong * (64 * 16))()
    L.cs_debug_read_stamps.argtypes = [C.c_void_p]
    assert L.cs_debug_read_stamps(buf) == 0
    st = np.frombuffer(buf, dtype=np.uint64).reshape(64, 16).astype(np.int64)
    d = st[1:, 8] - st[:-1, 8]
    print("D step-to-step, steps 0..15:", d[:16].tolist())
    print("   steps 16..63 median", int(np.median(d[16:])), "mean", int(d[16:].mean()), "max", int(d[16:].max()))
