"""Quick throughput probe of the rollout kernels: python tools/quick_lane.py [n] [kernel] [B ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import cooperative_search_amd as cs

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
kernel = sys.argv[2] if len(sys.argv) > 2 else "lane"
Bs = [int(x) for x in sys.argv[3:]] or [65536, 262144]
T = 100
for B in Bs:
    env = cs.BatchedFlightEnv(cs.make_env_args("flight_easy", n_agents=n), batch=B, freeze_done=False, auto_reset=True,
                              kernel=kernel)
    acts = torch.randint(0, 3, (T, B, n), dtype=torch.int32, device="cuda")
    out = env.rollout(acts, emit=os.environ.get("EMIT", "1") == "1")
    for _ in range(2):
        env.rollout(acts, out=out, update_views=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 6
    e0.record()
    for _ in range(reps):
        env.rollout(acts, out=out, update_views=False)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    alg = 36 * n + 12 * 15 + 6
    v = B * T / (ms / 1e3)
    print(f"n={n} {kernel} B={B}: {ms*1e3/T:.2f} us/step, {v:.3e} env-steps/s, {alg*v/8e12*100:.1f} % of 8 TB/s", flush=True)
    del env, acts, out
    torch.cuda.empty_cache()
