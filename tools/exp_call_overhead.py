"""Host-side cost of one env.rollout call and what an event-bracketed region of ONE 20-step launch pays on top of the
kernel: python tools/exp_call_overhead.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import cooperative_search_amd as cs
B, n, T = 4096, 3, 20
env = cs.BatchedFlightEnv(cs.make_env_args("flight_easy", n_agents=n), batch=B, freeze_done=False, auto_reset=True)
acts = torch.randint(0, 3, (T, B, n), dtype=torch.int32, device="cuda")
out = env.rollout(acts)
for _ in range(5):
    env.rollout(acts, out=out, update_views=False)
torch.cuda.synchronize()
# (a) enqueue cost: the GPU is kept busy, so the loop runs at the host's pace only if the host is the slower side
t0 = time.perf_counter()
for _ in range(500):
    env.rollout(acts, out=out, update_views=False)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"500 calls: host enqueue {1e6 * (t1 - t0) / 500:.1f} us per call, incl. drain {1e6 * (t2 - t0) / 500:.1f} us per call")


def region(k):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k):
        env.rollout(acts, out=out, update_views=False)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3


for k in (1, 2, 5, 10):
    xs = sorted(region(k) for _ in range(200))
    print(f"{k} launches per region: median {xs[100]:.1f} us, per launch {xs[100] / k:.1f} us")

# (c) how much of a 20-step launch is MT19937 top-up work (rows refreshed by a separate pre-pass before each launch)
ops = cs.lib.torch_ops()


def region_fresh():
    ops.mt_advance(env._cfg_t, env._blob, 600)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    env.rollout(acts, out=out, update_views=False)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3


xs = sorted(region_fresh() for _ in range(200))
print(f"one launch per region, rows topped up beforehand: median {xs[100]:.1f} us")
