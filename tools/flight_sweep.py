"""flight (probability-map variant), cs_rollout (k_step, then k_flight_pipe: the map sweep of step t beside step t + 1):
    python tools/flight_sweep.py batch      3 agents, B = 2048 .. 65536: the maps of a batch go from 20 MB to 655 MB, i.e. from well
                                            inside the 256 MiB Infinity Cache to well outside it
    python tools/flight_sweep.py teams      B = 8192, 1 .. 8 agents
One byte model everywhere -- bench.py's (algorithmic_bytes_per_env_step: n observation copies of the map written, the map READ once,
state row, the flight_easy remainder; the data-dependent write-back of changed cells not counted): 40 535 B per env-step at 3a15t.
Output buffers are allocated once and reused (HIP-event time of `reps` T-step calls)."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import cooperative_search_amd as cs
import bench


def measure(n, B, T):
    env = cs.BatchedFlightEnv(cs.make_env_args("flight", n_agents=n), batch=B, freeze_done=False, auto_reset=True)
    acts = torch.randint(0, 3, (T, B, n), dtype=torch.int32, device="cuda")
    out = env.rollout(acts)
    env.rollout(acts, out=out, update_views=False)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        env.rollout(acts, out=out, update_views=False)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / T)
    del env, acts, out
    torch.cuda.empty_cache()
    return statistics.median(ts), min(ts), max(ts)


what = sys.argv[1] if len(sys.argv) > 1 else "batch"
if what == "batch":
    print("| envs | maps of the batch | us per step (median of 5 calls; min .. max) | env-steps/s | algorithmic GB/s | of 8 TB/s |")
    print("|---|---|---|---|---|---|")
    for B in (2048, 4096, 8192, 16384, 32768, 65536):
        T = 40 if B <= 16384 else (20 if B == 32768 else 10)     # obs table: T x B x 30 KB (65536 envs x 10 steps = 19.7 GB)
        us, lo, hi = measure(3, B, T)
        alg = bench.algorithmic_bytes_per_env_step("flight", 3, 15, "rollout")
        gbs = alg * B / us / 1e3
        print(f"| {B} | {B * 10000 / 1e6:.0f} MB | {us:.1f} ({lo:.1f} .. {hi:.1f}) | {B / us * 1e6:.3e} | {gbs:.0f} | {gbs / 80:.1f} % |", flush=True)
else:
    B = int(os.environ.get("B", 8192))
    print("| agents | algorithmic B per env-step | us per step | env-steps/s | algorithmic GB/s | of 8 TB/s |")
    print("|---|---|---|---|---|---|")
    for n in (1, 2, 3, 4, 5, 6, 8):
        us, lo, hi = measure(n, B, 40)
        alg = bench.algorithmic_bytes_per_env_step("flight", n, 15, "rollout")
        gbs = alg * B / us / 1e3
        print(f"| {n} | {alg} | {us:.1f} ({lo:.1f} .. {hi:.1f}) | {B / us * 1e6:.3e} | {gbs:.0f} | {gbs / 80:.1f} % |", flush=True)
