#!/usr/bin/env python3
"""CPU-side cost of one env.rollout() call (python -> torch op / ctypes -> cs_rollout -> kernel launch), and the GPU-side gap
an event-timed single launch sees."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cooperative_search_amd as cs
dev = torch.device("cuda", 0)
for binding in ("torch", "ctypes"):
    env = cs.BatchedFlightEnv(cs.make_env_args("flight_easy", n_agents=3), batch=4096, device=dev, freeze_done=False, auto_reset=True, binding=binding)
    acts = torch.randint(0, 3, (20, 4096, 3), dtype=torch.int32, device=dev)
    out = env.rollout(acts, update_views=False)
    for _ in range(5):
        env.rollout(acts, out=out, update_views=False)
    torch.cuda.synchronize()
    # (a) CPU time per call while the GPU queue is backed up (pure host cost)
    t0 = time.perf_counter()
    for _ in range(200):
        env.rollout(acts, out=out, update_views=False)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    # (b) event-timed single launches from an idle GPU
    ts = []
    for _ in range(50):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); env.rollout(acts, out=out, update_views=False); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    # (c) the same with the GPU kept busy first (a queued sleep), so the launch is already in the queue when the GPU gets there
    tb = []
    for _ in range(50):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(200000)
        e0.record(); env.rollout(acts, out=out, update_views=False); e1.record()
        torch.cuda.synchronize()
        tb.append(e0.elapsed_time(e1) * 1e3)
    tb.sort()
    print(f"{binding}: host {1e6 * (t1 - t0) / 200:.1f} us per call issuing 200 back-to-back ({1e6 * (t2 - t0) / 200:.1f} us per call until done); "
          f"event-timed from idle: median {ts[25]:.1f} us, min {ts[0]:.1f}; behind a queued sleep: median {tb[25]:.1f}, min {tb[0]:.1f}")
