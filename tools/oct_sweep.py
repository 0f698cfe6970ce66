#!/usr/bin/env python3
"""Batch sweep of the flight_easy rollout kernels (pair / one-wavefront / octet / lane): us per step and env-steps/s of
100-step cs_rollout calls with auto-reset and obs + state emission (bench.py's protocol, shortened).
    python tools/oct_sweep.py [--n 3,5] [--batches 4096,8192,...] [--kernels duo,solo,oct,lane] [--T 100] [--tag x]
Prints one JSON line per point; COOPSEARCH_LIB selects an experimental build of the library."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import cooperative_search_amd as cs

ap = argparse.ArgumentParser()
ap.add_argument("--n", default="3,5")
ap.add_argument("--batches", default="4096,8192,16384,32768,65536,262144")
ap.add_argument("--kernels", default="duo,solo,oct,lane")
ap.add_argument("--T", type=int, default=100)
ap.add_argument("--reps", type=int, default=6)
ap.add_argument("--tag", default=os.environ.get("COOPSEARCH_LIB", "in-tree"))
ap.add_argument("--detect-prob", type=float, default=None, help="override detect_prob (0: no target is ever found -> no wins)")
ap.add_argument("--no-reset", action="store_true", help="done envs keep stepping (no auto-reset): what the resets cost at run time")
a = ap.parse_args()
dev = torch.device("cuda", 0)
for n in [int(v) for v in a.n.split(",")]:
    for B in [int(v) for v in a.batches.split(",")]:
        for kernel in a.kernels.split(","):
            if kernel in ("duo", "solo", "od") and B > 65536:
                continue
            eargs = cs.make_env_args("flight_easy", n_agents=n)
            if a.detect_prob is not None:
                eargs.detect_prob = a.detect_prob
            env = cs.BatchedFlightEnv(eargs, batch=B, device=dev, freeze_done=False,
                                      auto_reset=not a.no_reset, kernel=kernel)
            acts = torch.randint(0, 3, (a.T, B, n), dtype=torch.int32, device=dev)
            out = env.rollout(acts, update_views=False)
            for _ in range(2):
                env.rollout(acts, out=out, update_views=False)
            torch.cuda.synchronize()
            ts = []
            for _ in range(a.reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                env.rollout(acts, out=out, update_views=False)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3 / a.T)
            ts.sort()
            us = ts[len(ts) // 2]
            alg = 36 * n + 12 * 15 + 6
            print(json.dumps({"tag": a.tag, "n": n, "B": B, "kernel": kernel, "us_per_step": round(us, 3),
                              "env_steps_per_s": round(B / us * 1e6, -5), "hbm_frac": round(alg * B / us * 1e6 / 8e12, 4)}), flush=True)
            del env, out, acts
            torch.cuda.empty_cache()
