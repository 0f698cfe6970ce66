"""Closed-loop rates (bench.py's closed_loop_measurement) on their own: flight_easy 3 agents at 4096 and 65536 envs, flight 3 agents
at 8192, and the policy kernel alone.    [COOPSEARCH_LIB=...] python tools/exp_closed_loop.py [easy|flight|all]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import cooperative_search_amd as cs
import bench
what = sys.argv[1] if len(sys.argv) > 1 else "all"
dev = torch.device("cuda", 0)
runs = []
if what in ("easy", "all"):
    runs += [(3, 4096, 2000, 200, "flight_easy"), (3, 65536, 400, 100, "flight_easy")]
if what in ("flight", "all"):
    runs += [(3, 8192, 400, 100, "flight")]
for n, B, K, W, env in runs:
    r = bench.closed_loop_measurement(cs, dev, n, B, K, W, env)
    print(f"{env} n={n} B={B}: {r['value']:.3e} env-steps/s, {r['ms_per_step'] * 1e3:.2f} us per step, policy kernel(s) {r['policy_kernels_us']:.1f} us"
          + (f", policy roofline {r['policy_roofline']['frac']:.3f} of {r['policy_roofline']['peak']} TFLOP/s" if 'policy_roofline' in r else ""), flush=True)
