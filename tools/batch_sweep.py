"""Batch sweep of the flight_easy rollout kernels (SURVEY.md section 8d asks for 2^12..2^22): env-steps/s and the
algorithmic-bytes roofline fraction per batch size: the octet kernels (ode:
kinematics, detection and emitting wavefront per 8 envs; od: kinematics and detection wavefront; oct: one wavefront per 8 envs)
and the lane-per-env kernels (lane: round 2's, lanev: round 4's).  Writes a markdown table."""
import json, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = []
for n in (3, 5):
    for B in (1024, 2048, 4096, 8192, 16384, 32768, 65536, 262144, 1048576, 4194304):
        for kernel in ("ode", "od", "oct", "lane", "lanev"):   # (the 16-lane rollout kernels of rounds 1-2 are gone: profiles/r04_batch_sweep.md has them)
            if kernel in ("lane", "lanev") and B < 16384:
                continue
            if kernel in ("od", "oct") and B > (1 << 20):
                continue
            if kernel == "ode" and B > (1 << 15):
                continue
            steps = 40 if B >= (1 << 22) else (400 if B >= (1 << 18) else 1000)   # 2^22: the output tables of 100 steps would not fit
            out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-also",
                                  "--workload", "c2" if n == 3 else "c3", "--mode", "rollout", "--kernel", kernel,
                                  "--batch", str(B), "--steps", str(steps), "--warmup", str(min(100, steps))],
                                 capture_output=True, text=True).stdout.strip().splitlines()
            d = json.loads([ln for ln in out if ln.startswith("{")][-1])
            rows.append((n, B, kernel, d["value"], d["ms_per_step"] * 1e3, d["roofline"]["frac"]))
            print(rows[-1], flush=True)
print("\n| agents | batch | kernel | env-steps/s | us per step | algorithmic GB/s / 8000 |")
print("|---|---|---|---|---|---|")
for n, B, k, v, us, f in rows:
    print(f"| {n} | {B} | {k} | {v:.3e} | {us:.2f} | {100*f:.1f} % |")
