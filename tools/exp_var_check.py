"""An experimental one-team-size build (COOPSEARCH_LIB=build/var/x.so) checked and timed:
    python tools/exp_var_check.py <n_agents> <kernel,kernel,...> <B,B,...> [T=100] [--nocheck]
Check: each rollout kernel against the 16-lane step kernel of the SAME library, bit for bit (rewards, flags, obs / state at
intervals, raw state, canonical MT rows), on the shipped configuration with auto-reset, on a crowded one (small map: the
repulsion active in most steps) and on one where nearly every target is in view (draw slots past 64 for teams of 5).
Timing: median of 7 regions of `reps` launches, env-steps/s."""
import os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import cooperative_search_amd as cs

n = int(sys.argv[1])
kernels = sys.argv[2].split(",")
Bs = [int(x) for x in sys.argv[3].split(",")]
T = int(sys.argv[4]) if len(sys.argv) > 4 and not sys.argv[4].startswith("--") else 100
check = "--nocheck" not in sys.argv


def custom(**kw):
    a = cs.make_env_args("flight_easy", n_agents=n)
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def run(kernel, args, B, lengths, mode):
    seeds = np.arange(B, dtype=np.uint32) + 5150
    g = torch.Generator("cuda").manual_seed(17)
    e1 = cs.BatchedFlightEnv(args, batch=B, seeds=seeds, kernel="group", **mode)
    e2 = cs.BatchedFlightEnv(args, batch=B, seeds=seeds, kernel=kernel, **mode)
    for L in lengths:
        acts = torch.randint(0, 3, (L, B, n), dtype=torch.int32, device="cuda", generator=g)
        out = e2.rollout(acts)
        for k in range(L):
            r, term, win = e1.step(acts[k])
            assert torch.equal(r, out["reward"][k]) and torch.equal(term, out["terminated"][k]) and torch.equal(win, out["win"][k]), (kernel, k)
            if k % 5 == 0 or k == L - 1:
                assert torch.equal(e1.get_obs(), out["obs"][k]) and torch.equal(e1.get_state(), out["state"][k]), (kernel, k)
        r1, r2 = e1.raw(), e2.raw()
        for key in ("tgt", "agent", "hdr"):
            assert torch.equal(r1[key], r2[key]), (kernel, key)
        assert torch.equal(e1.mt_canonical(), e2.mt_canonical()), kernel


name = os.path.basename(cs.lib.library_path())
if check:
    for kernel in kernels:
        run(kernel, custom(), 2048, (100, 20, 57), dict(freeze_done=False, auto_reset=True))
        run(kernel, custom(map_size=8, view_range=2), 1024, (60, 33), dict(freeze_done=False, auto_reset=True))   # crowded: forces in most steps
        run(kernel, custom(view_range=70, detect_prob=0.3), 520, (50, 5, 33), dict(freeze_done=False, auto_reset=True))
        run(kernel, custom(), 520, (40, 9), dict(freeze_done=True))
        print(f"{name}, n={n}, {kernel}: bit-identical to the step kernel", flush=True)
for kernel in kernels:
    for B in Bs:
        env = cs.BatchedFlightEnv(cs.make_env_args("flight_easy", n_agents=n), batch=B, freeze_done=False, auto_reset=True, kernel=kernel)
        acts = torch.randint(0, 3, (T, B, n), dtype=torch.int32, device="cuda")
        out = env.rollout(acts)
        for _ in range(3):
            env.rollout(acts, out=out, update_views=False)
        torch.cuda.synchronize()
        reps = max(2, int(2000 / T))
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                env.rollout(acts, out=out, update_views=False)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / reps)
        ms = statistics.median(ts)
        v = B * T / (ms / 1e3)
        print(f"{name} n={n} {kernel} B={B} T={T}: {ms * 1e3 / T:.3f} us/step, {v:.4e} env-steps/s  (min {min(ts) * 1e3 / T:.3f} max {max(ts) * 1e3 / T:.3f})", flush=True)
        del env, acts, out
        torch.cuda.empty_cache()
