"""Quick check of k_rollout_lanev against k_rollout_lane (same seeds, same actions: everything bit for bit), then its speed.
    N=5 python tools/exp_lanev_check.py [B ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cooperative_search_amd as cs
n = int(os.environ.get("N", 3))
T = 260
for am in (0, 3):
    B = 640 + 37
    args = cs.make_env_args("flight_easy", n_agents=n); args.agent_mode = am
    seeds = np.arange(B, dtype=np.uint32) + 11
    acts = torch.from_numpy(np.random.RandomState(5).randint(0, 3, size=(T, B, n)).astype(np.int32)).cuda()
    res = {}
    for k in ("lane", "lanev"):
        env = cs.BatchedFlightEnv(args, batch=B, seeds=seeds, freeze_done=False, auto_reset=True, kernel=k)
        env.seed(seeds); env.reset(init=True)
        o1 = env.rollout(acts[:100]); o1 = {a: b.clone() for a, b in o1.items()}
        o2 = env.rollout(acts[100:]); o2 = {a: b.clone() for a, b in o2.items()}
        raw = {a: b.clone() for a, b in env.raw().items() if a in ("agent", "tgt", "hdr")}
        res[k] = (o1, o2, raw, env.mt_canonical().clone())
    ok = True
    for part in (0, 1):
        for key in res["lane"][part]:
            same = torch.equal(res["lane"][part][key], res["lanev"][part][key])
            ok &= same
            if not same:
                d = (res["lane"][part][key].float() - res["lanev"][part][key].float()).abs()
                print("  MISMATCH", am, part, key, float(d.max()), int((d > 0).sum()))
    for key in res["lane"][2]:
        same = torch.equal(res["lane"][2][key], res["lanev"][2][key]); ok &= same
        if not same: print("  MISMATCH raw", am, key)
    same = torch.equal(res["lane"][3], res["lanev"][3]); ok &= same
    if not same: print("  MISMATCH mt rows", am)
    print(f"n={n} agent_mode={am}: lanev == lane: {ok}", flush=True)
for B in [int(x) for x in sys.argv[1:]] or [262144]:
    for k in ("lane", "lanev"):
        env = cs.BatchedFlightEnv(cs.make_env_args("flight_easy", n_agents=n), batch=B, freeze_done=False, auto_reset=True, kernel=k)
        acts = torch.randint(0, 3, (100, B, n), dtype=torch.int32, device="cuda")
        out = env.rollout(acts)
        for _ in range(2): env.rollout(acts, out=out, update_views=False)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): env.rollout(acts, out=out, update_views=False)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 500
        print(f"n={n} {k} B={B}: {us:.2f} us/step, {B / us * 1e6:.3e} env-steps/s, {(36*n+186) * B / us * 1e6 / 8e12 * 100:.1f} %", flush=True)
        del env, acts, out; torch.cuda.empty_cache()
