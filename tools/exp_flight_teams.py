"""flight, B = 8192: cs_rollout (k_step, then k_flight_pipe: the map sweep of step t beside step t + 1) against one cs_step per step
(k_step + k_map) for every team size -- k_flight_pipe<4..8> spill VGPRs (profiles/kernel_resources.txt): where is the crossover?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cooperative_search_amd as cs
B, T = int(os.environ.get("B", 8192)), 40
for n in (1, 2, 3, 4, 5, 6, 8):
    res = {}
    for mode in ("rollout", "step"):
        env = cs.BatchedFlightEnv(cs.make_env_args("flight", n_agents=n), batch=B, freeze_done=False, auto_reset=True)
        acts = torch.randint(0, 3, (T, B, n), dtype=torch.int32, device="cuda")
        out = None

        def run():
            global out
            if mode == "rollout":
                return env.rollout(acts, out=None, update_views=False)
            for t in range(T):
                env.step(acts[t])
        run(); run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): run()
        e1.record(); torch.cuda.synchronize()
        res[mode] = e0.elapsed_time(e1) * 1e3 / (3 * T)
        del env, acts
        torch.cuda.empty_cache()
    alg = n * (2500 + 4) * 4 + 2 * 2500 * 4 + 36 * n + 186   # obs copies of the map + map read / write + the rest
    print(f"flight n={n} B={B}: cs_rollout {res['rollout']:.1f} us per step ({alg * B / res['rollout'] * 1e6 / 8e12 * 100:.0f} % of 8 TB/s), "
          f"cs_step loop {res['step']:.1f} us per step (eager launches)", flush=True)
