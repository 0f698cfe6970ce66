"""Scratch experiment: where does the time go? (pure step vs reset vs rollout without resets)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import cooperative_search_amd as cs

def timeit(fn, iters):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters  # us

for n, B in ((3, 4096), (5, 16384)):
    args = cs.make_env_args("flight_easy", n_agents=n)
    args.time_limit = 10 ** 9          # no time-outs -> (almost) no resets
    env = cs.BatchedFlightEnv(args, batch=B, freeze_done=False, auto_reset=True)
    T = 100
    acts = torch.randint(0, 3, (T, B, n), dtype=torch.int32, device="cuda")
    out = env.rollout(acts)
    t_roll = timeit(lambda: env.rollout(acts, out=out, update_views=False), 10)
    t_step = timeit(lambda: env.step(acts[0]), 200)
    t_reset = timeit(lambda: env.reset(init=False), 20)
    m = torch.zeros(B, dtype=torch.uint8, device="cuda"); m[::64] = 1
    t_reset_sparse = timeit(lambda: env.reset(init=False, mask=m), 20)
    t_emit = timeit(lambda: env.refresh(), 50)
    print(f"n={n} B={B}: rollout(no resets) {t_roll/T:.2f} us/step -> {B*T/t_roll:.1f} M env-steps/s | step launch {t_step:.2f} us | "
          f"reset all {t_reset:.2f} us | reset 1/64 {t_reset_sparse:.2f} us | emit {t_emit:.2f} us")
