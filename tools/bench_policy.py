"""Fused policy kernel (csrc/policy.hip): kernel time alone vs the torch module, and the closed loop
policy -> env.step per step, eager and as one hipGraph."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cooperative_search_amd as cs


def timed(fn, iters):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters  # us


def main():
    sizes = ((3, 4096), (5, 16384), (3, 65536))
    if len(sys.argv) == 3:   # python tools/bench_policy.py N_AGENTS BATCH
        sizes = ((int(sys.argv[1]), int(sys.argv[2])),)
    for n, B in sizes:
        args = cs.make_env_args("flight_easy", n_agents=n)
        env = cs.BatchedFlightEnv(args, batch=B, freeze_done=True)
        cs.apply_env_info(args, env)
        torch.manual_seed(0)
        fused = cs.FusedAgents(args, B)
        ref = cs.BatchedAgents(args, B, net=fused.net)
        obs = env.get_obs()
        last = torch.zeros(B, n, 3, device="cuda")
        t_f = timed(lambda: fused.choose_action(obs), 200)
        t_t = timed(lambda: ref.choose_action(obs, last, evaluate=True), 50)
        flops = 2.0 * B * n * (16 * 64 + 2 * 192 * 64 + 64 * 64 + 64 * 16)

        def step():
            env.step(fused.choose_action(env.get_obs()))

        def run(fn, steps=200):
            env.reset(); fused.init_hidden(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                fn()
            torch.cuda.synchronize()
            return B * steps / (time.perf_counter() - t0)

        run(step, 20)
        eager = run(step)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(10):
                step()
        graphed = run(g.replay, 20) * 10
        # the whole closed loop in one launch per 100 steps (k_rollout_policy)
        one = None
        if n <= 5:
            out = env.rollout_policy(fused, 100)
            def fused_loop():
                env.rollout_policy(fused, 100, out=out, update_views=False)
            fused_loop(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                fused_loop()
            torch.cuda.synchronize()
            one = B * 1000 / (time.perf_counter() - t0)
        print(json.dumps(dict(workload=f"flight_easy {n}a B={B}", rows=B * n, fused_us=round(t_f, 2), torch_us=round(t_t, 2),
                              fused_tflops=round(flops / t_f / 1e6, 2), closed_loop_eager=eager, closed_loop_hipgraph=graphed,
                              closed_loop_one_launch=one)),
              flush=True)
        del env, fused, ref
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
