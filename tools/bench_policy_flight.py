"""flight: conv front end (k_conv_features) + k_policy<8> vs the torch modules of the same network, and the closed loop."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cooperative_search_amd as cs
from tools.bench_policy import timed


def main():
    for n, B in ((3, 1024), (3, 8192)):
        args = cs.make_env_args("flight", n_agents=n)
        env = cs.BatchedFlightEnv(args, batch=B, freeze_done=True)
        cs.apply_env_info(args, env)
        torch.manual_seed(0)
        fused = cs.FusedAgents(args, B)
        ref = cs.BatchedAgents(args, B, net=fused.net)
        obs = env.get_obs()
        last = torch.zeros(B, n, 3, device="cuda")
        t_all = timed(lambda: fused.choose_action(obs), 100)
        t_conv = timed(lambda: fused._conv_features(obs, n * 2504, B, fused.feat), 100)
        t_t = timed(lambda: ref.choose_action(obs, last, evaluate=True), 20)

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
        print(json.dumps(dict(workload=f"flight {n}a B={B}", conv_us=round(t_conv, 2), conv_plus_policy_us=round(t_all, 2),
                              torch_us=round(t_t, 2), map_read_GBps=round(B * 10000 / t_conv / 1e3, 1),
                              closed_loop=run(step))), flush=True)
        del env, fused, ref
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
