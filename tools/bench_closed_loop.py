"""Closed-loop throughput of the 'next' rows: a recurrent Q-network (agents.BatchedAgents, the
reference's RNN architecture) picks the actions of all B*n (env, agent) pairs from the live obs each step, the env
steps, and (optionally) the collector materialises the episode batch and the HBM replay buffer stores it.
Reports env-steps/s for: policy + env.step eager, the same captured in one hipGraph per step, and the full
EpisodeCollector.generate_episodes + DeviceReplayBuffer.store_episode path."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cooperative_search_amd as cs

def main():
    res = []
    only = int(sys.argv[1]) if len(sys.argv) > 1 else None   # index of a single workload (for profiling)
    for idx, (env_name, n, B) in enumerate((("flight_easy", 3, 4096), ("flight_easy", 5, 16384), ("flight_easy", 3, 65536), ("flight", 3, 1024))):
        if only is not None and idx != only:
            continue
        args = cs.make_env_args(env_name, n_agents=n)
        env = cs.BatchedFlightEnv(args, batch=B, freeze_done=True)
        cs.apply_env_info(args, env)
        torch.manual_seed(0)
        fused = os.environ.get("TORCH_POLICY") != "1"
        agents = cs.FusedAgents(args, B) if fused else cs.BatchedAgents(args, B)   # csrc/policy.hip vs torch modules
        T = args.episode_limit
        last = torch.zeros(B, n, 3, device="cuda")
        actions = torch.zeros(B, n, dtype=torch.int64, device="cuda")

        def one_step():
            if fused:   # one launch: input assembly, fc1, GRU, fc2, argmax; last action kept in the action buffer
                env.step(agents.choose_action(env.get_obs()))
                return
            a = agents.choose_action(env.get_obs(), last, evaluate=True)
            actions.copy_(a)
            env.step(actions)
            last.copy_(torch.nn.functional.one_hot(actions, 3).to(torch.float32))

        def run(fn, steps):
            env.reset()
            agents.init_hidden()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                fn()
            torch.cuda.synchronize()
            return B * steps / (time.perf_counter() - t0)

        run(one_step, 20)
        eager = run(one_step, T)
        # one hipGraph per step: policy forward + epsilon-greedy + env kernels
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                one_step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            one_step()
        graphed = run(g.replay, T)
        # full collection + storage
        col = cs.EpisodeCollector(env)
        rb = cs.DeviceReplayBuffer(args, 2 * B)
        pol = agents.policy(0.0, True)
        kw = dict(agents=agents) if fused else dict(policy=pol)
        col.generate_episodes(into=rb, **kw)          # warm-up (allocator, kernel loading)
        ep, rew, win, found = col.generate_episodes(**kw)
        torch.cuda.synchronize()
        reps = 3
        t0 = time.perf_counter()
        for _ in range(reps):
            col.generate_episodes(into=rb, **kw)      # episodes go straight into the replay ring
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        executed = float((ep["padded"][:, :, 0] == 0).sum().item())
        res.append(dict(workload=f"{env_name} {n}a15t B={B}", policy="fused HIP" if fused else "torch", policy_plus_step_eager=eager, policy_plus_step_hipgraph=graphed,
                        collect_and_store_episodes=B * T / dt, executed_env_steps_per_s=executed / dt,
                        episodes_per_s=B / dt))
        print(json.dumps(res[-1]), flush=True)
        del env, agents, rb, col, ep
        torch.cuda.empty_cache()

if __name__ == "__main__":
    main()
