#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the batched flight_easy / flight environment path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3|c4|c5] [--mode step|rollout]

One "step" = one env.step() of EVERY environment of the batch, including the obs + state emission that
common/rollout.py:45-63 needs each step.  Inputs (pre-drawn uniform actions) are resident in HBM before the
timed region.  Done environments are auto-reset (CS_AUTO_RESET), so every environment does a full step every
time: value = batch x K / time.

Workloads (BASELINE.json configs):
    c2 (default, N = 1 headline)  flight_easy, 3 agents, 15 targets, 4096 envs per GPU
    c3                            flight_easy, 5 agents, 15 targets, 16384 envs per GPU
    c4                            flight (probability map), 3 agents, 15 targets, 8192 envs per GPU
    c5                            flight_easy, 5 agents, 15 targets, 8192 envs per GPU (65536 over 8 GPUs)
Modes:
    step     one cs_step launch per step (the closed-loop path a policy drives), replayed from a hipGraph
    rollout  cs_rollout: T = 100 steps per launch with the env resident in registers (open-loop action table;
             flight_easy only) -- default for flight_easy
Multi-GPU: one process per GPU (torch.distributed, backend nccl = RCCL), the batch is sharded by global env index
with no data-path collective ("scaling": "weak": per-GPU batch fixed); the only collective is the all-gather of
the evaluation-metric partials after the timed region.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

WORKLOADS = {
    "c2": dict(env="flight_easy", n_agents=3, batch=4096),
    "c3": dict(env="flight_easy", n_agents=5, batch=16384),
    "c4": dict(env="flight", n_agents=3, batch=8192),
    "c5": dict(env="flight_easy", n_agents=5, batch=8192),
}
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s HBM3E


def algorithmic_bytes_per_env_step(env, n, m, mode):
    """SURVEY.md section 8(d).  flight_easy, one launch per step: 61n + 22m + 22 (535 B at 3a15t, 657 B at 5a15t);
    fused T-step rollout (state resident in registers): 36n + 12m + 6 (294 B / 366 B);
    flight: obs n*(2500+4)*4 + state 4*(4n+3m) + ONE read of the 10 000-byte map + the flight_easy remainder
    (40 535 B at 3a15t).  This is the model that matches k_map: it writes the map back only where a cell changed
    (data dependent, not counted); SURVEY's full-map streaming model (+10 000 B write-back) would be 50 535 B."""
    if env == "flight":
        return n * 2504 * 4 + 4 * (4 * n + 3 * m) + 10000 + (61 * n + 22 * m + 22 - 16 * n - 4 * (4 * n + 3 * m))
    if mode == "rollout":
        return 36 * n + 12 * m + 6
    return 61 * n + 22 * m + 22


def largest_divisor_leq(k, cap):
    for d in range(min(cap, k), 0, -1):
        if k % d == 0:
            return d
    return 1


def cpu_baseline(env_name, n, batch, budget_s=12.0):
    """The C oracle (a port of the reference's algorithm, parity-pinned by tests/) timed on this host's cores, on a
    bounded sample of the same workload: same batch, auto-reset, obs+state emission, x*x squares (its fast mode)."""
    from oracle import oracle as orc
    threads = max(1, min(orc.OracleBatch.max_threads(), os.cpu_count() or 1))
    B = batch if env_name != "flight" else min(batch, 256)
    cfg = orc.make_config(variant=env_name, n_agents=n)
    seeds = (20240000 + np.arange(B)).astype(np.uint32)
    orc.set_exact_pow(False)
    try:
        ob = orc.OracleBatch(cfg, B, seeds)
        ob.reset(init=True, threads=threads)
        rng = np.random.RandomState(1)
        acts = rng.randint(0, 3, size=(16, B, n)).astype(np.int32)
        # calibrate, then run ~budget_s
        t0 = time.perf_counter()
        for s in range(4):
            ob.step(acts[s % 16], auto_reset=True, freeze_done=False, threads=threads)
        per = (time.perf_counter() - t0) / 4
        steps = int(max(8, min(20000, budget_s / max(per, 1e-6))))
        t0 = time.perf_counter()
        for s in range(steps):
            ob.step(acts[s % 16], auto_reset=True, freeze_done=False, threads=threads)
        dt = time.perf_counter() - t0
        multi = B * steps / dt
        # single thread, shorter
        steps1 = max(4, steps // (4 * threads))
        t0 = time.perf_counter()
        for s in range(steps1):
            ob.step(acts[s % 16], auto_reset=True, freeze_done=False, threads=1)
        single = B * steps1 / (time.perf_counter() - t0)
    finally:
        orc.set_exact_pow(True)
    return {"value": multi, "unit": "env-steps/s", "cores": threads, "kind": "port",
            "sample": f"C oracle (oracle/flight_oracle.c, OpenMP over envs), {B} envs x {steps} steps, auto-reset, "
                      f"obs+state emitted, {dt:.1f} s wall", "single_thread_value": single}


def measure(cs, dev, env_name, n, B, mode, K, W, kernel, rank=0, no_graph=False, barrier=lambda: None):
    """Times K steps of one workload on `dev`; returns (seconds wall, event ms, env, S)."""
    m = 15
    S = largest_divisor_leq(K, 100)  # steps per graph replay / per rollout launch
    env = cs.BatchedFlightEnv(cs.make_env_args(env_name, n_agents=n), batch=B, device=dev, env_offset=rank * B,
                              freeze_done=False, auto_reset=True, kernel=kernel)
    g = torch.Generator(device=dev).manual_seed(1 + rank)
    acts = torch.randint(0, 3, (S, B, n), dtype=torch.int32, device=dev, generator=g)
    if mode == "rollout":
        out = dict(
            reward=torch.empty(S, B, dtype=torch.float32, device=dev),
            terminated=torch.empty(S, B, dtype=torch.uint8, device=dev),
            win=torch.empty(S, B, dtype=torch.uint8, device=dev),
            obs=torch.empty(S, B, n, 4, dtype=torch.float32, device=dev),
            state=torch.empty(S, B, env.state_shape, dtype=torch.float32, device=dev))

        def chunk():
            env.rollout(acts, out=out, update_views=False)
    else:
        def chunk_eager():
            for s in range(S):
                env.step(acts[s])
        if no_graph:
            chunk = chunk_eager
        else:
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                chunk_eager()
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize(dev)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                chunk_eager()

            def chunk():
                graph.replay()
    for _ in range(max(1, math.ceil(W / S))):
        chunk()
    torch.cuda.synchronize(dev)
    barrier()
    torch.cuda.synchronize(dev)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(K // S):
        chunk()
    ev1.record()
    torch.cuda.synchronize(dev)
    barrier()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    return dt, ev0.elapsed_time(ev1), env, S


def side_measurement(cs, dev, label, env_name, n, B, mode, K, W, kernel):
    """Compact entry for the `also` object: other workloads measured in the same run (N = 1 only)."""
    dt, ev_ms, env, S = measure(cs, dev, env_name, n, B, mode, K, W, kernel)
    launches = K // S if mode == "rollout" else K
    alg = algorithmic_bytes_per_env_step(env_name, n, 15, mode)
    achieved = alg * B * (S if mode == "rollout" else 1) / ((ev_ms / 1e3) / launches) / 1e9
    del env
    torch.cuda.empty_cache()
    return {"workload": label, "mode": mode, "kernel": kernel, "value": B * K / dt, "unit": "env-steps/s",
            "ms_per_step": dt * 1e3 / K, "algorithmic_bytes_per_env_step": alg,
            "roofline_achieved_GBps": achieved, "roofline_frac": achieved / HBM_PEAK_GBPS}


def closed_loop_measurement(cs, dev, n, B, K, W, env_name="flight_easy"):
    """`also` entry: the reference's recurrent agent network (agents.FusedAgents -> csrc/policy.hip) picks every action
    from the live obs, then env.step: two launches per env step (flight: conv features, policy, step, map), nothing
    leaves the device."""
    args = cs.make_env_args(env_name, n_agents=n)
    env = cs.BatchedFlightEnv(args, batch=B, device=dev, freeze_done=False, auto_reset=True)
    cs.apply_env_info(args, env)
    torch.manual_seed(0)
    agents = cs.FusedAgents(args, B, device=dev)
    one_launch = env_name == "flight_easy"   # k_rollout_policy: 100 closed-loop steps per launch
    chunk = 100
    out = env.rollout_policy(agents, chunk) if one_launch else None

    def advance(steps):
        if one_launch:
            for _ in range(steps // chunk):
                env.rollout_policy(agents, chunk, out=out, update_views=False)
        else:
            for _ in range(steps):
                env.step(agents.choose_action(env.get_obs()))

    K, W = (K // chunk) * chunk, max(chunk, (W // chunk) * chunk)
    advance(W)
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    advance(K)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    obs = env.get_obs()
    e0.record()
    for _ in range(200):
        agents.choose_action(obs)
    e1.record()
    torch.cuda.synchronize(dev)
    pol_us = e0.elapsed_time(e1) * 1e3 / 200
    del env, agents
    torch.cuda.empty_cache()
    out = {"workload": f"closed loop: {env_name} {n}a15t B={B}, recurrent policy picks every action"
                       + (" (k_rollout_policy: network + env step fused, 100 steps per launch)" if one_launch
                          else " (conv features, policy, step, map kernels per step)"),
           "mode": "closed-loop", "value": B * K / dt, "unit": "env-steps/s", "ms_per_step": dt * 1e3 / K,
           "policy_kernels_us": pol_us}
    if env_name == "flight_easy":   # one kernel, GEMM-shaped: price it against the fp32 matrix peak
        flops = 2.0 * B * n * (16 * 64 + 2 * 192 * 64 + 64 * 64 + 64 * 16)
        out["policy_roofline"] = {"bound": "mfma", "achieved": flops / pol_us / 1e6, "peak": 157.3, "unit": "TFLOP/s",
                                  "frac": flops / pol_us / 1e6 / 157.3, "dtype": "f32"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20000)
    ap.add_argument("--warmup", type=int, default=2000)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--mode", default=None, choices=["step", "rollout"])
    ap.add_argument("--batch", type=int, default=None, help="override envs per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the secondary workloads reported under 'also'")
    ap.add_argument("--kernel", default="auto", choices=["auto", "group", "lane"],
                    help="flight_easy kernel: 16 lanes per env, one lane per env, or by batch size")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the HIP path has no CPU fallback)")
    # BENCH_SHARE_GPU=1 (tests only): all ranks use cuda:0 and the gloo backend, to exercise the N > 1 control flow on
    # a one-GPU box.  The real multi-GPU run is one process per GPU over RCCL (backend "nccl").
    share = os.environ.get("BENCH_SHARE_GPU") == "1"
    dev_index = 0 if share else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        import torch.distributed as dist
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    import cooperative_search_amd as cs

    wl = dict(WORKLOADS[a.workload])
    if a.batch:
        wl["batch"] = a.batch
    env_name, n, B, m = wl["env"], wl["n_agents"], wl["batch"], 15
    mode = a.mode or ("rollout" if env_name == "flight_easy" else "step")
    if env_name == "flight" and mode == "rollout":
        raise SystemExit("rollout mode is flight_easy only")
    K, W = a.steps, a.warmup

    def barrier():
        if world > 1:
            dist.barrier()

    dt, ev_ms, env, S = measure(cs, dev, env_name, n, B, mode, K, W, a.kernel, rank=rank, no_graph=a.no_graph,
                                barrier=barrier)
    cdev = torch.device("cpu") if share else dev  # gloo collectives on host tensors in the shared-GPU test mode
    tmax = torch.tensor([dt], dtype=torch.float64, device=cdev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt_max = float(tmax.item())

    # evaluation-metric reduction (runner.py:86-96): the path's only collective, outside the timed region
    part = env.metric_partials().clone().to(cdev)
    if world > 1:
        gathered = [torch.zeros_like(part) for _ in range(world)]
        dist.all_gather(gathered, part)
        part = torch.stack(gathered).sum(0)
    part = part.cpu().numpy()

    if rank == 0:
        launches = K // S if mode == "rollout" else K
        steps_per_launch = S if mode == "rollout" else 1
        alg = algorithmic_bytes_per_env_step(env_name, n, m, mode)
        launch_s = (ev_ms / 1e3) / launches
        achieved = alg * B * steps_per_launch / launch_s / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            traffic = (json.load(open(tpath)).get(f"{a.workload}:{mode}") or {}).get("per_launch_bytes")
            if a.batch or a.kernel != "auto":
                traffic = None  # the committed PMC passes were taken on the default batch / kernel
        lane = env_name == "flight_easy" and (a.kernel == "lane" or (a.kernel == "auto" and B >= 32768))
        kernel = {"rollout": f"k_rollout<{n}>", "step": f"k_step<{n},{1 if env_name == 'flight' else 0}>"}[mode]
        if lane:
            kernel = f"k_rollout_lane<{n}>"
        if env_name == "flight":
            kernel += f" + k_map<{n}>"
        line = {
            "metric": "env-steps/sec", "value": B * world * K / dt_max, "unit": "env-steps/s", "n_gpus": world,
            "steps": K, "warmup": W, "ms_per_step": dt_max * 1e3 / K, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{env_name}, {n} agents, {m} targets, batch={B} envs per GPU, agent_mode=0, "
                                   f"target_mode=0 ({a.workload})", "mode": mode, "steps_per_launch": steps_per_launch,
                       "auto_reset": True, "emits": "obs+state every step", "kernel": a.kernel, "hip_graph": mode == "step" and not a.no_graph},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "kernel": kernel,
                         "algorithmic_bytes_per_env_step": alg, "avg_launch_us": launch_s * 1e6,
                         "timing": "HIP events on the launch stream over the timed region / launches"},
            "eval": {"mean_episode_reward_so_far": part[0] / part[3], "win_rate_now": part[1] / part[3],
                     "mean_targets_found_now": part[2] / part[3], "envs": int(part[3])},
        }
        if not a.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(env_name, n, B)
        if world == 1 and not a.no_also and a.workload == "c2" and not a.batch:
            del env
            torch.cuda.empty_cache()
            line["also"] = [
                side_measurement(cs, dev, "c2 flight_easy 3a15t B=4096, one launch per step (hipGraph)", "flight_easy", 3,
                                 4096, "step", 10000, 1000, "auto"),
                side_measurement(cs, dev, "c3 flight_easy 5a15t B=16384", "flight_easy", 5, 16384, "rollout", 2000, 200, "auto"),
                side_measurement(cs, dev, "c4 flight 3a15t B=8192 (k_step + k_map per step)", "flight", 3, 8192, "step",
                                 2000, 200, "auto"),
                side_measurement(cs, dev, "flight_easy 3a15t B=262144 (lane-per-env kernel; batch sweep asymptote)",
                                 "flight_easy", 3, 262144, "rollout", 400, 100, "lane"),
                closed_loop_measurement(cs, dev, 3, 4096, 4000, 400),
                closed_loop_measurement(cs, dev, 3, 65536, 1000, 100),
                closed_loop_measurement(cs, dev, 3, 8192, 1000, 100, "flight"),
            ]
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
