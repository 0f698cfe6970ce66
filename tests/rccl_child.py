"""Child of tests/test_gpu_rccl.py: ONE rank of a real RCCL process group (backend "nccl"), started fresh by
`python -m torch.distributed.run` so that this process initialises the GPU itself and is never re-exec'ed.

Runs the path's only collective the way bench.py / collector.evaluate run it: the device tensor of
`BatchedFlightEnv.metric_partials()` (runner.py:86-96 sums) through `dist.all_gather_sum` (ONE all-gather), the found-
fraction curve through `FoundCurve.result`, `dist.barrier()`.  Prints one JSON line on rank 0.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", "0"))
    share = os.environ.get("BENCH_SHARE_GPU") == "1"   # several ranks on cuda:0 cannot form an RCCL group: gloo then
    dev = torch.device("cuda", 0 if share else local)
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo" if share else "nccl", **({} if share else {"device_id": dev}))

    import cooperative_search_amd as cs
    from cooperative_search_amd import dist as csd

    B, n, T = 4096, 3, 40
    off, cnt = csd.shard(B * world, rank, world)
    env = cs.BatchedFlightEnv(cs.make_env_args("flight_easy", n_agents=n), batch=cnt, device=dev, env_offset=off,
                              freeze_done=True)
    gen = torch.Generator(device=dev).manual_seed(11 + rank)
    acts = torch.randint(0, 3, (T, cnt, n), dtype=torch.int32, device=dev, generator=gen)
    curve = csd.FoundCurve(T, env.target_num, "cpu" if share else dev)
    for t in range(T):
        env.step(acts[t])
        tf = env.target_find
        curve.add_step(t, tf.cpu() if share else tf)
    curve.end_episodes(cnt)
    local_part = env.metric_partials().clone()
    assert local_part.is_cuda
    part = local_part.cpu() if share else local_part
    total = csd.all_gather_sum(part)                       # RCCL all-gather of the device tensor
    assert total.device == part.device
    metrics = csd.reduce_metrics(part)
    res = curve.result()
    dist.barrier()
    torch.cuda.synchronize(dev)
    if world == 1:
        assert torch.equal(total.cpu(), local_part.cpu()), (total, local_part)
    maps = open(f"/proc/{os.getpid()}/maps").read()
    if rank == 0:
        print(json.dumps({"backend": dist.get_backend(), "world_size": world, "envs": int(total[3].item()),
                          "local": local_part.cpu().tolist(), "reduced": total.cpu().tolist(), "metrics": metrics,
                          "curve_last": float(res[-1]), "librccl_mapped": "librccl" in maps,
                          "coopsearch_mapped": "libcoopsearch_hip" in maps}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
