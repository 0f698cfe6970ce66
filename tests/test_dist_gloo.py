"""The N > 1 path on CPU: 2-rank gloo process group exercising the sharding helpers and the metric all-gather
(the path's only collective).  The env kernels themselves need a GPU and are covered by the shard-invariance test in
test_gpu_parity.py; here each rank fabricates the per-env counters of its shard deterministically from the global
env index, so the reduced result must not depend on the sharding."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cooperative_search_amd import dist as csd


def fake_counters(offset, count):
    g = np.arange(offset, offset + count, dtype=np.int64)
    total_reward = (g * 7919) % 400 - 300
    target_find = (g * 31) % 16
    win = (target_find == 15).astype(np.int64)
    return total_reward, win, target_find


def _worker(rank, world, port, B, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        off, cnt = csd.shard(B, rank, world)
        tr, win, tf = fake_counters(off, cnt)
        part = torch.tensor([tr.sum(), win.sum(), tf.sum(), cnt], dtype=torch.float64)
        m = csd.reduce_metrics(part)
        curve = csd.FoundCurve(200, 15, "cpu")
        for t in range(200):
            curve.add_step(t, torch.from_numpy(np.minimum(tf, t // 10)))
        curve.end_episodes(cnt)
        res = curve.result()
        seeds = csd.seeds_for(off, cnt)
        if rank == 0:
            torch.save({"metrics": m, "curve": res, "seeds0": seeds[:4].tolist()}, out)
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("B", [4096, 1001])
def test_two_rank_gloo_metric_reduction_is_shard_invariant(tmp_path, B):
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, _free_port(), B, out), nprocs=2, join=True)
    got = torch.load(out, weights_only=False)
    tr, win, tf = fake_counters(0, B)
    assert got["metrics"]["episodes"] == B
    assert got["metrics"]["episode_reward"] == pytest.approx(tr.mean(), abs=1e-12)
    assert got["metrics"]["win_rate"] == pytest.approx(win.mean(), abs=1e-12)
    assert got["metrics"]["targets_find"] == pytest.approx(tf.mean(), abs=1e-12)
    want = np.array([np.minimum(tf, t // 10).mean() / 15 * 100 for t in range(200)])
    np.testing.assert_allclose(got["curve"], want, rtol=0, atol=1e-9)
    assert got["seeds0"] == [20240000, 20240001, 20240002, 20240003]


def test_shard_covers_batch_exactly():
    for B in (1, 7, 64, 4096, 65536, 1001):
        for world in (1, 2, 3, 8):
            spans = [csd.shard(B, r, world) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == B
            for (o0, c0), (o1, _) in zip(spans, spans[1:]):
                assert o0 + c0 == o1
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1


def test_all_gather_sum_is_identity_without_process_group():
    x = torch.arange(5, dtype=torch.float64)
    assert torch.equal(csd.all_gather_sum(x), x)


def _run_bench(args, env_extra=None, timeout=300):
    import json
    import subprocess
    import sys
    env = dict(os.environ)
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)
    env.update(env_extra or {})
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=env, capture_output=True, text=True,
                       timeout=timeout)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p, (json.loads(lines[-1]) if lines else None)


def test_bench_gpus_flag_spawns_that_many_ranks():
    """`bench.py --gpus 2` outside torchrun must start two rank processes itself: the dry-run ranks
    all-gather their shard's env count over gloo, so the line proves the process group had two members."""
    p, line = _run_bench(["--gpus", "2", "--dry-run", "--steps", "20", "--warmup", "5"])
    assert p.returncode == 0, p.stderr[-2000:]
    assert len([ln for ln in p.stdout.splitlines() if ln.startswith("{")]) == 1   # rank 0's line only
    assert line["n_gpus"] == 2 and line["eval"] == {"envs": 2 * 4096, "world_size": 2}
    assert line["steps"] == 20 and line["warmup"] == 5


def test_bench_launcher_propagates_rank_failure():
    """No GPU in this container: the real (non-dry) ranks must fail loudly and the launcher must return non-zero
    without printing a result line."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a GPU-less host")
    p, line = _run_bench(["--gpus", "2", "--steps", "20", "--warmup", "5"])
    assert p.returncode != 0 and line is None
    assert "needs an MI355X" in p.stderr


def test_bench_rejects_world_size_mismatch():
    p, line = _run_bench(["--gpus", "2", "--dry-run"], env_extra={"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and line is None and "WORLD_SIZE=1" in p.stderr
