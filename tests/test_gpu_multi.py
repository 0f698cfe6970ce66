"""Multi-GPU path on ONE MI355X (BASELINE config 5).

* c5 emulation: the 65536-env flight_easy 5a15t batch as eight rank shards (`env_offset = r * 8192`, the sharding
  `bench.py --gpus 8` uses) against the single 65536-env batch: bit-equal raw state and per-step outputs, equal metric
  partials -- i.e. what the 8-GPU run computes is what one device computes, whatever the split.
* `bench.py --gpus 2` through its own launcher with BENCH_SHARE_GPU=1 (both ranks on cuda:0, gloo): the N > 1 control
  flow -- spawn, rendezvous, barrier-bracketed timed region, max over ranks, metric all-gather -- on a one-GPU box.
"""
import json
import os
import subprocess
import sys

import pytest
import torch

import cooperative_search_amd as cs
from cooperative_search_amd import dist as csd

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c5_eight_shards_equal_one_batch():
    n, G, world, T = 5, 65536, 8, 120
    args = cs.make_env_args("flight_easy", n_agents=n)
    args.time_limit = 50   # several auto-resets inside the horizon: the reset path is sharded too
    whole = cs.BatchedFlightEnv(args, batch=G, freeze_done=False, auto_reset=True)
    gen = torch.Generator("cuda").manual_seed(5)
    acts = torch.randint(0, 3, (T, G, n), dtype=torch.int32, device="cuda", generator=gen)
    ow = whole.rollout(acts)
    pw = whole.metric_partials().clone()
    raw_w = {k: v.clone() for k, v in whole.raw().items()}
    total = torch.zeros(4, dtype=torch.float64, device="cuda")
    for r in range(world):
        off, cnt = csd.shard(G, r, world)
        assert (off, cnt) == (r * 8192, 8192)
        part = cs.BatchedFlightEnv(args, batch=cnt, env_offset=off, freeze_done=False, auto_reset=True)
        op = part.rollout(acts[:, off:off + cnt].contiguous())
        for key in ("reward", "terminated", "win", "obs", "state"):
            assert torch.equal(op[key], ow[key][:, off:off + cnt]), f"rank {r}: {key}"
        for key in ("tgt", "agent", "hdr"):
            assert torch.equal(part.raw()[key], raw_w[key][off:off + cnt]), f"rank {r}: raw {key}"
        assert torch.equal(part.mt_canonical(), whole.mt_canonical()[off:off + cnt]), f"rank {r}: MT stream"
        total += part.metric_partials()
        del part, op
    assert torch.equal(total, pw) and int(pw[3].item()) == G
    assert int(whole.raw()["hdr"][:, cs.lib.H_EPISODES].min().item()) >= 2


def _bench(args, env_extra, timeout=900):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    env.update(env_extra)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                       timeout=timeout)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p, (json.loads(lines[-1]) if lines else None)


def test_bench_two_ranks_share_one_gpu_through_the_launcher():
    p, line = _bench(["--gpus", "2", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--min-gpu-s", "0.05"],
                     {"BENCH_SHARE_GPU": "1"})
    assert p.returncode == 0, p.stderr[-3000:]
    assert line["n_gpus"] == 2 and line["steps"] == 20 and line["warmup"] == 5
    assert line["eval"]["envs"] == 2 * 4096 and line["eval"]["world_size"] == 2      # the all-gather saw both ranks
    assert line["value"] > 1e6 and abs(line["ms_per_step"] * line["value"] / 1e3 - 2 * 4096) < 1e-6 * 8192
    assert line["roofline"]["frac"] == pytest.approx(
        line["roofline"]["algorithmic_bytes_per_env_step"] * 4096 / (line["ms_per_step"] / 1e3) / 1e9 / 8000.0, rel=1e-9)
    labels = [e["workload"] for e in line["also"]]
    assert any(lb.startswith("c5 weak") for lb in labels) and any(lb.startswith("c5 strong") for lb in labels)
    strong = [e for e in line["also"] if e["workload"].startswith("c5 strong")][0]
    assert strong["value"] > 1e6 and strong["roofline"]["algorithmic_bytes_per_env_step"] == 366


def test_bench_single_gpu_line_is_single_clocked():
    """value, ms_per_step and roofline.achieved come from the same HIP-event interval."""
    p, line = _bench(["--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-also", "--min-gpu-s", "0.05"], {})
    assert p.returncode == 0, p.stderr[-3000:]
    assert line["n_gpus"] == 1 and line["timing"]["repeats"] >= 5
    per_step_s = line["ms_per_step"] / 1e3
    assert line["value"] == pytest.approx(4096 / per_step_s, rel=1e-9)
    assert line["roofline"]["achieved"] == pytest.approx(294 * line["value"] / 1e9, rel=1e-9)
    assert line["roofline"]["avg_launch_us"] == pytest.approx(per_step_s * 20 * 1e6, rel=1e-9)
