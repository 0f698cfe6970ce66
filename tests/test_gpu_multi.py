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


def _tail_line(p, keep=8000):
    """The line as a consumer that keeps only the last `keep` characters of stdout reads it."""
    tail = p.stdout[-keep:]
    start = tail.find('{"metric"')
    assert start >= 0, f"no whole JSON line in the last {keep} characters of stdout ({len(p.stdout)} in all)"
    return json.loads(tail[start:].splitlines()[0])


def test_bench_two_ranks_share_one_gpu_through_the_launcher(tmp_path):
    p, line = _bench(["--gpus", "2", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--min-gpu-s", "0.05"],
                     {"BENCH_SHARE_GPU": "1", "BENCH_DETAIL_DIR": str(tmp_path)})
    assert p.returncode == 0, p.stderr[-3000:]
    assert line["n_gpus"] == 2 and line["steps"] == 20 and line["warmup"] == 5
    assert line["eval"]["envs"] == 2 * 4096 and line["eval"]["world_size"] == 2      # the all-gather saw both ranks
    assert line["value"] > 1e6 and abs(line["ms_per_step"] * line["value"] / 1e3 - 2 * 4096) < 1e-6 * 8192
    assert line["roofline"]["frac"] == pytest.approx(
        line["roofline"]["algorithmic_bytes_per_env_step"] * 4096 / (line["ms_per_step"] / 1e3) / 1e9 / 8000.0, rel=1e-9)
    assert len(p.stdout) < 6000 and _tail_line(p) == line
    # N > 1: the headline and the two c5 points only; each as a [value, fraction] pair in the line, in full in the detail file
    assert set(line["also_summary"]) - {"_"} == {"c5w", "c5s"} and line["also_summary"]["c5s"][0] > 1e6
    assert line["c5_strong_total"]["value"] == pytest.approx(line["also_summary"]["c5s"][0], rel=1e-3)
    detail = json.load(open(line["detail"]))
    labels = [e["workload"] for e in detail["also"]]
    assert any(lb.startswith("c5 weak") for lb in labels) and any(lb.startswith("c5 strong") for lb in labels)
    strong = [e for e in detail["also"] if e["key"] == "c5s"][0]
    assert strong["value"] > 1e6 and strong["roofline"]["algorithmic_bytes_per_env_step"] == 366
    assert detail["value"] == line["value"] and detail["roofline"]["frac"] == line["roofline"]["frac"]


def test_bench_default_line_fits_a_tail_consumer(tmp_path):
    """The driver's own command shape (every secondary workload on, no --no-also): stdout is ONE line of less than 6000
    characters that parses from the last 8000 characters of stdout; the 17 secondary workloads are [value, fraction] pairs in
    it and full entries in the detail file it names (r05: a 21.8 KB line, cut by its consumer, parsed = null)."""
    p, line = _bench(["--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--min-gpu-s", "0.05"],
                     {"BENCH_DETAIL_DIR": str(tmp_path)})
    assert p.returncode == 0, p.stderr[-3000:]
    assert p.stdout.count("\n") == 1 and len(p.stdout) < 6000 and _tail_line(p) == line
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in line, key
    assert "dropped_to_fit" not in line
    summ = {k: v for k, v in line["also_summary"].items() if k != "_"}
    assert {"c3", "c4", "c5w", "c5s", "fe3_2^18", "fe5_2^18"} <= set(summ) and len(summ) == 17
    assert all(v[0] > 1e6 for v in summ.values())
    detail = json.load(open(line["detail"]))
    assert [e["key"] for e in detail["also"]] == list(summ)
    c3 = next(e for e in detail["also"] if e["key"] == "c3")
    assert c3["roofline"]["frac"] == pytest.approx(summ["c3"][1], rel=2e-3) and c3["roofline"]["kernel"] == "k_rollout_od<5>"


def test_bench_single_gpu_line_is_single_clocked():
    """value, ms_per_step and roofline.achieved come from the same HIP-event interval."""
    p, line = _bench(["--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-also", "--min-gpu-s", "0.05"], {})
    assert p.returncode == 0, p.stderr[-3000:]
    assert line["n_gpus"] == 1 and line["timing"]["repeats"] >= 5
    per_step_s = line["ms_per_step"] / 1e3
    assert line["value"] == pytest.approx(4096 / per_step_s, rel=1e-9)
    assert line["roofline"]["achieved"] == pytest.approx(294 * line["value"] / 1e9, rel=1e-9)
    assert line["roofline"]["avg_launch_us"] == pytest.approx(per_step_s * 20 * 1e6, rel=1e-9)
