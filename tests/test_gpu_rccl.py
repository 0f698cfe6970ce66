"""RCCL for real (backend "nccl"), at the world size one MI355X allows: 1.

The evaluation-metric reduction (the reference's Runner.evaluate, runner.py:86-96) is the path's only collective.  On
the 8-GPU node it is an RCCL all-gather over xGMI; here the SAME calls -- init_process_group("nccl", device_id=...),
all_gather of the device tensor of metric_partials(), barrier -- run in a one-rank group, through (a) a fresh child
started by torch.distributed.run and (b) bench.py's own control flow (`--pg on`, and under an external torchrun with
WORLD_SIZE=1), so a broken RCCL stack or a wrong call sequence shows up on the one-GPU box and not first on the node.
"""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _clean_env(extra=None):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "BENCH_SHARE_GPU"):
        env.pop(k, None)
    env["MASTER_ADDR"] = "127.0.0.1"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update(extra or {})
    return env


def _torchrun(script_args, nproc=1, timeout=900, extra=None):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port())] + script_args
    p = subprocess.run(cmd, env=_clean_env(extra), capture_output=True, text=True, timeout=timeout)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p, (json.loads(lines[-1]) if lines else None)


def test_rccl_world_size_one_all_gather_of_device_metric_partials():
    p, line = _torchrun([os.path.join(ROOT, "tests", "rccl_child.py")])
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    assert line["backend"] == "nccl" and line["world_size"] == 1 and line["envs"] == 4096
    assert line["reduced"] == line["local"]
    assert line["librccl_mapped"] and line["coopsearch_mapped"]
    m = line["metrics"]
    assert m["episodes"] == 4096 and 0.0 <= m["targets_find"] <= 15.0 and -45.0 <= m["episode_reward"] <= 260.0
    assert 0.0 <= line["curve_last"] <= 100.0


def test_bench_n1_with_the_process_group_up():
    """`bench.py --gpus 1 --pg on`: the N = 1 line produced with RCCL initialised, barrier and all-reduce(MAX) over it
    around the timed region, the metric partials all-gathered on the device."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--pg", "on",
           "--no-cpu-baseline", "--no-also", "--min-gpu-s", "0.05"]
    p = subprocess.run(cmd, env=_clean_env(), capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-4000:]
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["config"]["backend"] == "nccl (RCCL)" and "backend_error" not in line["config"]
    assert line["eval"]["envs"] == 4096 and line["eval"]["world_size"] == 1
    assert line["eval"]["reduced_by"].startswith("all_gather")
    assert line["n_gpus"] == 1 and line["value"] > 1e6


def test_bench_under_external_torchrun_world_size_one():
    """What the driver does at N > 1, at N = 1: torchrun starts bench.py as a rank; RANK / WORLD_SIZE come from the env."""
    p, line = _torchrun([os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5",
                         "--no-cpu-baseline", "--no-also", "--min-gpu-s", "0.05"])
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    assert line["config"]["backend"] == "nccl (RCCL)" and line["eval"]["envs"] == 4096
    assert line["eval"]["reduced_by"].startswith("all_gather")
