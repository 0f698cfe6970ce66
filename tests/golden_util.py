"""Helpers shared by the parity tests: load golden traces, replay them on any env object."""
import glob
import json
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def trace_names(prefix=""):
    """Step-by-step env traces (the episode_*.npz files are RolloutWorker episode dicts, see test_gpu_collector.py)."""
    names = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, prefix + "*.npz")))
    return [n for n in names if n.startswith(("easy_", "flight_"))]


def load_trace(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    meta = json.loads(str(z["meta"]))
    return meta, z


def target_table():
    with open(os.path.join(GOLDEN_DIR, "flight_targets_parsed.json")) as f:
        return json.load(f)
