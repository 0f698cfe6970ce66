#!/usr/bin/env python3
"""profiles/r06_traffic_raw.json (tools/pmc_traffic.sh: FETCH_SIZE / WRITE_SIZE sums per kernel and workload) ->
profiles/traffic.json (HBM bytes per env-step per bench kernel label, what bench.py's roofline.traffic reads).
    python tools/make_traffic_json.py [raw.json] [out.json]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
raw_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r06_traffic_raw.json")
out_path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "traffic.json")
raw = json.load(open(raw_path))

# bench label -> (workload tag of pmc_traffic.sh, kernels of that run that make up the label)
# FETCH_SIZE is doubled for the map-streaming kernels (16 B/lane coalesced stream: gfx950 reports exactly half,
# MI355X_MICROARCH.md section HBM) and left raw elsewhere (uncalibrated width).
# (bench label @ steps per launch, workload tag, kernels, steps per launch of that pass): bench.py only uses an entry for
# a run with the SAME number of steps per launch (prologue traffic per env-step depends on the launch length)
LABELS = [
    ("k_rollout_od<3,E>@100", "c2_rollout", ["k_rollout_od<3, true, true, true>"], 100),
    ("k_rollout_od<3,E>@20", "c2_rollout20", ["k_rollout_od<3, true, true, true>"], 20),
    ("k_step<3,0>@1", "c2_step", ["k_step<3, 0>"], 1),
    ("k_rollout_od<5>@100", "c3_rollout", ["k_rollout_od<5, true, true, false>"], 100),
    ("k_rollout_od<5,E>@100", "c5_rollout", ["k_rollout_od<5, true, true, true>"], 100),
    ("k_rollout_od<3>@100", "od3_16384", ["k_rollout_od<3, true, true, false>"], 100),
    ("k_rollout_oct<3>@100", "oct3_32768", ["k_rollout_oct<3, true, true>"], 100),
    ("k_rollout_oct<5>@100", "oct5_32768", ["k_rollout_oct<5, true, true>"], 100),
    # (one label per kernel: the FIRST workload listed here that was measured; bytes per env-step of the lane-per-env kernel are
    # batch independent above its dispatch threshold)
    ("k_rollout_lanev<5>@100", "lane5_rollout", ["k_rollout_lanev<5, true, 2>", "k_rollout_lanev<5, false, 2>", "k_rollout_lanev<5, true>", "k_rollout_lanev<5, false>"], 100),
    ("k_rollout_lanev<5>@100", "c5s_rollout", ["k_rollout_lanev<5, true, 2>", "k_rollout_lanev<5, false, 2>", "k_rollout_lanev<5, true>", "k_rollout_lanev<5, false>"], 100),
    ("k_rollout_lanev<3>@100", "lane3_rollout", ["k_rollout_lanev<3, true, 2>", "k_rollout_lanev<3, false, 2>", "k_rollout_lanev<3, true>", "k_rollout_lanev<3, false>"], 100),
    ("k_rollout_lane<5>@100", "c5s_rollout", ["k_rollout_lane<5, true>", "k_rollout_lane<5, false>"], 100),
    ("k_rollout_lane<3>@100", "lane3_rollout", ["k_rollout_lane<3, true>", "k_rollout_lane<3, false>"], 100),
    ("k_step<3,1> + k_map<3>@1", "c4_step", ["k_map<3>", "k_step<3, 1>"], 1),
    ("k_flight_pipe<3>@1", "c4_rollout", ["k_flight_pipe<3>", "k_map<3>", "k_step<3, 1>"], 1),
]
DOUBLED = ("k_map", "k_flight_pipe")


def part(tag, kernel):
    ent = raw[tag]
    k = ent["kernels"].get(kernel)
    if k is None:
        return None
    if "FETCH_SIZE_KB_total" not in k or "WRITE_SIZE_KB_total" not in k:
        raise SystemExit(f"{tag} / {kernel}: a PMC pass is missing (FETCH_SIZE or WRITE_SIZE) -- refusing a partial summary")
    f = 2.0 if kernel.startswith(DOUBLED) else 1.0
    fetch = k["FETCH_SIZE_KB_total"] * 1024 * f / ent["env_steps"]
    write = k["WRITE_SIZE_KB_total"] * 1024 / ent["env_steps"]
    return {"FETCH_SIZE_bytes_per_env_step": round(fetch, 1), "WRITE_SIZE_bytes_per_env_step": round(write, 1),
            "hbm_bytes_per_env_step": round(fetch + write, 1), "measured_on": tag, "kernel": kernel,
            "launches": k.get("launches"), "env_steps": ent["env_steps"]}


kernels = {}
for label, tag, names, spl in LABELS:
    if tag not in raw or label in kernels:
        continue
    parts = [p for p in (part(tag, k) for k in names) if p]
    if not parts:
        continue
    ent = {k: round(sum(p[k] for p in parts), 1) for k in
           ("FETCH_SIZE_bytes_per_env_step", "WRITE_SIZE_bytes_per_env_step", "hbm_bytes_per_env_step")}
    ent.update(measured_on=tag, env_steps=raw[tag]["env_steps"], steps_per_launch=spl)
    if len(parts) > 1:
        ent["parts"] = parts
    else:
        ent["kernel"] = parts[0]["kernel"]
    kernels[label] = ent
doc = ("HBM traffic per env-step from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/pmc_traffic.sh over "
       "tools/exp_workload.py, which runs a known number of env-steps; KB units x 1024, summed over every launch of the "
       "kernel; raw sums in r06_traffic_raw.json; this file = tools/make_traffic_json.py). FETCH_SIZE is doubled for k_map / "
       "k_flight_pipe (16 B/lane coalesced stream: gfx950 reports exactly half, MI355X_MICROARCH.md section HBM) and left raw "
       "for the other kernels (uncalibrated width). bench.py reports roofline.traffic = hbm_bytes_per_env_step x batch x "
       "steps per launch of ITS run. The rollout kernels refresh their MT19937 rows themselves (prologue / in-loop), so "
       "their figures include that traffic. k_flight_pipe<3> is the whole cs_rollout call (99 pipelined launches + the "
       "leading k_step and the trailing k_map per 100 steps).")
json.dump({"_doc": doc, "kernels": kernels}, open(out_path, "w"), indent=1)
print(json.dumps({k: v["hbm_bytes_per_env_step"] for k, v in kernels.items()}, indent=1))
