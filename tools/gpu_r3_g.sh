#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:?}"
cd "$R"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "octet or stepwise or (long_horizon and oct) or full_size_rollout" > gpurun_out/g_oct.log 2>&1; echo "oct rc=$?"; tail -4 gpurun_out/g_oct.log
python tools/oct_sweep.py --n 3,5 --batches 4096,8192,16384,32768,65536 --kernels oct --tag w3 > gpurun_out/g_sweep_w3.jsonl 2> gpurun_out/g_sweep_w3.err; echo "sweep w3 rc=$?"
cat gpurun_out/g_sweep_w3.jsonl | python -c "
import sys, json
for ln in sys.stdin:
    d = json.loads(ln); print(d['tag'][-12:], d['n'], d['B'], d['kernel'], d['us_per_step'], '%.3e' % d['env_steps_per_s'], d['hbm_frac'])
"
