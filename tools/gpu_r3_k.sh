#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:?}"
cd "$R"
mkdir -p gpurun_out
: > gpurun_out/k_sweep.jsonl
for rep in 1 2; do for n in 3 5; do for v in base trigD; do
  COOPSEARCH_LIB=$R/build/var/v${n}_$v.so python tools/oct_sweep.py --n $n --batches 4096,8192,16384 --kernels od --reps 10 --tag $v >> gpurun_out/k_sweep.jsonl 2>> gpurun_out/k_sweep.err
done; done; done
cat gpurun_out/k_sweep.jsonl | python -c "
import sys, json
for ln in sys.stdin:
    d = json.loads(ln); print(d['tag'][-12:], d['n'], d['B'], d['kernel'], d['us_per_step'], '%.3e' % d['env_steps_per_s'])
"
COOPSEARCH_LIB=$R/build/var/v3_trigD.so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "(octet and od and (flight_easy-3 or frozen-flight_easy-3)) or (stepwise and od)" 2>&1 | tail -3
