#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:?}"
cd "$R"
mkdir -p gpurun_out
: > gpurun_out/j_sweep.jsonl
for n in 3 5; do for v in A0F0D1 A0F1D1 A0F0D0 A1F0D0; do
  COOPSEARCH_LIB=$R/build/var/v${n}_$v.so python tools/oct_sweep.py --n $n --batches 4096,8192,16384 --kernels od,oct --reps 10 --tag $v >> gpurun_out/j_sweep.jsonl 2>> gpurun_out/j_sweep.err
done; done
cat gpurun_out/j_sweep.jsonl | python -c "
import sys, json
for ln in sys.stdin:
    d = json.loads(ln); print(d['tag'][-12:], d['n'], d['B'], d['kernel'], d['us_per_step'], '%.3e' % d['env_steps_per_s'])
"
