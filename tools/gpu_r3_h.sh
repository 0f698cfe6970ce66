#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:?}"
cd "$R"
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_parity.py -x -q -k "(octet and od) or (stepwise and od) or (long_horizon and od)" > gpurun_out/h_od.log 2>&1; echo "od rc=$?"; tail -12 gpurun_out/h_od.log
python tools/oct_sweep.py --n 3,5 --batches 4096,8192,16384,32768 --kernels duo,oct,od --tag od4 > gpurun_out/h_sweep.jsonl 2> gpurun_out/h_sweep.err; echo "sweep rc=$?"
python tools/oct_sweep.py --n 3,5 --batches 4096,8192 --kernels duo,oct,od --T 20 --reps 12 --tag od4-T20 >> gpurun_out/h_sweep.jsonl 2>> gpurun_out/h_sweep.err
for w in 2 3; do
  COOPSEARCH_LIB=$R/build/var/lib_od$w.so python tools/oct_sweep.py --n 3,5 --batches 4096,8192,16384,32768 --kernels od --tag od$w >> gpurun_out/h_sweep.jsonl 2>> gpurun_out/h_sweep.err
done
cat gpurun_out/h_sweep.jsonl | python -c "
import sys, json
for ln in sys.stdin:
    d = json.loads(ln); print(d['tag'][-12:], d['n'], d['B'], d['kernel'], d['us_per_step'], '%.3e' % d['env_steps_per_s'], d['hbm_frac'])
"
tail -5 gpurun_out/h_sweep.err
