#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:?}"
cd "$R"
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_parity.py -x -q -k "octet or stepwise or (long_horizon and (oct or od))" > gpurun_out/i_od.log 2>&1; echo "oct+od rc=$?"; tail -6 gpurun_out/i_od.log
for cfg in "3 4096" "5 8192"; do set -- $cfg; N=$1 B=$2 COOPSEARCH_LIB=$R/build/var/lib_tl$1.so python tools/exp_od_timeline.py 2>&1 | grep -v amdgpu; done
python tools/oct_sweep.py --n 3,5 --batches 4096,8192,16384,32768 --kernels duo,oct,od --tag T100 > gpurun_out/i_sweep.jsonl 2> gpurun_out/i_sweep.err; echo "sweep rc=$?"
for T in 1 5 20; do python tools/oct_sweep.py --n 3 --batches 4096 --kernels duo,od --T $T --reps 20 --tag T$T >> gpurun_out/i_sweep.jsonl 2>> gpurun_out/i_sweep.err; done
cat gpurun_out/i_sweep.jsonl | python -c "
import sys, json
for ln in sys.stdin:
    d = json.loads(ln); print(d['tag'][-12:], d['n'], d['B'], d['kernel'], d['us_per_step'], '%.3e' % d['env_steps_per_s'], d['hbm_frac'])
"
