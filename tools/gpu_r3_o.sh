#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:?}"
cd "$R"
mkdir -p gpurun_out/o
timeout 2400 python -m pytest tests/test_gpu_parity.py -x -q -k "octet or stepwise or (long_horizon and (oct or od)) or full_size or interleaved" > gpurun_out/o/tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/o/tests.log
for cfg in "3 4096" "5 8192"; do set -- $cfg; N=$1 B=$2 COOPSEARCH_LIB=$R/build/var/lib_tl$1.so python tools/exp_od_timeline.py 2>&1 | grep -v amdgpu; done
for rep in 1 2; do
python tools/oct_sweep.py --n 3,5 --batches 4096,8192,16384 --kernels od --reps 10 --tag async 2>/dev/null
for n in 3 5; do COOPSEARCH_LIB=$R/build/var/v${n}_sync.so python tools/oct_sweep.py --n $n --batches 4096,8192,16384 --kernels od --reps 10 --tag sync 2>/dev/null; done
python tools/oct_sweep.py --n 3 --batches 4096 --kernels od --T 20 --reps 30 --tag async-T20 2>/dev/null
COOPSEARCH_LIB=$R/build/var/v3_sync.so python tools/oct_sweep.py --n 3 --batches 4096 --kernels od --T 20 --reps 30 --tag sync-T20 2>/dev/null
done | python -c "
import sys, json
for ln in sys.stdin:
    d = json.loads(ln); print(d['tag'][-12:], d['n'], d['B'], d['kernel'], d['us_per_step'], '%.3e' % d['env_steps_per_s'])
"
python bench.py --gpus 1 --steps 20 --warmup 5 --no-also --no-cpu-baseline > gpurun_out/o/bench20.json 2> gpurun_out/o/bench20.err; echo "bench20 rc=$? lines=$(wc -l < gpurun_out/o/bench20.json)"
python -c "
import json; d=json.loads(open('gpurun_out/o/bench20.json').read()); print('bench20', '%.4e' % d['value'], d['timing']['region_ms_min_median_max'])"
