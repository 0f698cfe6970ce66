#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:?}"
cd "$R"
mkdir -p gpurun_out/q
timeout 2400 python -m pytest tests/test_gpu_parity.py -x -q -k "octet or stepwise or long_horizon or full_size or golden or interleaved" > gpurun_out/q/tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/q/tests.log
for cfg in "3 4096" "5 8192"; do set -- $cfg; N=$1 B=$2 COOPSEARCH_LIB=$R/build/var/lib_tl$1.so timeout 300 python tools/exp_od_timeline.py 2>&1 | grep -v amdgpu; done
timeout 600 python tools/oct_sweep.py --n 3,5 --batches 4096,8192,16384,32768 --kernels od,oct --reps 10 --tag ring4 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    d = json.loads(ln); print(d['tag'][-12:], d['n'], d['B'], d['kernel'], d['us_per_step'], '%.3e' % d['env_steps_per_s'])
"
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-also --no-cpu-baseline > gpurun_out/q/bench20.json 2> gpurun_out/q/bench20.err; echo "bench20 rc=$?"
timeout 300 python bench.py --no-also --no-cpu-baseline > gpurun_out/q/bench100.json 2> /dev/null
python -c "
import json
for f in ('bench20','bench100'):
    d=json.loads(open('gpurun_out/q/%s.json' % f).read()); print(f, '%.4e' % d['value'], d['timing']['region_ms_min_median_max'])"
