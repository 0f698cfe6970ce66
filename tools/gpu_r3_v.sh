#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:?}"
cd "$R"
timeout 2000 python -m pytest tests/test_gpu_parity.py -x -q -k "octet or stepwise or long_horizon" 2>&1 | tail -4
for k in od ode; do
python tools/oct_sweep.py --n 3 --batches 4096 --kernels $k --reps 10 --tag $k-T100 2>/dev/null | grep '^{"tag'
python tools/oct_sweep.py --n 3 --batches 4096 --kernels $k --T 20 --reps 40 --tag $k-T20 2>/dev/null | grep '^{"tag'
done
python tools/oct_sweep.py --n 3,5 --batches 2048,8192,16384 --kernels od,ode --reps 6 --tag x 2>/dev/null | grep '^{"tag'
python bench.py --steps 20 --warmup 5 --no-also --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200
python bench.py --no-also --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200
