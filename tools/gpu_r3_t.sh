#!/bin/bash
# lean reset: parity of the octet kernels, then timing
set -u
R="${GRAFT_REPO_ROOT:?}"
cd "$R"
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "octet or stepwise or long_horizon" 2>&1 | tail -4
python tools/oct_sweep.py --n 3 --batches 4096 --kernels od --reps 10 --tag lean-T100 2>/dev/null
python tools/oct_sweep.py --n 3 --batches 4096 --kernels od --T 20 --reps 40 --tag lean-T20 2>/dev/null
python tools/oct_sweep.py --n 3,5 --batches 16384,65536 --kernels od,oct --reps 6 --tag lean 2>/dev/null
COOPSEARCH_LIB=$R/build/var/lib_tl3.so python tools/exp_od_events.py 2>&1 | grep -v amdgpu | cut -c1-250 | tail -12
python bench.py --steps 20 --warmup 5 --no-also --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200
python bench.py --no-also --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200
