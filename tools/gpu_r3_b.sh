#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:?}"
cd "$R"
mkdir -p gpurun_out
python -m pytest tests/test_gpu_rccl.py -x -q > gpurun_out/b_rccl.log 2>&1; echo "rccl rc=$?"; tail -5 gpurun_out/b_rccl.log
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "octet or stepwise or (long_horizon and oct) or full_size_rollout" > gpurun_out/b_oct.log 2>&1; echo "oct rc=$?"; tail -15 gpurun_out/b_oct.log
for cfg in "3 8192" "5 8192" "5 16384"; do set -- $cfg; N=$1 B=$2 COOPSEARCH_LIB=$R/build/var/lib_tl$1.so python tools/exp_oct_timeline.py 2>&1 | tail -10; done
python tools/oct_sweep.py --n 3,5 --batches 4096,8192,16384,32768,65536,262144 --tag w4 > gpurun_out/b_sweep_w4.jsonl 2> gpurun_out/b_sweep_w4.err; echo "sweep w4 rc=$?"
for w in 2 3; do
  COOPSEARCH_LIB=$R/build/var/lib_w$w.so python tools/oct_sweep.py --n 3,5 --batches 4096,8192,16384,32768,65536,262144 --kernels oct --tag w$w > gpurun_out/b_sweep_w$w.jsonl 2> gpurun_out/b_sweep_w$w.err; echo "sweep w$w rc=$?"
done
cat gpurun_out/b_sweep_w4.jsonl gpurun_out/b_sweep_w2.jsonl gpurun_out/b_sweep_w3.jsonl | python -c "
import sys, json
for ln in sys.stdin:
    d = json.loads(ln); print(d['tag'][-12:], d['n'], d['B'], d['kernel'], d['us_per_step'], '%.3e' % d['env_steps_per_s'], d['hbm_frac'])
"
