#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:?}"
cd "$R"
mkdir -p gpurun_out/n
python bench.py --gpus 1 --steps 20 --warmup 5 --no-also --no-cpu-baseline > gpurun_out/n/bench20.json 2> gpurun_out/n/bench20.err; echo "bench20 rc=$? lines=$(wc -l < gpurun_out/n/bench20.json)"
python bench.py --gpus 1 --steps 20 --warmup 5 --no-also --no-cpu-baseline --no-graph > gpurun_out/n/bench20_nograph.json 2>/dev/null
python bench.py --no-also --no-cpu-baseline > gpurun_out/n/bench100.json 2>/dev/null
python - <<'PY'
import json
for f in ("bench20", "bench20_nograph", "bench100"):
    d = json.loads(open(f"gpurun_out/n/{f}.json").read())
    print(f, "%.4e" % d["value"], d["timing"]["region_ms_min_median_max"], d["config"]["hip_graph"])
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/n/trace20" -o p -- python3 "$R/tools/exp_workload.py" flight_easy 3 auto 4096 rollout 200 20 > "$R/gpurun_out/n/trace20.log" 2>&1
cd "$R"; python tools/prof_summary.py gpurun_out/n/trace20/p_results.db | head -5 | cut -c1-150; rm -f gpurun_out/n/trace20/p_results.db
for v in base lanetrig; do
  COOPSEARCH_LIB=$R/build/var/v3_$v.so python tools/oct_sweep.py --n 3 --batches 65536,262144,1048576 --kernels lane --reps 8 --tag $v 2>/dev/null
done | python -c "
import sys, json
for ln in sys.stdin:
    d = json.loads(ln); print(d['tag'][-12:], d['n'], d['B'], d['kernel'], d['us_per_step'], '%.3e' % d['env_steps_per_s'], d['hbm_frac'])
"
COOPSEARCH_LIB=$R/build/var/v3_lanetrig.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "lane and (flight_easy-3 or stepwise or interleaved or long_horizon)" 2>&1 | tail -3
python tools/batch_sweep.py > gpurun_out/batch_sweep.md 2> gpurun_out/batch_sweep.err; echo "sweep rc=$?"; tail -75 gpurun_out/batch_sweep.md
