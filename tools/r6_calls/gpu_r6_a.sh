#!/bin/bash
# round 6, first call: the compact bench line as the driver runs it (stdout kept apart from stderr, sizes recorded), the bench-line tests
set -u
R="${GRAFT_REPO_ROOT:?}"
cd "$R"
mkdir -p gpurun_out/r6a
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6a/bench_driver_args.stdout 2> gpurun_out/r6a/bench_driver_args.stderr; echo "bench rc=$?"
wc -c gpurun_out/r6a/bench_driver_args.stdout
cp gpurun_out/bench_detail.json gpurun_out/r6a/bench_driver_args_detail.json
tail -c 8000 gpurun_out/r6a/bench_driver_args.stdout | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('parsed', d['value'], d['roofline']['frac'], d['cpu_baseline']['value'], d['also_summary'])"
python -m pytest tests/test_gpu_multi.py tests/test_gpu_rccl.py -m gpu -q -x > gpurun_out/r6a/tests_multi.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r6a/tests_multi.log
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "action or check" > gpurun_out/r6a/tests_actions.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r6a/tests_actions.log
