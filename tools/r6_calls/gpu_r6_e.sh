#!/bin/bash
# round 6, fifth call: smoke + the whole GPU suite after the removal of the round-1/2 rollout kernels, the render test, the extended
# check_actions test; then the bench line in the driver's command shape once more
set -u
R="${GRAFT_REPO_ROOT:?}"
cd "$R"
mkdir -p gpurun_out/r6e
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6e/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r6e/smoke.log
python -m pytest tests -m gpu -q --durations=8 > gpurun_out/r6e/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -14 gpurun_out/r6e/gpu_tests.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6e/bench_driver_args.json 2> gpurun_out/r6e/bench_driver_args.err; echo "bench rc=$?"; wc -c gpurun_out/r6e/bench_driver_args.json
