#!/bin/bash
# round 4, call E: the whole GPU suite on the rebuilt library (k_rollout_lanev default from 65536 envs, exploration schedule, ABI 6)
set -u
R="${GRAFT_REPO_ROOT:?}"
cd "$R"
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/e_gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -40 gpurun_out/e_gpu_tests.log
