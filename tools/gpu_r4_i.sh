#!/bin/bash
# round 4, call I: the role-split closed loop (k_rollout_policy_roles): parity at 3 agents on the experimental build, then its rate
set -u
R="${GRAFT_REPO_ROOT:?}"
cd "$R"
mkdir -p gpurun_out
COOPSEARCH_LIB=$R/build/var/rp_n3.so timeout 900 python -m pytest tests/test_gpu_policy.py -m gpu -q -x -k "fused_closed_loop_rollout_equals_stepwise and 3-" > gpurun_out/i_tests.log 2>&1; echo "roles parity rc=$?"; tail -15 gpurun_out/i_tests.log
for v in pol_n3 rp_n3; do echo "== $v"; COOPSEARCH_LIB=$R/build/var/$v.so timeout 600 python tools/exp_closed_loop.py easy 2>&1 | grep -v amdgpu.ids; done
