#!/bin/bash
# round 4, call J: closed loop with fewer barriers (x kept in place, ping-pong hidden planes, in-wavefront selection) and tape draws:
# parity at 3 agents on the experimental builds, then the rates
set -u
R="${GRAFT_REPO_ROOT:?}"
cd "$R"
mkdir -p gpurun_out
for v in pol2_n3 pol3_n3; do
  COOPSEARCH_LIB=$R/build/var/$v.so timeout 900 python -m pytest tests/test_gpu_policy.py -m gpu -q -k "(fused_closed_loop_rollout_equals_stepwise and 3-6) or (fused_closed_loop_rollout_equals_stepwise and 3-37) or fused_forward_matches_torch_module or epsilon_step_schedule or fused_forward_matches_reference" > gpurun_out/j_tests_$v.log 2>&1; echo "$v parity rc=$?"; tail -4 gpurun_out/j_tests_$v.log
done
for v in pol_n3 pol2_n3 pol3_n3; do echo "== $v"; COOPSEARCH_LIB=$R/build/var/$v.so timeout 600 python tools/exp_closed_loop.py easy 2>&1 | grep -v amdgpu.ids; done
