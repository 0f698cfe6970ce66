#!/bin/bash
# round 4, call L: closed loop without the a planes / fused split_sum (pol4), and compiled for two wavefronts per SIMD (pol4w2)
set -u
R="${GRAFT_REPO_ROOT:?}"
cd "$R"
mkdir -p gpurun_out
env | grep -i -E "visible|ordinal|^OMP|^GOMP" 
for v in pol4_n3 pol4w2_n3; do
  COOPSEARCH_LIB=$R/build/var/$v.so timeout 900 python -m pytest tests/test_gpu_policy.py -m gpu -q -k "(fused_closed_loop_rollout_equals_stepwise and 3-6) or (fused_closed_loop_rollout_equals_stepwise and 3-37) or fused_forward_matches_torch_module or epsilon_step_schedule or fused_forward_matches_reference" > gpurun_out/l_tests_$v.log 2>&1; echo "$v parity rc=$?"; tail -4 gpurun_out/l_tests_$v.log
done
for v in pol3_n3 pol4_n3 pol4w2_n3; do echo "== $v"; COOPSEARCH_LIB=$R/build/var/$v.so timeout 600 python tools/exp_closed_loop.py easy 2>&1 | grep -v amdgpu.ids; done
