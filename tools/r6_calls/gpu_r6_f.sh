#!/bin/bash
# round 6, sixth call: knob sweep of the 5-agent pair kernels (BASELINE configs 3 and 5): shared-reciprocal division in K, role order
# and priorities of the three-wavefront variant.  us per step, A/B on one box, two passes.
set -u
R="${GRAFT_REPO_ROOT:?}"
cd "$R"
mkdir -p gpurun_out/r6f
for pass in 1 2; do
for v in k_base_n5 k_sdiv_n5 k_kde_n5 k_dek_n5 k_dp3_n5 k_dp1_n5 k_k2d3_n5; do
  COOPSEARCH_LIB=build/var/$v.so python tools/exp_var_check.py 5 ode 8192 100 --nocheck 2>&1 | grep -v amdgpu.ids | sed "s/^/$v /"
done
for v in k_base_n5 k_sdiv_n5; do
  COOPSEARCH_LIB=build/var/$v.so python tools/exp_var_check.py 5 od 16384 100 --nocheck 2>&1 | grep -v amdgpu.ids | sed "s/^/$v /"
done
done 2>&1 | tee gpurun_out/r6f/sweep.log
COOPSEARCH_LIB=build/var/k_sdiv_n5.so python tools/exp_var_check.py 5 od,ode 2048 100 > gpurun_out/r6f/check_sdiv.log 2>&1; echo "check sdiv rc=$?"; grep -c bit-identical gpurun_out/r6f/check_sdiv.log
