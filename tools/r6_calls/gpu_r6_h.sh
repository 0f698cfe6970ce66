#!/bin/bash
# round 6, eighth call: k_rollout_oct compiled for four wavefronts per SIMD (128 VGPRs: 32768 envs = 1024 workgroups then fit the chip in
# ONE resident round instead of 1.33) against the shipped three, over the batches the kernel is dispatched for
set -u
R="${GRAFT_REPO_ROOT:?}"
cd "$R"
mkdir -p gpurun_out/r6h
for v in w4_n5 w4_n3; do
  n=${v: -1}
  COOPSEARCH_LIB=build/var/$v.so python tools/exp_var_check.py $n oct 8192 100 > gpurun_out/r6h/check_$v.log 2>&1; echo "check $v rc=$?"; grep -c "bit-identical" gpurun_out/r6h/check_$v.log
done
for pass in 1 2; do
for v in k_final_n5 w4_n5; do
  COOPSEARCH_LIB=build/var/$v.so python tools/exp_var_check.py 5 oct 20480,24576,32768,40960,49152,65536 100 --nocheck 2>&1 | grep -v amdgpu.ids | sed "s/^/$v /"
done
for v in w3_n3 w4_n3; do
  COOPSEARCH_LIB=build/var/$v.so python tools/exp_var_check.py 3 oct 20480,24576,32768,40960,49152,65536 100 --nocheck 2>&1 | grep -v amdgpu.ids | sed "s/^/$v /"
done
done 2>&1 | tee gpurun_out/r6h/ab.log
