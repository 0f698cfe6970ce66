#!/bin/bash
# round 6, second call: the packed-fp32 sensor pre-filter of the octet kernels (oct_detect_impl<PRE>) against the fp64-only build:
# bit-identity with the step kernel (exp_var_check), then A/B timing, two passes
set -u
R="${GRAFT_REPO_ROOT:?}"
cd "$R"
mkdir -p gpurun_out/r6b
for v in pre1_n5 prew_n5 pre1_n3; do
  n=${v: -1}
  COOPSEARCH_LIB=build/var/$v.so python tools/exp_var_check.py $n od,ode,oct 8192 100 > gpurun_out/r6b/check_$v.log 2>&1; echo "check $v rc=$?"; grep -c "bit-identical" gpurun_out/r6b/check_$v.log
done
for pass in 1 2; do
for v in pre0_n5 pre1_n5; do
  COOPSEARCH_LIB=build/var/$v.so python tools/exp_var_check.py 5 od 8192,16384 100 --nocheck 2>&1 | grep -v amdgpu.ids | sed "s/^/$v /"
  COOPSEARCH_LIB=build/var/$v.so python tools/exp_var_check.py 5 ode 8192 100 --nocheck 2>&1 | grep -v amdgpu.ids | sed "s/^/$v /"
  COOPSEARCH_LIB=build/var/$v.so python tools/exp_var_check.py 5 oct 32768 100 --nocheck 2>&1 | grep -v amdgpu.ids | sed "s/^/$v /"
done
for v in pre0_n3 pre1_n3; do
  COOPSEARCH_LIB=build/var/$v.so python tools/exp_var_check.py 3 od 16384 100 --nocheck 2>&1 | grep -v amdgpu.ids | sed "s/^/$v /"
  COOPSEARCH_LIB=build/var/$v.so python tools/exp_var_check.py 3 ode 4096,8192 100 --nocheck 2>&1 | grep -v amdgpu.ids | sed "s/^/$v /"
  COOPSEARCH_LIB=build/var/$v.so python tools/exp_var_check.py 3 oct 32768 100 --nocheck 2>&1 | grep -v amdgpu.ids | sed "s/^/$v /"
done
done 2>&1 | tee gpurun_out/r6b/ab.log
