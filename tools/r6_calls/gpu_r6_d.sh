#!/bin/bash
# round 6, fourth call: K's flow-control read issued in the middle of the step as compiler-visible volatile loads (CS_OD_EARLY_PEEK=1)
# against the blocking read at the loop head (0, default): bit-identity, then A/B, two passes
set -u
R="${GRAFT_REPO_ROOT:?}"
cd "$R"
mkdir -p gpurun_out/r6d
for v in peek1_n5 peek1_n3; do
  n=${v: -1}
  COOPSEARCH_LIB=build/var/$v.so python tools/exp_var_check.py $n od,ode 4096 100 > gpurun_out/r6d/check_$v.log 2>&1; echo "check $v rc=$?"; grep -c "bit-identical" gpurun_out/r6d/check_$v.log
done
for pass in 1 2; do
for v in peek0_n5 peek1_n5; do
  COOPSEARCH_LIB=build/var/$v.so python tools/exp_var_check.py 5 od 16384 100 --nocheck 2>&1 | grep -v amdgpu.ids | sed "s/^/$v /"
  COOPSEARCH_LIB=build/var/$v.so python tools/exp_var_check.py 5 ode 8192 100 --nocheck 2>&1 | grep -v amdgpu.ids | sed "s/^/$v /"
done
for v in peek0_n3 peek1_n3; do
  COOPSEARCH_LIB=build/var/$v.so python tools/exp_var_check.py 3 od 16384 100 --nocheck 2>&1 | grep -v amdgpu.ids | sed "s/^/$v /"
  COOPSEARCH_LIB=build/var/$v.so python tools/exp_var_check.py 3 ode 4096,8192 100 --nocheck 2>&1 | grep -v amdgpu.ids | sed "s/^/$v /"
  COOPSEARCH_LIB=build/var/$v.so python tools/exp_var_check.py 3 ode 4096 20 --nocheck 2>&1 | grep -v amdgpu.ids | sed "s/^/$v /"
done
done 2>&1 | tee gpurun_out/r6d/ab.log
