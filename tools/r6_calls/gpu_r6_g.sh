#!/bin/bash
# round 6, seventh call: the shared-reciprocal division in the three-wavefront variant (teams of 5 and 6: bit-identity against the
# step kernel on the crowded configuration too), then the whole measurement set again on the final kernels
set -u
R="${GRAFT_REPO_ROOT:?}"
cd "$R"
mkdir -p gpurun_out/r6g
for v in k_final_n5 k_final_n6; do
  n=${v: -1}
  COOPSEARCH_LIB=build/var/$v.so python tools/exp_var_check.py $n od,ode 8192 100 > gpurun_out/r6g/check_$v.log 2>&1; echo "check $v rc=$?"; grep -c "bit-identical" gpurun_out/r6g/check_$v.log; tail -2 gpurun_out/r6g/check_$v.log
done
bash tools/gpu_r6_final.sh
