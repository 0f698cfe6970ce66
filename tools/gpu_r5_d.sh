#!/bin/bash
# round 5, call D: k_rollout_lanev with the target count kept a scalar (SGPR spills out of the state deposit): region frequencies,
# before / after on one box, refresh threshold A/B.
set -u
R="${GRAFT_REPO_ROOT:?}"
cd "$R"
O=gpurun_out/r5d; rm -rf $O; mkdir -p $O
V=$R/build/var
for n in 3 5; do python tools/spill_exec.py run $n 262144 2>&1 | grep -v amdgpu.ids | tee $O/spill_exec_n$n.txt; done
{
for rep in 1 2; do
for lib in od3_opt5 lv3_fix lv3_fix224; do COOPSEARCH_LIB=$V/$lib.so python tools/quick_lane.py 3 lanev 65536 262144 1048576 2>&1 | grep "n=" | sed "s/^/$lib /"; done
for lib in od5_opt5 lv5_fix lv5_fix288; do COOPSEARCH_LIB=$V/$lib.so python tools/quick_lane.py 5 lanev 65536 262144 1048576 2>&1 | grep "n=" | sed "s/^/$lib /"; done
done
} | tee $O/lanev_ab.txt
COOPSEARCH_LIB=$V/lv5_fix.so timeout 600 python tools/exp_var_check.py 5 lanev,lane 65536 100 2>&1 | grep -v amdgpu.ids | tee $O/check5.txt
COOPSEARCH_LIB=$V/lv3_fix.so timeout 600 python tools/exp_var_check.py 3 lanev,lane 65536 100 2>&1 | grep -v amdgpu.ids | tee $O/check3.txt
