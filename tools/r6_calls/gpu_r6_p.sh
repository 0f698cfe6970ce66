#!/bin/bash
# round 6: the 312 hit bits of a refreshed row computed in one batch (row_hits_all: loads first, straight-line first-word decision, the rare
# second-word comparison behind one wave-uniform test) against the per-group loop: bit-identity of every kernel that refreshes rows, A/B
set -u
R="${GRAFT_REPO_ROOT:?}"
cd "$R"
mkdir -p gpurun_out/r6p
for v in new_n5 new_n3; do
  n=${v: -1}
  COOPSEARCH_LIB=build/var/$v.so python tools/exp_var_check.py $n od,ode,oct,lane,lanev 4096 100 > gpurun_out/r6p/check_$v.log 2>&1; echo "check $v rc=$?"; grep -c "bit-identical" gpurun_out/r6p/check_$v.log
done
for pass in 1 2; do
for v in old_n5 new_n5; do
  COOPSEARCH_LIB=build/var/$v.so python tools/exp_var_check.py 5 lanev 65536,262144 100 --nocheck 2>&1 | grep -v amdgpu.ids | sed "s/^/$v /"
  COOPSEARCH_LIB=build/var/$v.so python tools/exp_var_check.py 5 od 16384 100 --nocheck 2>&1 | grep -v amdgpu.ids | sed "s/^/$v /"
  COOPSEARCH_LIB=build/var/$v.so python tools/exp_var_check.py 5 ode 8192 100 --nocheck 2>&1 | grep -v amdgpu.ids | sed "s/^/$v /"
  COOPSEARCH_LIB=build/var/$v.so python tools/exp_var_check.py 5 oct 32768 100 --nocheck 2>&1 | grep -v amdgpu.ids | sed "s/^/$v /"
done
for v in old_n3 new_n3; do
  COOPSEARCH_LIB=build/var/$v.so python tools/exp_var_check.py 3 lanev 65536,262144 100 --nocheck 2>&1 | grep -v amdgpu.ids | sed "s/^/$v /"
  COOPSEARCH_LIB=build/var/$v.so python tools/exp_var_check.py 3 ode 4096 100 --nocheck 2>&1 | grep -v amdgpu.ids | sed "s/^/$v /"
  COOPSEARCH_LIB=build/var/$v.so python tools/exp_var_check.py 3 ode 4096 20 --nocheck 2>&1 | grep -v amdgpu.ids | sed "s/^/$v /"
  COOPSEARCH_LIB=build/var/$v.so python tools/exp_var_check.py 3 od 16384 100 --nocheck 2>&1 | grep -v amdgpu.ids | sed "s/^/$v /"
done
done 2>&1 | tee gpurun_out/r6p/ab.log
COOPSEARCH_LIB=build/var/tl_n5.so python tools/lanev_timeline.py 5 65536 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6p/timeline.log
