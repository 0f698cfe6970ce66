#!/bin/bash
# round 4, call C: the clock the big kernels really run at; first run of k_rollout_lanev (2-wave build: no spills; default build: spills)
set -u
R="${GRAFT_REPO_ROOT:?}"
cd "$R"
mkdir -p gpurun_out
for cfg in "3 lane 262144" "5 lane 262144" "3 oct 262144" "5 oct 262144" "3 oct 32768" "5 oct 32768"; do
  set -- $cfg
  COOPSEARCH_LIB=$R/build/var/tl_n$1.so N=$1 KERNEL=$2 B=$3 python tools/exp_clock.py 2>&1 | grep -v amdgpu.ids
done
for v in lv2 lv; do for n in 3 5; do
  echo "== $v n=$n"; COOPSEARCH_LIB=$R/build/var/${v}_n$n.so N=$n python tools/exp_lanev_check.py 262144 1048576 2>&1 | grep -v amdgpu.ids | tail -8
done; done
