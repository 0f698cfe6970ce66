#!/bin/bash
# round 4, call D: k_rollout_lanev after the tape-in-LDS / miss-walk / chunked write-out changes: parity vs k_rollout_lane, speed at
# 2 and 3 wavefronts per SIMD, its timeline, the crossover against the octet kernel
set -u
R="${GRAFT_REPO_ROOT:?}"
cd "$R"
mkdir -p gpurun_out
for v in lv2_n3 lv3_n3 lv2_n5; do
  n=${v: -1}
  echo "== $v"; COOPSEARCH_LIB=$R/build/var/$v.so N=$n python tools/exp_lanev_check.py 262144 1048576 2>&1 | grep -v amdgpu.ids | tail -6
done
for n in 3 5; do
  COOPSEARCH_LIB=$R/build/var/tlv_n$n.so KERNEL=lanev N=$n B=262144 python tools/exp_lane_timeline.py 2>&1 | grep -v amdgpu.ids
  COOPSEARCH_LIB=$R/build/var/tlv_n$n.so KERNEL=lanev N=$n B=262144 python tools/exp_clock.py 2>&1 | grep -v amdgpu.ids
done
for n in 3 5; do for k in oct lanev; do
  COOPSEARCH_LIB=$R/build/var/lv2_n$n.so python tools/quick_lane.py $n $k 16384 32768 65536 131072 2>&1 | grep -v amdgpu.ids
done; done
