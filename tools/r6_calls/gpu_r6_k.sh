#!/bin/bash
# round 6: the lane kernel's refresh row requested a step ahead straight into LDS (global_load_lds) with a counted wait (CS_LV_DMA=1)
# against round 4's order (registers, requested after the kinematics of the same step): bit-identity, A/B timing, phase timeline
set -u
R="${GRAFT_REPO_ROOT:?}"
cd "$R"
mkdir -p gpurun_out/r6k
for v in dma1_n5 dma1_n3 dmasafe_n5; do
  n=${v: -1}
  COOPSEARCH_LIB=build/var/$v.so python tools/exp_var_check.py $n lanev 4096 100 > gpurun_out/r6k/check_$v.log 2>&1; echo "check $v rc=$?"; grep -c "bit-identical" gpurun_out/r6k/check_$v.log; tail -1 gpurun_out/r6k/check_$v.log
done
for pass in 1 2; do
for v in dma0_n5 dma1_n5 dmasafe_n5; do
  COOPSEARCH_LIB=build/var/$v.so python tools/exp_var_check.py 5 lanev 65536,131072,262144,1048576 100 --nocheck 2>&1 | grep -v amdgpu.ids | sed "s/^/$v /"
done
for v in dma0_n3 dma1_n3; do
  COOPSEARCH_LIB=build/var/$v.so python tools/exp_var_check.py 3 lanev 65536,131072,262144 100 --nocheck 2>&1 | grep -v amdgpu.ids | sed "s/^/$v /"
done
done 2>&1 | tee gpurun_out/r6k/ab.log
for B in 65536 262144; do COOPSEARCH_LIB=build/var/tl_n5.so python tools/lanev_timeline.py 5 $B 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r6k/lanev5_timeline_dma.log
