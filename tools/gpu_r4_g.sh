#!/bin/bash
# round 4, call G: the exploration-schedule tests; k_rollout_lanev A/B on one box (refresh request before / after the kinematics,
# shared-reciprocal division on / off); per-workgroup stamps of a 20-step launch of the c2 kernel
set -u
R="${GRAFT_REPO_ROOT:?}"
cd "$R"
mkdir -p gpurun_out
python -m pytest tests/test_gpu_policy.py tests/test_gpu_binding.py -m gpu -q -k "epsilon or schedule or binding or agree" > gpurun_out/g_eps_tests.log 2>&1; echo "eps tests rc=$?"; tail -8 gpurun_out/g_eps_tests.log
for rep in 1 2; do for v in lvA lvB lvC lvD; do
  COOPSEARCH_LIB=$R/build/var/${v}_n5.so python tools/quick_lane.py 5 lanev 262144 1048576 2>&1 | grep -v amdgpu.ids | sed "s/^/$v /"
done; done
for rep in 1 2; do for v in lvA lvC lvD; do
  COOPSEARCH_LIB=$R/build/var/${v}_n3.so python tools/quick_lane.py 3 lanev 262144 1048576 2>&1 | grep -v amdgpu.ids | sed "s/^/$v /"
done; done
COOPSEARCH_LIB=$R/build/var/tl_n3.so python tools/exp_od_blocks.py --n 3 --B 4096 --T 20 --reps 5 2>&1 | grep -v amdgpu.ids | tail -40
