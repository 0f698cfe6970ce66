#!/bin/bash
# round 6, third call: smoke + the whole GPU suite on the split sources with the pre-filter in the pair kernel; then the
# three-wavefronts-per-SIMD build of k_rollout_lanev<5> against the shipped two-wavefront build
set -u
R="${GRAFT_REPO_ROOT:?}"
cd "$R"
mkdir -p gpurun_out/r6c
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6c/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r6c/smoke.log
python -m pytest tests -m gpu -q -x --durations=8 > gpurun_out/r6c/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -14 gpurun_out/r6c/gpu_tests.log
for pass in 1 2; do
for v in pre0_n5 lv3w_n5; do
  COOPSEARCH_LIB=build/var/$v.so python tools/exp_var_check.py 5 lanev 65536,262144,1048576 100 --nocheck 2>&1 | grep -v amdgpu.ids | sed "s/^/$v /"
done
done 2>&1 | tee gpurun_out/r6c/lanev5_waves.log
COOPSEARCH_LIB=build/var/lv3w_n5.so python tools/exp_var_check.py 5 lanev 4096 100 > gpurun_out/r6c/lv3w_check.log 2>&1; echo "lv3w check rc=$?"; tail -2 gpurun_out/r6c/lv3w_check.log
