#!/bin/bash
# round 4, call H: the whole GPU suite with the split-fp16 policy kernels; closed-loop rates against the fp32 matrix path
set -u
R="${GRAFT_REPO_ROOT:?}"
cd "$R"
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --durations=10 > gpurun_out/h_gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -30 gpurun_out/h_gpu_tests.log
for v in pol32_n3 pol_n3; do echo "== $v"; COOPSEARCH_LIB=$R/build/var/$v.so python tools/exp_closed_loop.py all 2>&1 | grep -v amdgpu.ids; done
echo "== shipped library"; python tools/exp_closed_loop.py all 2>&1 | grep -v amdgpu.ids
