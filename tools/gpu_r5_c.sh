#!/bin/bash
# round 5, call C: GPU tests of the in-tree build (K: stage reads hoisted, early flow-control read; D / K ballots from bare comparisons,
# DPP bound_ctrl), then the variants side by side on ONE box.
set -u
R="${GRAFT_REPO_ROOT:?}"
cd "$R"
O=gpurun_out/r5c; rm -rf $O; mkdir -p $O
V=$R/build/var
python -m pytest tests -m gpu -q -x --durations=5 > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/gpu_tests.log
run() { lib=$1; shift; COOPSEARCH_LIB=$V/$lib.so timeout 600 python tools/exp_var_check.py "$@" 2>&1 | grep -v amdgpu.ids; }
{
run od5_base 5 od,ode 8192,16384 100 --nocheck
run od5_opt3 5 od,ode 8192,16384 100 --nocheck
run od5_opt4 5 od,ode 8192,16384 100
run od5_opt5 5 od,ode,oct 8192,16384,32768 100
run od3_base 3 ode 4096 100 --nocheck
run od3_opt 3 ode 4096 100 --nocheck
run od3_opt4 3 ode,od 4096,16384 100
run od3_opt5 3 ode,od 4096,16384 100
run od3_opt5np 3 ode 4096 100 --nocheck
run od3_base 3 ode 4096 20 --nocheck
run od3_opt5 3 ode 4096 20 --nocheck
run od3_opt5np 3 ode 4096 20 --nocheck
} | tee $O/od.txt
