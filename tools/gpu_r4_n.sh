#!/bin/bash
# round 4, call N: k_policy_h with double-buffered tile inputs and the fc3 fragments in LDS (polh3) against the shipped one; timeline of
# the c2 pair kernel (where do D's resets and K's fixes go in a 20-step launch)
set -u
R="${GRAFT_REPO_ROOT:?}"
cd "$R"
mkdir -p gpurun_out
COOPSEARCH_LIB=$R/build/var/polh3_n3.so timeout 900 python -m pytest tests/test_gpu_policy.py -m gpu -q -k "forward or stepwise or epsilon_step or select or softmax" > gpurun_out/n_tests_polh3.log 2>&1; echo "polh3 parity rc=$?"; tail -4 gpurun_out/n_tests_polh3.log
for v in pol5_n3 polh3_n3; do echo "== $v"; COOPSEARCH_LIB=$R/build/var/$v.so timeout 600 python tools/exp_closed_loop.py all 2>&1 | grep -v amdgpu.ids; done
echo "== od timeline"
N=3 B=4096 COOPSEARCH_LIB=$R/build/var/tlp_n3.so timeout 300 python tools/exp_od_timeline.py 2>&1 | grep -v amdgpu.ids
