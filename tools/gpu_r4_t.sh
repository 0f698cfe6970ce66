#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:?}"
cd "$R"
for i in 1 2 3 4 5 6; do N=5 B=262144 KERNEL=lanev COOPSEARCH_LIB=$R/build/var/tlp_n5.so timeout 200 python tools/exp_clock.py 2>&1 | grep -v amdgpu.ids; done
for i in 1 2 3; do N=5 B=262144 KERNEL=lanev T=25 COOPSEARCH_LIB=$R/build/var/tlp_n5.so timeout 200 python tools/exp_clock.py 2>&1 | grep -v amdgpu.ids; done
rocm-smi --showclocks --showpower --showtemp 2>/dev/null | head -30
