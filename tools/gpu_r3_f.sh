#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:?}"
cd "$R"
for cfg in "3 8192" "5 8192" "3 65536"; do set -- $cfg; N=$1 B=$2 COOPSEARCH_LIB=$R/build/var/lib_tl$1.so python tools/exp_oct_timeline.py 2>&1 | grep -v amdgpu.ids | tail -3; done
