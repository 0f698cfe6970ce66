#!/bin/bash
# flight c4: kernel-trace stats of the per-step loop (k_step + k_map per step) and of the pipelined cs_rollout call
# (k_flight_pipe), with the in-tree library or with variants built with other -DCS_PIPE_ILP / -DCS_PIPE_WAVES /
# -DCS_MAP_ILP values:  bash tools/exp_map.sh [path/to/variant.so ...]      (on the GPU box; profiles/r02_flight_pipe.md)
R="${GRAFT_REPO_ROOT:?}"
cd /tmp && export TMPDIR=/tmp
run() {  # tag, mode
  rm -rf /tmp/pm_$1_$2
  rocprofv3 --kernel-trace --stats -d /tmp/pm_$1_$2 -o m -- python3 $R/tools/exp_workload.py flight 3 auto 8192 $2 3 100 > /tmp/pm_$1_$2.log 2>&1
  echo "== $1 $2"; python3 $R/tools/prof_summary.py /tmp/pm_$1_$2/m_results.db | grep -E "k_map|k_step|k_flight"
}
if [ $# -eq 0 ]; then run intree step; run intree rollout; fi
for so in "$@"; do export COOPSEARCH_LIB=$so; tag=$(basename $so .so); run $tag step; run $tag rollout; done
