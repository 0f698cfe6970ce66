#!/bin/bash
R="${GRAFT_REPO_ROOT:?}"
cd /tmp && export TMPDIR=/tmp
export COOPSEARCH_LIB=$R/cooperative-search_amd/csrc/libcs_pipe_b.so
for v in "$@"; do
  export CS_PIPE_SPREAD=$v
  rm -rf /tmp/pm_$v
  rocprofv3 --kernel-trace --stats -d /tmp/pm_$v -o m -- python3 $R/tools/exp_workload.py flight 3 auto 8192 rollout 3 100 > /tmp/pm_$v.log 2>&1
  echo "== spread $v"; python3 $R/tools/prof_summary.py /tmp/pm_$v/m_results.db | grep -E "k_flight"
done
