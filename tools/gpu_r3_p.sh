#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:?}"
cd /tmp && export TMPDIR=/tmp
for v in async sync; do
  if [ $v = sync ]; then export COOPSEARCH_LIB=$R/build/var/v3_sync.so; else unset COOPSEARCH_LIB; fi
  for T in 1 20 100; do
    rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/p/$v$T" -o p -- python3 "$R/tools/exp_workload.py" flight_easy 3 od 4096 rollout 300 $T > "$R/gpurun_out/p/$v$T.log" 2>&1
    echo "$v T=$T"; python3 "$R/tools/prof_summary.py" "$R/gpurun_out/p/$v$T/p_results.db" | grep k_rollout_od | cut -c60-140
    rm -f "$R/gpurun_out/p/$v$T/p_results.db"
  done
done
