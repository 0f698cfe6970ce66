#!/bin/bash
# kernel-only durations of the 20-step and 100-step c2 launches (no side workloads in the trace)
set -u
R="${GRAFT_REPO_ROOT:?}"
cd /tmp && export TMPDIR=/tmp
for T in 20 100; do
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/s_T$T -- python3 $R/bench.py --steps $T --warmup 5 --no-also --no-cpu-baseline > $R/gpurun_out/s_T$T.log 2>&1
  tail -1 $R/gpurun_out/s_T$T.log | cut -c1-300
  python3 - <<PY
import csv, glob, statistics
f = glob.glob("$R/gpurun_out/s_T$T/**/*kernel_trace.csv", recursive=True)[0]
d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if "k_rollout_od" in r["Kernel_Name"]]
d.sort()
n = len(d)
print("T=$T launches", n, "min %.1f p10 %.1f median %.1f p90 %.1f max %.1f mean %.1f us" % (d[0]/1e3, d[n//10]/1e3, d[n//2]/1e3, d[9*n//10]/1e3, d[-1]/1e3, statistics.mean(d)/1e3))
PY
done
