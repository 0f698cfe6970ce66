#!/bin/bash
# round 5, call B: experiments on one-team-size builds (build/var/*.so): the pair kernels' instruction trimming for teams of 5,
# four envs per workgroup at c2, the lane kernel's refresh threshold; then the default bench line.
set -u
R="${GRAFT_REPO_ROOT:?}"
cd "$R"
O=gpurun_out/r5b; rm -rf $O; mkdir -p $O
V=$R/build/var
run() { lib=$1; shift; COOPSEARCH_LIB=$V/$lib.so timeout 600 python tools/exp_var_check.py "$@" 2>&1 | grep -v amdgpu.ids; }
{
run od5_base 5 od,ode,oct 8192,16384,32768 100 --nocheck
run od5_opt1 5 od,ode,oct 8192,16384,32768 100
run od5_opt3 5 od,ode,oct 8192,16384,32768 100
run od5_opt2 5 od,ode,oct 8192,16384,32768 100
run od5_sdiv 5 od,ode 8192,16384 100 --nocheck
} | tee $O/od5.txt
{
run od3_base 3 ode,od 4096,16384 100 --nocheck
run od3_opt 3 ode,od 4096,16384 100
run od4_n3 3 ode,od 4096 100
run od3_base 3 ode 4096 20 --nocheck
run od3_opt 3 ode 4096 20 --nocheck
run od4_n3 3 ode 4096 20 --nocheck
} | tee $O/od3.txt
{
for lib in lv3 lv3_nrm; do COOPSEARCH_LIB=$V/$lib.so python tools/quick_lane.py 3 lanev 65536 262144 1048576 2>&1 | grep "n=" | sed "s/^/$lib /"; done
for lib in lv5 lv5_nrm; do COOPSEARCH_LIB=$V/$lib.so python tools/quick_lane.py 5 lanev 65536 262144 1048576 2>&1 | grep "n=" | sed "s/^/$lib /"; done
} | tee $O/lv_nrm.txt
cd /tmp && export TMPDIR=/tmp
for lib in od3_base od4_n3; do
  COOPSEARCH_LIB=$V/$lib.so timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/trace_$lib -o p -- python3 $R/tools/exp_workload.py flight_easy 3 ode 4096 rollout 200 20 > $R/$O/trace_$lib.log 2>&1
  echo "trace $lib rc=$?"
  python3 $R/tools/prof_summary.py $R/$O/trace_$lib/p_results.db 2>/dev/null | grep -i "rollout_od" | tee -a $R/$O/od3.txt
done
for lib in lv3_nrm lv5_nrm; do
  n=${lib:2:1}
  for c in FETCH_SIZE WRITE_SIZE; do
    COOPSEARCH_LIB=$V/$lib.so timeout 300 rocprofv3 --pmc $c --output-format csv -d $R/$O/pmc_${lib}_$c -- python3 $R/tools/exp_workload.py flight_easy $n lanev 262144 rollout 3 100 > $R/$O/pmc_${lib}_$c.log 2>&1
    echo "pmc $lib $c rc=$?"
  done
done
cd $R
python3 - $O <<'PY'
import csv, glob, sys, collections, re, os
out = sys.argv[1]
for d in sorted(glob.glob(out + "/pmc_*_SIZE")):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "k_rollout" in row["Kernel_Name"] and row["Counter_Name"].endswith("_SIZE"):
                m = re.search(r"(k_[a-z_0-9]+)(<[^>(]*>)?", row["Kernel_Name"])
                acc[(m.group(0), row["Counter_Name"])][0] += float(row["Counter_Value"]); acc[(m.group(0), row["Counter_Name"])][1] += 1
    for k, (v, n) in acc.items():
        print(os.path.basename(d), k, "KB total", round(v, 1), "launches", n, "B per env-step", round(v * 1024 / (n * 100 * 262144), 1))
PY
python tools/flight_sweep.py batch 2>&1 | grep "^|" | tee $O/flight_batch.md
python tools/flight_sweep.py teams 2>&1 | grep "^|" | tee $O/flight_teams.md
cd /tmp
for B in 8192 65536; do
  T=40; [ $B = 65536 ] && T=10
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --output-format csv -d $R/$O/pmcf_${B}_$c -- python3 $R/tools/exp_workload.py flight 3 auto $B rollout 2 $T > $R/$O/pmcf_${B}_$c.log 2>&1
    echo "pmc flight $B $c rc=$?"
  done
done
cd $R
python3 - $O <<'PY'
import csv, glob, sys, collections, re, os
out = sys.argv[1]
for d in sorted(glob.glob(out + "/pmcf_*_SIZE")):
    B = int(os.path.basename(d).split("_")[1]); T = 40 if B == 8192 else 10
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"].endswith("_SIZE"):
                m = re.search(r"(k_[a-z_0-9]+)(<[^>(]*>)?", row["Kernel_Name"])
                if m:
                    acc[(m.group(0), row["Counter_Name"])][0] += float(row["Counter_Value"]); acc[(m.group(0), row["Counter_Name"])][1] += 1
    for k, (v, n) in sorted(acc.items()):
        print(os.path.basename(d), k, "KB total", round(v, 1), "launches", n, "KB per launch", round(v / n, 1), "B per env per launch", round(v * 1024 / n / B, 1))
PY
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; python3 -c "
import json; d=json.load(open('$O/bench_default.json')); print(d['value'], d['roofline']['frac']); [print(a['workload'][:60], '%.3e' % a['value'], round(a.get('roofline',{}).get('frac',0),3)) for a in d.get('also',[])]"
