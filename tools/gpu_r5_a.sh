#!/bin/bash
# round 5, call A: the lane kernels with the observation block staged through LDS (1a): GPU tests, rates in fresh processes,
# FETCH / WRITE traffic, and the list of counters this box's rocprofv3 offers.
set -u
R="${GRAFT_REPO_ROOT:?}"
cd "$R"
O=gpurun_out/r5a; rm -rf $O; mkdir -p $O
python -m pytest tests -m gpu -q -x --durations=5 > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/gpu_tests.log
for i in 1 2 3 4 5; do python tools/quick_lane.py 5 lanev 262144 2>&1 | grep "n=5"; done | tee $O/lanev5_fresh.txt
python tools/quick_lane.py 5 lanev 65536 1048576 2>&1 | grep "n=" | tee -a $O/lanev5_fresh.txt
python tools/quick_lane.py 3 lanev 65536 262144 1048576 2>&1 | grep "n=" | tee $O/lanev3.txt
python tools/quick_lane.py 4 lanev 262144 2>&1 | grep "n=" | tee -a $O/lanev3.txt
python tools/quick_lane.py 6 lane 262144 1048576 2>&1 | grep "n=" | tee $O/lane6.txt
python tools/quick_lane.py 5 lane 262144 2>&1 | grep "n=" | tee -a $O/lane6.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $R/$O/avail_counters.txt 2>&1
for tag in "5 lanev 262144" "3 lanev 262144" "6 lane 262144"; do
  set -- $tag
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --output-format csv -d $R/$O/pmc_$1_$2_$c -- python3 $R/tools/exp_workload.py flight_easy $1 $2 $3 rollout 3 100 > $R/$O/pmc_$1_$2_$c.log 2>&1
    echo "pmc $tag $c rc=$?"
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
        print(os.path.basename(d), k, "KB total", round(v, 1), "launches", n)
PY
