#!/bin/bash
# HBM traffic of the bench workloads: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (they do not fit one
# pass; MI355X_MICROARCH.md, PMC slots) over tools/exp_workload.py, which runs a known number of env-steps.
#   usage (on the GPU box): bash tools/pmc_traffic.sh      -> gpurun_out/traffic/<tag>_{fetch,write}/ + traffic_raw.json
set -u
R="${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun exports it)}"
OUT="$R/gpurun_out/traffic"
cd /tmp && export TMPDIR=/tmp
rm -rf "$OUT"; mkdir -p "$OUT"
fail=0
run() {  # tag, args...
  tag=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --output-format csv -d "$OUT/${tag}_$c" -- python3 "$R/tools/exp_workload.py" "$@" > "$OUT/${tag}_$c.log" 2>&1
    rc=$?; echo "$tag $c rc=$rc"; [ $rc -ne 0 ] && { fail=1; tail -3 "$OUT/${tag}_$c.log"; }
  done
}
run c2_rollout flight_easy 3 auto 4096 rollout 4 100
run c2_rollout20 flight_easy 3 auto 4096 rollout 8 20
run c2_step flight_easy 3 group 4096 step 2 100
run c3_rollout flight_easy 5 auto 16384 rollout 4 100
run c5_rollout flight_easy 5 auto 8192 rollout 4 100
run od3_16384 flight_easy 3 auto 16384 rollout 4 100
run oct3_32768 flight_easy 3 auto 32768 rollout 4 100
run oct5_32768 flight_easy 5 auto 32768 rollout 4 100
run c5s_rollout flight_easy 5 auto 65536 rollout 3 100
run lane3_rollout flight_easy 3 auto 262144 rollout 3 100
run lane5_rollout flight_easy 5 auto 262144 rollout 3 100
run lane3_1m flight_easy 3 auto 1048576 rollout 2 100
run lane5_1m flight_easy 5 auto 1048576 rollout 2 100
run c4_step flight 3 group 8192 step 1 100
run c4_rollout flight 3 group 8192 rollout 1 100
python3 - "$OUT" <<'PY'
import csv, glob, json, os, re, sys, collections
out = sys.argv[1]
res = {}
for d in sorted(glob.glob(out + "/*_FETCH_SIZE")):
    tag = os.path.basename(d)[:-len("_FETCH_SIZE")]
    env_steps = int(re.search(r"env_steps (\d+)", open(out + f"/{tag}_FETCH_SIZE.log").read()).group(1))
    ent = {"env_steps": env_steps, "kernels": {}}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for f in glob.glob(out + f"/{tag}_{c}/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                if row["Counter_Name"] != c:
                    continue
                name = row["Kernel_Name"]
                m = re.search(r"(k_[a-z_0-9]+)(<[^>(]*>)?", name)
                if not m:
                    continue
                key = m.group(1) + (m.group(2) or "")
                acc[key][0] += float(row["Counter_Value"]); acc[key][1] += 1
        for k, (v, n) in acc.items():
            ent["kernels"].setdefault(k, {})[c + "_KB_total"] = round(v, 1)
            ent["kernels"][k]["launches"] = n
    res[tag] = ent
json.dump(res, open(out + "/traffic_raw.json", "w"), indent=1)
print(json.dumps(res, indent=1)[:6000])
PY
exit $fail
