#!/bin/bash
# PMC passes over k_policy at one size (tools/exp_policy.py N B); one counter group per run.  Usage: pmc_policy.sh 3 65536
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
N=${1:-3}; B=${2:-65536}
rm -rf $R/gpurun_out/pmc; mkdir -p $R/gpurun_out/pmc
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" \
           "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_WAIT_ANY"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmc/g$i -- python3 $R/tools/exp_policy.py $N $B > $R/gpurun_out/pmc/g$i.log 2>&1
  echo "group $i rc=$?"
done
python3 - <<PY
import csv, glob, collections, json
res = {}
for d in sorted(glob.glob("$R/gpurun_out/pmc/g*/")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for row in csv.DictReader(open(f)):
            if "k_policy" in row["Kernel_Name"]:
                acc[row["Counter_Name"]][0] += float(row["Counter_Value"]); acc[row["Counter_Name"]][1] += 1
        for k, (v, n) in sorted(acc.items()):
            res[k] = round(v / n, 1)
print(json.dumps(res))
json.dump(res, open("$R/gpurun_out/pmc/policy_pmc_${N}_${B}.json", "w"), indent=1)
PY
