#!/bin/bash
# PMC passes over the fused policy kernel (tools/exp_policy.py); one counter group per run.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc
rocprofv3 --list-avail 2>/dev/null | grep -oE "\bSQ_[A-Z_0-9]+" | sort -u > $R/gpurun_out/pmc/sq_counters.txt
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_WAIT_ANY"; do
  i=$((i+1))
  CS_POLICY_M=${CS_POLICY_M:-1} timeout 200 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmc/g$i -- python3 $R/tools/exp_policy.py > $R/gpurun_out/pmc/g$i.log 2>&1
  echo "group $i rc=$?"
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$R/gpurun_out/pmc/g*/")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for row in csv.DictReader(open(f)):
            if "k_policy" in row["Kernel_Name"]:
                key = (row["Counter_Name"], row["Grid_Size"])
                acc[key][0] += float(row["Counter_Value"]); acc[key][1] += 1
        for k, (v, n) in sorted(acc.items()):
            print(k[0], "grid", k[1], "avg/launch", round(v / n, 1), "launches", n)
PY
