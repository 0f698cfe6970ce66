#!/bin/bash
# rocprofv3 PMC passes (one counter group per run, no tracing flags) over a python script of this repo.
#   usage: tools/pmc.sh <tag> <kernel-substring> <script relative to the repo root> [script args...]
# Writes gpurun_out/pmc_<tag>/summary.json with the per-launch average of every counter for kernels whose name contains
# the substring (PMC_ONLY="2 3" runs only those groups).  FETCH_SIZE / WRITE_SIZE are collected in separate passes (MI355X_MICROARCH.md, PMC slots).
set -u
R="${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun exports it)}"
TAG="${1:?tag}"; KPAT="${2:?kernel substring}"; SCRIPT="${3:?script}"; shift 3
OUT="$R/gpurun_out/pmc_$TAG"
cd /tmp && export TMPDIR=/tmp
rm -rf "$OUT"; mkdir -p "$OUT"
fail=0
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAVES" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INST_CYCLES_VMEM" \
           "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SMEM GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH"; do
  i=$((i+1))
  case " ${PMC_ONLY:-1 2 3 4 5 6 7} " in *" $i "*) ;; *) continue ;; esac
  timeout 600 rocprofv3 --pmc $grp --output-format csv -d "$OUT/g$i" -- python3 "$R/$SCRIPT" "$@" > "$OUT/g$i.log" 2>&1
  rc=$?
  echo "group $i ($grp) rc=$rc"
  [ $rc -ne 0 ] && fail=1
done
python3 - "$OUT" "$KPAT" <<'PY'
import csv, glob, collections, json, sys
out, pat = sys.argv[1], sys.argv[2]
res = {}
launches = 0
for f in sorted(glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(f)):
        if pat in row["Kernel_Name"]:
            acc[row["Counter_Name"]][0] += float(row["Counter_Value"])
            acc[row["Counter_Name"]][1] += 1
            res.setdefault("_kernel", row["Kernel_Name"][:100])
            res["_vgpr"] = int(row["VGPR_Count"]); res["_lds"] = int(row["LDS_Block_Size"]); res["_scratch"] = int(row["Scratch_Size"])
            res["_grid"] = int(row["Grid_Size"])
    for k, (v, n) in sorted(acc.items()):
        res[k] = round(v / n, 1)
        launches = n
res["_launches_averaged"] = launches
print(json.dumps(res, indent=1))
json.dump(res, open(out + "/summary.json", "w"), indent=1)
PY
exit $fail
