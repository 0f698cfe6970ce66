#!/bin/bash
# Memory-system PMC passes (one group per run) over a python script: tools/pmc_mem.sh <tag> <kernel-substring> <script> [args...]
set -u
R="${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun exports it)}"
TAG="${1:?tag}"; KPAT="${2:?kernel substring}"; SCRIPT="${3:?script}"; shift 3
OUT="$R/gpurun_out/pmcmem_$TAG"
cd /tmp && export TMPDIR=/tmp
rm -rf "$OUT"; mkdir -p "$OUT"
fail=0; i=0
for grp in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum GRBM_GUI_ACTIVE" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum" \
           "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_BUSY_sum TCC_REQ_sum TCC_TAG_STALL_sum" \
           "TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_READ_sum TCC_WRITE_sum TCC_WRITEBACK_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" \
           "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_CACHE_MISS_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp --output-format csv -d "$OUT/g$i" -- python3 "$R/$SCRIPT" "$@" > "$OUT/g$i.log" 2>&1
  rc=$?; echo "group $i ($grp) rc=$rc"; [ $rc -ne 0 ] && { fail=1; tail -3 "$OUT/g$i.log"; }
done
python3 - "$OUT" "$KPAT" <<'PY'
import csv, glob, collections, json, sys
out, pat = sys.argv[1], sys.argv[2]
res = {}
for f in sorted(glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(f)):
        if pat in row["Kernel_Name"]:
            acc[row["Counter_Name"]][0] += float(row["Counter_Value"]); acc[row["Counter_Name"]][1] += 1
    for k, (v, n) in sorted(acc.items()):
        res[k] = round(v / n, 1)
print(json.dumps(res, indent=1))
json.dump(res, open(out + "/summary.json", "w"), indent=1)
PY
exit $fail
