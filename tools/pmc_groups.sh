#!/bin/bash
# rocprofv3 PMC passes with caller-chosen counter groups (one group per run, no tracing flags) over a python script of this repo.
#   usage: tools/pmc_groups.sh <tag> <kernel-substring> "<group 1>;<group 2>;..." <script relative to the repo root> [script args...]
# LIB=<path> runs the script on another build of the library (COOPSEARCH_LIB).
set -u
R="${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (gpurun exports it)}"
TAG="${1:?tag}"; KPAT="${2:?kernel substring}"; GROUPS_="${3:?groups}"; SCRIPT="${4:?script}"; shift 4
OUT="$R/gpurun_out/pmc_$TAG"
cd /tmp && export TMPDIR=/tmp
rm -rf "$OUT"; mkdir -p "$OUT"
[ -n "${LIB:-}" ] && export COOPSEARCH_LIB="$LIB"
i=0
IFS=';' read -ra GS <<< "$GROUPS_"
for grp in "${GS[@]}"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp --output-format csv -d "$OUT/g$i" -- python3 "$R/$SCRIPT" "$@" > "$OUT/g$i.log" 2>&1
  echo "group $i ($grp) rc=$?"
done
python3 - "$OUT" "$KPAT" <<'PY'
import csv, glob, collections, json, sys
out, pat = sys.argv[1], sys.argv[2]
res = {}
for f in sorted(glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(f)):
        if pat in row["Kernel_Name"]:
            acc[row["Counter_Name"]][0] += float(row["Counter_Value"])
            acc[row["Counter_Name"]][1] += 1
            res.setdefault("_kernel", row["Kernel_Name"][:100])
            res["_vgpr"] = int(row["VGPR_Count"]); res["_grid"] = int(row["Grid_Size"])
    for k, (v, n) in sorted(acc.items()):
        res[k] = round(v / n, 1)
        res["_launches_averaged"] = n
print(json.dumps(res, indent=1))
json.dump(res, open(out + "/summary.json", "w"), indent=1)
PY
