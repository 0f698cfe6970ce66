#!/bin/bash
# round 4, call A: this box's baseline, occupancy sensitivity of the lane kernel (1 vs 2 wavefronts per SIMD), timeline + SQ counters of the 5-agent kernels
set -u
R="${GRAFT_REPO_ROOT:?}"
cd "$R"
mkdir -p gpurun_out
python bench.py --steps 20 --warmup 5 > gpurun_out/a_bench.json 2> gpurun_out/a_bench.err; echo "bench rc=$?"; python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/a_bench.json") if l.startswith("{")][-1])
print("headline", d["value"], d["roofline"]["frac"], d.get("cpu_baseline", {}).get("value"))
for a in d.get("also", []):
    print(" ", a.get("workload") or a.get("config"), a.get("value"), (a.get("roofline") or {}).get("frac"))
PY
for v in base_n3 occ1_n3; do COOPSEARCH_LIB=$R/build/var/$v.so python tools/quick_lane.py 3 lane 262144 1048576 2>&1 | grep -v amdgpu.ids | sed "s/^/$v /"; done
for v in base_n5 occ1_n5; do COOPSEARCH_LIB=$R/build/var/$v.so python tools/quick_lane.py 5 lane 262144 1048576 2>&1 | grep -v amdgpu.ids | sed "s/^/$v /"; done
COOPSEARCH_LIB=$R/build/var/base_n5.so python tools/quick_lane.py 5 oct 262144 2>&1 | grep -v amdgpu.ids
COOPSEARCH_LIB=$R/build/var/base_n5.so python tools/quick_lane.py 5 od 8192 16384 2>&1 | grep -v amdgpu.ids
COOPSEARCH_LIB=$R/build/var/tl_n5.so N=5 B=262144 python tools/exp_lane_timeline.py 2>&1 | grep -v amdgpu.ids
bash tools/pmc.sh lane5 k_rollout_lane tools/exp_workload.py flight_easy 5 lane 262144 rollout 3 100 > gpurun_out/pmc_lane5.log 2>&1; echo "pmc lane5 rc=$?"; tail -32 gpurun_out/pmc_lane5.log
bash tools/pmc.sh oct5 k_rollout_oct tools/exp_workload.py flight_easy 5 oct 262144 rollout 3 100 > gpurun_out/pmc_oct5.log 2>&1; echo "pmc oct5 rc=$?"; tail -32 gpurun_out/pmc_oct5.log
# topology of the box (for the rank -> NUMA binding of bench.py): KFD nodes, render minors, NUMA nodes, CPUs
{ for n in /sys/class/kfd/kfd/topology/nodes/*; do echo "== $n"; grep -E "simd_count|drm_render_minor|location_id|domain|cpu_cores_count" $n/properties; done
  for d in /sys/class/drm/renderD*; do echo "$d numa=$(cat $d/device/numa_node 2>/dev/null) cpus=$(cat $d/device/local_cpulist 2>/dev/null)"; done
  lscpu | grep -E "Model name|Socket|Core|Thread|NUMA|^CPU\(s\)"; python -c "import os; print('affinity', len(os.sched_getaffinity(0)))"; } > gpurun_out/a_topology.txt 2>&1
tail -30 gpurun_out/a_topology.txt
