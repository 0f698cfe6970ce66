#!/bin/bash
# round 4, call K: timeline of the fused closed-loop kernel; run-to-run spread of the 5-agent lane kernel in fresh processes;
# what sysfs offers for the rank -> CPU binding; the reworked cpu_baseline
set -u
R="${GRAFT_REPO_ROOT:?}"
cd "$R"
mkdir -p gpurun_out
echo "== timeline 4096"; COOPSEARCH_LIB=$R/build/var/tlp_n3.so timeout 300 python tools/exp_policy_loop_timeline.py 2>&1 | grep -v amdgpu.ids
echo "== timeline 65536"; B=65536 COOPSEARCH_LIB=$R/build/var/tlp_n3.so timeout 300 python tools/exp_policy_loop_timeline.py 2>&1 | grep -v amdgpu.ids
echo "== sysfs"
for n in /sys/class/kfd/kfd/topology/nodes/*; do
  if grep -q "simd_count [1-9]" $n/properties 2>/dev/null; then
    echo "gpu node $n"; grep -E "location_id|domain|drm_render_minor|unique_id" $n/properties
    for l in $n/io_links/*; do echo " link $l: $(tr '\n' ' ' < $l/properties 2>/dev/null | cut -c1-200)"; done
  fi
done
ls /sys/class/drm/ | tr '\n' ' '; echo
ls -d /sys/bus/pci/devices/*/drm/renderD* 2>/dev/null | head -20
for d in /sys/bus/pci/devices/*; do if [ -d $d/drm ]; then echo "$d numa=$(cat $d/numa_node 2>/dev/null) cpus=$(cat $d/local_cpulist 2>/dev/null) $(ls $d/drm | tr '\n' ' ')"; fi; done
ls /sys/devices/system/node/ | tr '\n' ' '; echo; cat /sys/devices/system/node/node*/cpulist
cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null
python - <<'PY'
import sys; sys.argv=["x"]
import importlib.util, os
spec = importlib.util.spec_from_file_location("bench_mod", "bench.py"); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
print("gpu_local_cpus(0):", (lambda f: (f[0], len(f[1])) if f else None)(b.gpu_local_cpus(0)))
PY
echo "== cpu baseline, twice"
for i in 1 2; do OMP_PROC_BIND=close OMP_PLACES=cores OMP_DYNAMIC=false timeout 300 python bench.py --cpu-baseline-child flight_easy,3,4096,10 | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['cores'], d['p90_over_p10'], d['steady'], d['measured_points'], d['thread_scaling_env_steps_per_s'])"; done
echo "== 5 agents 2^18, fresh processes"
for i in 1 2 3 4 5; do COOPSEARCH_LIB=$R/build/var/cur_n5.so timeout 300 python bench.py --workload c3 --mode rollout --kernel auto --batch 262144 --steps 400 --warmup 100 --no-cpu-baseline --no-also 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['roofline']['frac'], d.get('region_ms_min_median_max'))"; done
