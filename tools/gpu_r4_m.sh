#!/bin/bash
# round 4, call M: k_rollout_lanev with 128-thread workgroups (finer backfill) against 256; the closed loop with the gate-major GRU;
# cpu baseline under the cgroup quota; the binding with the box's device filters
set -u
R="${GRAFT_REPO_ROOT:?}"
cd "$R"
mkdir -p gpurun_out
for rep in 1 2; do
for v in cur_n5 lvb128_n5; do echo "== $v"; N=5 COOPSEARCH_LIB=$R/build/var/$v.so timeout 600 python tools/exp_lanev_check.py 65536 262144 1048576 2>&1 | grep -v "amdgpu.ids" | grep -v " lane B=" ; done
for v in cur_n3 lvb128_n3; do echo "== $v"; N=3 COOPSEARCH_LIB=$R/build/var/$v.so timeout 600 python tools/exp_lanev_check.py 65536 262144 1048576 2>&1 | grep -v "amdgpu.ids" | grep -v " lane B="; done
done
for v in pol4_n3 pol5_n3; do echo "== $v"; COOPSEARCH_LIB=$R/build/var/$v.so timeout 600 python tools/exp_closed_loop.py easy 2>&1 | grep -v amdgpu.ids; done
python - <<'PY'
import sys; sys.argv=["x"]
import importlib.util, os
spec = importlib.util.spec_from_file_location("bench_mod", "bench.py"); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
print("gpu_local_cpus(0):", (lambda f: (f[0], len(f[1])) if f else None)(b.gpu_local_cpus(0)), "quota", b.cpu_quota())
PY
echo "== cpu baseline, twice"
for i in 1 2; do OMP_PROC_BIND=close OMP_PLACES=cores OMP_DYNAMIC=false timeout 300 python bench.py --cpu-baseline-child flight_easy,3,4096,10 | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['value'], d['cores'], d['p90_over_p10'], d['steady'], d['measured_points'], d['thread_scaling_env_steps_per_s'])"; done
