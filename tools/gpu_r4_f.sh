#!/bin/bash
# round 4, call F: the whole GPU suite on the rebuilt library; A/B of the octet kinematics changes at c2 (driver arguments and 100 steps
# per launch); k_rollout_lanev after the refresh request moved ahead of the kinematics
set -u
R="${GRAFT_REPO_ROOT:?}"
cd "$R"
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --durations=25 > gpurun_out/f_gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -45 gpurun_out/f_gpu_tests.log
for rep in 1 2; do for v in ts0_n3 ts1_n3 ts2_n3; do
  for st in "20 5" "2000 200"; do set -- $st
    COOPSEARCH_LIB=$R/build/var/$v.so python bench.py --steps $1 --warmup $2 --no-also --no-cpu-baseline --pg off 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$v steps $1:', '%.3e' % d['value'], 'frac %.4f' % d['roofline']['frac'], 'region ms', ['%.4f' % x for x in d['timing']['region_ms_min_median_max']])"
  done
done; done
for v in ts0_n3 ts2_n3; do COOPSEARCH_LIB=$R/build/var/$v.so python tools/quick_lane.py 3 od 8192 16384 2>&1 | grep -v amdgpu.ids | sed "s/^/$v /"; COOPSEARCH_LIB=$R/build/var/$v.so python tools/quick_lane.py 3 oct 32768 2>&1 | grep -v amdgpu.ids | sed "s/^/$v /"; done
for n in 3 5; do COOPSEARCH_LIB=$R/build/var/ts2_n$n.so python tools/quick_lane.py $n lanev 65536 262144 1048576 2>&1 | grep -v amdgpu.ids; done
COOPSEARCH_LIB=$R/build/var/ts2_n5.so python tools/quick_lane.py 5 od 8192 16384 2>&1 | grep -v amdgpu.ids
