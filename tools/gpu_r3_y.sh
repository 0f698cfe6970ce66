#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:?}"
cd "$R"
for rep in 1 2; do for v in ${VARIANTS}; do
  COOPSEARCH_LIB=$R/build/var/abl_$v.so python tools/oct_sweep.py --n 3 --batches 4096 --kernels ode --reps 10 --tag $v-T100 2>/dev/null
  COOPSEARCH_LIB=$R/build/var/abl_$v.so python tools/oct_sweep.py --n 3 --batches 4096 --kernels ode --T 20 --reps 60 --tag $v-T20 2>/dev/null
  COOPSEARCH_LIB=$R/build/var/abl_$v.so python tools/oct_sweep.py --n 3 --batches 8192 --kernels ode --reps 10 --tag $v-8192 2>/dev/null
done; done | grep '^{"tag' | python3 -c "
import sys, json
for ln in sys.stdin:
    d = json.loads(ln); print(d['tag'], d['us_per_step'], '%.3e' % d['env_steps_per_s'])
"
