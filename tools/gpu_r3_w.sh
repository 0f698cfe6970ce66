#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:?}"
cd "$R"
for rep in 1 2; do for v in ${VARIANTS:-P1 P2}; do
  for kb in "od 16384" "oct 32768" "ode 4096"; do set -- $kb
  COOPSEARCH_LIB=$R/build/var/abl_$v.so python tools/oct_sweep.py --n 3 --batches $2 --kernels $1 --reps 8 --tag $v-$1-$2 2>/dev/null
  done
done; done | grep '^{"tag' | python3 -c "
import sys, json
for ln in sys.stdin:
    d = json.loads(ln); print(d['tag'], d['us_per_step'], '%.3e' % d['env_steps_per_s'])
"
COOPSEARCH_LIB=$R/build/var/lib_tl3.so python tools/exp_od_events.py 2>&1 | grep "reset at" | head -5
