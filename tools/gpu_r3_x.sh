#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:?}"
cd "$R"
for rep in 1 2; do for v in ${VARIANTS}; do
  for kb in ${POINTS}; do k=${kb%%:*}; b=${kb##*:}
  COOPSEARCH_LIB=$R/build/var/abl_$v.so python tools/oct_sweep.py --n ${NAG:-3} --batches $b --kernels $k --reps 8 --tag $v-$k-$b 2>/dev/null
  done
done; done | grep '^{"tag' | python3 -c "
import sys, json
for ln in sys.stdin:
    d = json.loads(ln); print(d['tag'], d['us_per_step'], '%.3e' % d['env_steps_per_s'])
"
