#!/bin/bash
# oct against lane around the crossover, bench.py's own protocol (as tools/batch_sweep.py)
set -u
R="${GRAFT_REPO_ROOT:?}"
cd "$R"
for spec in "c2 65536" "c2 98304" "c2 131072" "c3 262144" "c3 524288"; do set -- $spec
  for k in oct lane; do
    python bench.py --no-cpu-baseline --no-also --workload $1 --mode rollout --kernel $k --batch $2 --steps 400 --warmup 100 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('$1', $2, '$k', '%.3e' % d['value'], round(d['ms_per_step'] * 1e3, 2))"
  done
done
