#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:?}"
cd "$R"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k "octet or stepwise or (long_horizon and oct) or full_size_rollout" > gpurun_out/d_oct.log 2>&1; echo "oct rc=$?"; tail -4 gpurun_out/d_oct.log
for cfg in "3 8192" "5 8192" "5 16384"; do set -- $cfg; N=$1 B=$2 COOPSEARCH_LIB=$R/build/var/lib_tl$1.so python tools/exp_oct_timeline.py 2>&1 | grep -v amdgpu.ids | tail -10; done
for n in 3 5; do
for f in build/var/abl${n}_*.so; do
    COOPSEARCH_LIB=$R/$f NN=$n python - <<'PY'
import os, sys, json
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
import cooperative_search_amd as cs
n, T = int(os.environ["NN"]), 100
res = []
for B in (4096, 8192, 16384):
    env = cs.BatchedFlightEnv(cs.make_env_args("flight_easy", n_agents=n), batch=B, freeze_done=False, auto_reset=True, kernel="oct")
    acts = torch.randint(0, 3, (T, B, n), dtype=torch.int32, device="cuda")
    out = env.rollout(acts, update_views=False)
    for _ in range(2): env.rollout(acts, out=out, update_views=False)
    torch.cuda.synchronize()
    ts = []
    for _ in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); env.rollout(acts, out=out, update_views=False); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / T)
    ts.sort(); res.append(round(ts[3], 3))
print(os.path.basename(os.environ["COOPSEARCH_LIB"]), res, flush=True)
PY
done; done 2>&1 | grep -v amdgpu.ids
python tools/oct_sweep.py --n 3,5 --batches 4096,8192,16384,32768,65536,262144 --kernels oct --tag w4 > gpurun_out/d_sweep_w4.jsonl 2> gpurun_out/d_sweep_w4.err; echo "sweep w4 rc=$?"
for w in 2 3; do
  COOPSEARCH_LIB=$R/build/var/lib_w$w.so python tools/oct_sweep.py --n 3,5 --batches 4096,8192,16384,32768,65536,262144 --kernels oct --tag w$w > gpurun_out/d_sweep_w$w.jsonl 2> gpurun_out/d_sweep_w$w.err; echo "sweep w$w rc=$?"
done
cat gpurun_out/d_sweep_w4.jsonl gpurun_out/d_sweep_w2.jsonl gpurun_out/d_sweep_w3.jsonl | python -c "
import sys, json
for ln in sys.stdin:
    d = json.loads(ln); print(d['tag'][-12:], d['n'], d['B'], d['kernel'], d['us_per_step'], '%.3e' % d['env_steps_per_s'], d['hbm_frac'])
"
