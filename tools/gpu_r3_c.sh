#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:?}"
cd "$R"
mkdir -p gpurun_out
python -m pytest tests/test_gpu_rccl.py -x -q > gpurun_out/c_rccl.log 2>&1; echo "rccl rc=$?"; tail -3 gpurun_out/c_rccl.log
for f in build/var/abl3_*.so; do
  for emit in 1 0; do
    COOPSEARCH_LIB=$R/$f EMIT=$emit python - <<'PY'
import os, sys, json
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
import cooperative_search_amd as cs
n, T = 3, 100
emit = os.environ["EMIT"] == "1"
res = []
for B in (4096, 8192, 16384):
    env = cs.BatchedFlightEnv(cs.make_env_args("flight_easy", n_agents=n), batch=B, freeze_done=False, auto_reset=True, kernel="oct")
    acts = torch.randint(0, 3, (T, B, n), dtype=torch.int32, device="cuda")
    out = env.rollout(acts, emit=emit, update_views=False)
    for _ in range(2): env.rollout(acts, out=out, update_views=False)
    torch.cuda.synchronize()
    ts = []
    for _ in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); env.rollout(acts, out=out, update_views=False); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / T)
    ts.sort(); res.append(round(ts[3], 3))
print(os.path.basename(os.environ["COOPSEARCH_LIB"]), "emit" if emit else "noemit", res, flush=True)
PY
  done
done 2>&1 | grep -v amdgpu.ids
