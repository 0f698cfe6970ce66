"""Does the 5-agent lane kernel's rate at 2^18 envs depend on WHEN in a process its output tables were allocated?
(fresh process -> tables A; then 40 GB allocated and released; tables B allocated afterwards; A and B timed alternately)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cooperative_search_amd as cs
n, B, S = 5, 262144, 100
dev = torch.device("cuda", 0)
env = cs.BatchedFlightEnv(cs.make_env_args("flight_easy", n_agents=n), batch=B, device=dev, freeze_done=False, auto_reset=True, kernel="lanev")
acts = torch.randint(0, 3, (S, B, n), dtype=torch.int32, device=dev)

def tables():
    return dict(reward=torch.empty(S, B, dtype=torch.float32, device=dev), terminated=torch.empty(S, B, dtype=torch.uint8, device=dev),
                win=torch.empty(S, B, dtype=torch.uint8, device=dev), obs=torch.empty(S, B, n, 4, dtype=torch.float32, device=dev),
                state=torch.empty(S, B, env.state_shape, dtype=torch.float32, device=dev))

def rate(out, tag):
    for _ in range(2): env.rollout(acts, out=out, update_views=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): env.rollout(acts, out=out, update_views=False)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (5 * S)
    print(f"{tag}: {us:.2f} us per step ({(36 * n + 186) * B / us * 1e6 / 8e12 * 100:.1f} %)", flush=True)

A = tables()
rate(A, "tables A (fresh process)")
big = [torch.empty(10 << 30, dtype=torch.uint8, device=dev) for _ in range(4)]
for b in big: b.fill_(1)
torch.cuda.synchronize()
del big
torch.cuda.empty_cache()
Bt = tables()
rate(Bt, "tables B (after 40 GB came and went)")
rate(A, "tables A again")
rate(Bt, "tables B again")
del A
torch.cuda.empty_cache()
C = tables()
rate(C, "tables C (allocated into A's released space)")
