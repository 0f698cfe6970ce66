"""flight closed loop at B envs: stepwise (4 kernels per step with observation copies) vs cs_rollout_policy_flight
(emit False / True); rocprofv3 --kernel-trace shows the per-kernel split.
    python tools/exp_flight_loop.py [B=8192] [steps=200]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import cooperative_search_amd as cs
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
K = int(sys.argv[2]) if len(sys.argv) > 2 else 200
args = cs.make_env_args("flight", n_agents=3)
env = cs.BatchedFlightEnv(args, batch=B, freeze_done=False, auto_reset=True)
cs.apply_env_info(args, env)
torch.manual_seed(0)
agents = cs.FusedAgents(args, B)


def timed(fn, steps):
    fn(100)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn(steps)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / steps


def stepwise(steps):
    for _ in range(steps):
        env.step(agents.choose_action(env.get_obs()))


for emit in (False, True):
    out = env.rollout_policy(agents, 100, emit=emit)

    def call(steps, emit=emit, out=out):
        for _ in range(steps // 100):
            env.rollout_policy(agents, 100, emit=emit, out=out, update_views=False)
    us = timed(call, K)
    print(f"cs_rollout_policy_flight emit={emit}: {us:.1f} us/step  {B / us * 1e6:.3e} env-steps/s")
    del out
us = timed(stepwise, K)
print(f"stepwise choose_action + step: {us:.1f} us/step  {B / us * 1e6:.3e} env-steps/s")
