"""Child of tests/test_gpu_jitter.py: runs with COOPSEARCH_LIB pointing at one of its one-team-size builds (-DCS_JITTER,
-DCS_OD_SAFE_WAIT, -DCS_OD_ASYNC=0 for teams of 3; -DCS_JITTER and the wide-band pre-filter build for teams of 5).  The octet pair
kernels (K + D, and K + D + E) -- their hand-shakes stretched by pseudo-random pauses -- against the 16-lane step kernel of the
same library, bit for bit, on a scenario where nearly every episode ends with a win K could not predict (fix request, restore
from the ring, redo, acknowledge: every other step) and on the shipped configuration."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cooperative_search_amd as cs  # noqa: E402


N_AGENTS = 5 if os.path.basename(os.environ.get("COOPSEARCH_LIB", "")).endswith("_n5.so") else 3


def custom(**kw):
    a = cs.make_env_args("flight_easy", n_agents=N_AGENTS)
    for k, v in kw.items():
        setattr(a, k, v)
    return a


def run(kernel, args, B, lengths, mode):
    n = args.n_agents
    seeds = np.arange(B, dtype=np.uint32) + 5150
    g = torch.Generator("cuda").manual_seed(17)
    e1 = cs.BatchedFlightEnv(args, batch=B, seeds=seeds, kernel="group", **mode)
    e2 = cs.BatchedFlightEnv(args, batch=B, seeds=seeds, kernel=kernel, **mode)
    wins = 0
    for L in lengths:
        acts = torch.randint(0, 3, (L, B, n), dtype=torch.int32, device="cuda", generator=g)
        out = e2.rollout(acts)
        for k in range(L):
            r, term, win = e1.step(acts[k])
            assert torch.equal(r, out["reward"][k]) and torch.equal(term, out["terminated"][k]) and torch.equal(win, out["win"][k]), (kernel, k)
            if k % 7 == 0 or k == L - 1:
                assert torch.equal(e1.get_obs(), out["obs"][k]) and torch.equal(e1.get_state(), out["state"][k]), (kernel, k)
        wins += int(out["win"].sum().item())
        r1, r2 = e1.raw(), e2.raw()
        for key in ("tgt", "agent", "hdr"):
            assert torch.equal(r1[key], r2[key]), (kernel, key)
        assert torch.equal(e1.mt_canonical(), e2.mt_canonical()), kernel
    return wins


def main():
    name = os.path.basename(cs.lib.library_path())
    assert name in ("jitter_n3.so", "odsafe_n3.so", "odsync_n3.so", "jitter_n5.so", "prewide_n5.so") or os.environ.get("CS_CHILD_ANY_LIB") == "1", cs.lib.library_path()
    kernels = ("od", "ode", "oct") if name == "prewide_n5.so" else ("od", "ode")
    for kernel in kernels:
        if name == "jitter_n5.so":   # a batch with a tail of 4 envs (the scalar-store variant beside the vector one)
            run(kernel, custom(), 1004, (30, 7), dict(freeze_done=False, auto_reset=True))
        w = run(kernel, custom(target_num=2, target_mode=1, detect_prob=1.0, view_range=25), 1000, (7, 64, 3, 100, 26),
                dict(freeze_done=False, auto_reset=True))
        assert w > 5000, w   # tens of unpredicted wins per env
        run(kernel, custom(target_num=2, target_mode=1, detect_prob=1.0, view_range=25), 520, (40, 9), dict(freeze_done=True))
        run(kernel, custom(), 4096, (100, 20), dict(freeze_done=False, auto_reset=True))
        # every target within view of every agent most of the time: up to 90 stream words per env-step, a row refresh every few
        # steps (the three-wavefront variant hands those to E: request, old tape meanwhile, adoption -- and the waits for an
        # outstanding refresh in front of resets and on-the-spot top-ups), short and long launches
        run(kernel, custom(view_range=70, detect_prob=0.05), 520, (50, 5, 33, 1, 2, 64), dict(freeze_done=False, auto_reset=True))
        print(f"{name}, {kernel}: bit-identical to the step kernel ({w} unpredicted wins handled)", flush=True)


if __name__ == "__main__":
    main()
