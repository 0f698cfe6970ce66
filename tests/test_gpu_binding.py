"""The two bindings of the same C ABI: torch.ops.coopsearch.* (csrc/torch_ops.cpp, the default of BatchedFlightEnv:
tensor checks in C++, torch's current stream) and ctypes (cooperative-search_amd/_lib.py, torch-free).  Same kernels,
same results; the op layer rejects bad tensors with the messages the boundary promises."""
import numpy as np
import pytest
import torch

import cooperative_search_amd as cs
from cooperative_search_amd import _lib

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("env_name", ["flight_easy", "flight"])
def test_torch_ops_and_ctypes_bindings_agree(env_name):
    B, n, T = 300, 3, 50
    args = cs.make_env_args(env_name, n_agents=n)
    args.time_limit = 30
    seeds = np.arange(B, dtype=np.uint32) + 17
    envs = [cs.BatchedFlightEnv(args, batch=B, seeds=seeds, freeze_done=False, auto_reset=True, binding=b)
            for b in ("torch", "ctypes")]
    assert envs[0]._ops is not None and envs[1]._ops is None
    g = torch.Generator("cuda").manual_seed(9)
    for t in range(T):
        a = torch.randint(0, 3, (B, n), dtype=torch.int64 if t % 2 else torch.int32, device="cuda", generator=g)
        r = [e.step(a) for e in envs]
        for k in range(3):
            assert torch.equal(r[0][k], r[1][k]), (t, k)
        assert torch.equal(envs[0].get_obs(), envs[1].get_obs()) and torch.equal(envs[0].get_state(), envs[1].get_state())
    acts = torch.randint(0, 3, (20, B, n), dtype=torch.int32, device="cuda", generator=g)
    o = [e.rollout(acts) for e in envs]
    for key in ("reward", "terminated", "win", "obs", "state"):
        assert torch.equal(o[0][key], o[1][key]), key
    mask = torch.zeros(B, dtype=torch.uint8, device="cuda")
    mask[::3] = 1
    for e in envs:
        e.reset(init=False, mask=mask)
    assert torch.equal(envs[0].get_state(), envs[1].get_state())
    for key in ("tgt", "agent", "hdr", "mt", "ahead"):
        assert torch.equal(envs[0].raw()[key], envs[1].raw()[key]), key
    assert torch.equal(envs[0].metric_partials(), envs[1].metric_partials())
    assert torch.equal(envs[0].mt_canonical(), envs[1].mt_canonical())


def test_torch_ops_use_the_current_stream():
    """The ops take torch's CURRENT stream: work enqueued under torch.cuda.stream(s) is ordered with that stream."""
    B, n = 256, 3
    env = cs.BatchedFlightEnv(cs.make_env_args("flight_easy", n_agents=n), batch=B, freeze_done=False)
    ref = cs.BatchedFlightEnv(cs.make_env_args("flight_easy", n_agents=n), batch=B, freeze_done=False)
    s = torch.cuda.Stream()
    a = torch.randint(0, 3, (40, B, n), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        for t in range(40):
            env.step(a[t])
        got = env.get_state().clone()
    s.synchronize()
    for t in range(40):
        ref.step(a[t])
    assert torch.equal(got, ref.get_state())


def test_torch_ops_reject_bad_tensors():
    B, n = 64, 3
    env = cs.BatchedFlightEnv(cs.make_env_args("flight_easy", n_agents=n), batch=B)
    ops, cfg, blob = env._ops, env._cfg_t, env._blob
    good = dict(actions=torch.zeros(B, n, dtype=torch.int32, device="cuda"), reward=env._reward, term=env._terminated,
                win=env._win, obs=env._obs, state=env._state)

    def step(**kw):
        d = dict(good, **kw)
        ops.env_step(cfg, blob, d["actions"], 0, d["reward"], d["term"], d["win"], d["obs"], d["state"])

    step()
    with pytest.raises(RuntimeError, match="Act num mismatch agent"):          # flight_env_easy.py:256-257
        step(actions=torch.zeros(B, n + 1, dtype=torch.int32, device="cuda"))
    with pytest.raises(RuntimeError, match="int32 or int64"):
        step(actions=torch.zeros(B, n, dtype=torch.float32, device="cuda"))
    with pytest.raises(RuntimeError, match="must be a GPU tensor"):
        step(reward=torch.zeros(B))
    with pytest.raises(RuntimeError, match="reward must be Float"):
        step(reward=torch.zeros(B, dtype=torch.float64, device="cuda"))
    with pytest.raises(RuntimeError, match="contiguous"):
        step(obs=torch.zeros(B, n, 8, device="cuda")[:, :, ::2])
    with pytest.raises(RuntimeError, match="elements"):
        step(state=torch.zeros(B, 56, device="cuda"))
    with pytest.raises(RuntimeError, match="sizeof"):
        ops.env_step(cfg[:-1].clone(), blob, good["actions"], 0, env._reward, env._terminated, env._win, env._obs, env._state)
    with pytest.raises(RuntimeError, match="at least"):
        ops.env_init(cfg, blob[:1000])
    bad = cfg.clone()
    bad[4:8] = torch.tensor([9, 0, 0, 0], dtype=torch.uint8)                   # n_agents = 9
    with pytest.raises(RuntimeError, match="n_agents"):
        ops.state_bytes(bad)
    assert int(ops.abi_version()) == _lib.ABI_VERSION
