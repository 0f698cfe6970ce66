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


def test_caller_side_ops_agree_with_ctypes_and_check_their_tensors(monkeypatch):
    """policy_forward / rollout_policy / store_episodes through torch.ops.coopsearch against the ctypes route."""
    from cooperative_search_amd.agents import FusedAgents
    from cooperative_search_amd import collector
    n, B, T = 3, 128, 30
    args = cs.make_env_args("flight_easy", n_agents=n)
    env_t = cs.BatchedFlightEnv(args, batch=B, freeze_done=True)
    cs.apply_env_info(args, env_t)
    torch.manual_seed(2)
    ag_t = FusedAgents(args, B, seed=9)
    assert ag_t._ops is not None
    out_t = env_t.rollout_policy(ag_t, T, epsilon=0.2, evaluate=False)
    ep_t = collector.assemble_episodes(torch.cat([out_t["obs"][:1] * 0, out_t["obs"]]), torch.cat([out_t["state"][:1] * 0, out_t["state"]]),
                                       out_t["actions"], out_t["reward"], out_t["terminated"], 3)
    # the same through ctypes
    monkeypatch.setenv("COOPSEARCH_LIB", _lib.library_path())
    env_c = cs.BatchedFlightEnv(args, batch=B, freeze_done=True)
    ag_c = FusedAgents(args, B, net=ag_t.net, seed=9)
    assert env_c._ops is None and ag_c._ops is None
    out_c = env_c.rollout_policy(ag_c, T, epsilon=0.2, evaluate=False)
    ep_c = collector.assemble_episodes(torch.cat([out_c["obs"][:1] * 0, out_c["obs"]]), torch.cat([out_c["state"][:1] * 0, out_c["state"]]),
                                       out_c["actions"], out_c["reward"], out_c["terminated"], 3)
    monkeypatch.delenv("COOPSEARCH_LIB")
    for k in ("actions", "reward", "terminated", "obs", "state"):
        assert torch.equal(out_t[k], out_c[k]), k
    for k in ep_t:
        assert torch.equal(ep_t[k], ep_c[k]), k
    a_t = ag_t.choose_action(env_t.get_obs(), epsilon=0.1, want_q=True).clone()
    a_c = ag_c.choose_action(env_c.get_obs(), epsilon=0.1, want_q=True).clone()
    assert torch.equal(a_t, a_c) and torch.equal(ag_t.q, ag_c.q) and torch.equal(ag_t.hidden, ag_c.hidden)
    ops = _lib.torch_ops()
    with pytest.raises(RuntimeError, match="hidden"):
        ops.policy_forward(ag_t.packed, env_t.get_obs(), 4, 0, ag_t.actions, None, n, ag_t.hidden[:-1].contiguous(), None, ag_t.actions,
                           B * n, n, 3, 0.0, None, 0, 0, 0, 0)
    with pytest.raises(RuntimeError, match="actions"):
        ops.policy_forward(ag_t.packed, env_t.get_obs(), 4, 0, ag_t.actions, None, n, ag_t.hidden, None, ag_t.actions.int(),
                           B * n, n, 3, 0.0, None, 0, 0, 0, 0)
    with pytest.raises(RuntimeError, match="eps_env must be"):   # the per-env exploration schedule: float64 [B]
        ops.policy_forward(ag_t.packed, env_t.get_obs(), 4, 0, ag_t.actions, None, n, ag_t.hidden, None, ag_t.actions,
                           B * n, n, 3, 0.0, torch.zeros(B, device="cuda"), 0, 0, 0, 0)
    with pytest.raises(RuntimeError, match="eps_trace needs eps_env"):
        ops.rollout_policy(env_t._cfg_t, env_t._blob, ag_t.packed, ag_t.hidden, ag_t.actions, T, 0, 0.0, None, 0.0, 0.0, True,
                           torch.zeros(T, B, dtype=torch.float64, device="cuda"), 0, 0, 0, 0, out_t["actions"], out_t["reward"],
                           out_t["terminated"].view(torch.uint8), out_t["win"].view(torch.uint8), None, None)
    with pytest.raises(RuntimeError, match="11 destination"):
        ops.store_episodes(out_t["obs"], out_t["state"], out_t["actions"][:-1].contiguous(), out_t["reward"][:-1].contiguous(),
                           out_t["terminated"][:-1].contiguous().view(torch.uint8), None, 3, [])


@pytest.mark.parametrize("emit", [True, False])
def test_flight_closed_loop_call_agrees_between_bindings(monkeypatch, emit):
    """cs_rollout_policy_flight through torch.ops.coopsearch.rollout_policy_flight and through ctypes; the op layer
    refuses a scratch tensor of the wrong size."""
    from cooperative_search_amd.agents import FusedAgents
    n, B, T = 3, 48, 12
    args = cs.make_env_args("flight", n_agents=n)
    args.time_limit = 7
    env_t = cs.BatchedFlightEnv(args, batch=B, freeze_done=False, auto_reset=True)
    cs.apply_env_info(args, env_t)
    torch.manual_seed(4)
    ag_t = FusedAgents(args, B, seed=3)
    assert ag_t._ops is not None
    out_t = env_t.rollout_policy(ag_t, T, epsilon=0.2, evaluate=False, emit=emit)
    monkeypatch.setenv("COOPSEARCH_LIB", _lib.library_path())
    env_c = cs.BatchedFlightEnv(args, batch=B, freeze_done=False, auto_reset=True)
    ag_c = FusedAgents(args, B, net=ag_t.net, seed=3)
    assert env_c._ops is None and ag_c._ops is None
    out_c = env_c.rollout_policy(ag_c, T, epsilon=0.2, evaluate=False, emit=emit)
    monkeypatch.delenv("COOPSEARCH_LIB")
    for k in ("actions", "reward", "terminated", "win") + (("obs", "state") if emit else ()):
        assert torch.equal(out_t[k], out_c[k]), k
    assert torch.equal(ag_t.hidden, ag_c.hidden) and torch.equal(env_t.raw()["prob"], env_c.raw()["prob"])
    assert torch.equal(env_t.get_obs(), env_c.get_obs())
    ops = _lib.torch_ops()
    with pytest.raises(RuntimeError, match="scratch must"):
        ops.rollout_policy_flight(env_t._cfg_t, env_t._blob, ag_t.packed, *ag_t.conv_w, ag_t.hidden, ag_t.actions,
                                  torch.empty(B, 16, device="cuda"), T, 0, 0.0, None, 0.0, 0.0, False, None, 0, 0, 0, 0, out_t["actions"], out_t["reward"],
                                  out_t["terminated"].view(torch.uint8), out_t["win"].view(torch.uint8), None, None)
    easy = cs.BatchedFlightEnv(cs.make_env_args("flight_easy", n_agents=n), batch=B)
    with pytest.raises(RuntimeError, match="flight only"):
        ops.rollout_policy_flight(easy._cfg_t, easy._blob, ag_t.packed, *ag_t.conv_w, ag_t.hidden, ag_t.actions,
                                  torch.empty(B, 16 + 4 * n, device="cuda"), T, 0, 0.0, None, 0.0, 0.0, False, None, 0, 0, 0, 0, out_t["actions"], out_t["reward"],
                                  out_t["terminated"].view(torch.uint8), out_t["win"].view(torch.uint8), None, None)
