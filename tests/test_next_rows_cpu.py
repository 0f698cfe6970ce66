"""CPU tests of the 'next' rows f2 / f3: the replay buffer's FIFO index rule against indices
recorded from the reference ReplayBuffer, and the batched agent network against the reference RNN's forward on the
same seeded weights (fixtures made by tests/golden/gen_golden.py)."""
import json
import os
import types

import numpy as np
import torch

import cooperative_search_amd as cs
from golden_util import GOLDEN_DIR


def _rb_args(conv=False):
    return types.SimpleNamespace(n_actions=3, n_agents=3, state_shape=57, obs_shape=4, episode_limit=4, conv=conv,
                                 map_size=50)


def test_replay_storage_indices_match_reference():
    cases = json.load(open(os.path.join(GOLDEN_DIR, "replay_indices.json")))
    for case in cases:
        rb = cs.DeviceReplayBuffer(_rb_args(), case["size"], device="cpu")
        for st in case["steps"]:
            idx = rb._get_storage_idx(st["inc"])
            assert [int(v) for v in idx] == st["idx"]
            assert (rb.current_idx, rb.current_size) == (st["current_idx"], st["current_size"])
            if st["latest3"] is not None:
                assert rb.latest_indices(min(3, rb.current_size)) == st["latest3"]


def test_replay_store_and_sample_round_trip():
    args = _rb_args()
    rb = cs.DeviceReplayBuffer(args, 10, device="cpu")
    T, n = args.episode_limit, args.n_agents
    g = torch.Generator().manual_seed(0)

    def batch(k, tag):
        d = {}
        for key, buf in rb.buffers.items():
            d[key] = torch.full((k,) + tuple(buf.shape[1:]), float(tag))
        d["r"] = d["r"] + torch.arange(k, dtype=torch.float32).reshape(k, 1, 1) / 100
        return d
    rb.store_episode(batch(4, 1))
    assert not rb.can_sample(5) and rb.can_sample(4)
    rb.store_episode(batch(8, 2))          # wraps: rows 4..9 then 0..1
    assert rb.current_size == 10 and rb.current_idx == 2
    assert rb.buffers["o"][0, 0, 0, 0] == 2 and rb.buffers["o"][2, 0, 0, 0] == 1 and rb.buffers["o"][9, 0, 0, 0] == 2
    s = rb.sample(64, generator=g)
    assert s["o"].shape == (64, T, n, 4) and set(s["o"][:, 0, 0, 0].tolist()) <= {1.0, 2.0}
    latest = rb.sample_latest(3)
    assert latest["o"][:, 0, 0, 0].tolist() == [2.0, 2.0, 2.0]
    # flight: obs row is map_size**2 + 4 wide (replay_buffer.py:18-21)
    assert cs.DeviceReplayBuffer(_rb_args(conv=True), 2, device="cpu").buffers["o"].shape == (2, T, n, 2504)


def test_agent_rnn_matches_reference_forward_and_loads_its_state_dict():
    z = np.load(os.path.join(GOLDEN_DIR, "rnn_forward.npz"))
    for tag, conv in (("easy", False), ("flight", True)):
        args = types.SimpleNamespace(conv=conv, map_size=50, rnn_hidden_dim=64, n_actions=3, n_agents=3, obs_shape=4,
                                     dim_1=4, kernel_size_1=4, stride_1=2, dim_2=1, kernel_size_2=3, stride_2=1,
                                     padding_2=1, conv_out_dim=16, last_action=True, reuse_network=True)
        assert cs.rnn_input_shape(args) == (26 if conv else 10)
        net = cs.AgentRNN(cs.rnn_input_shape(args), args).double()
        sd = {k[len(tag) + 3:]: torch.from_numpy(z[k]).double() for k in z.files if k.startswith(tag + "_w_")}
        net.load_state_dict(sd)            # same parameter names as network/base_net.py
        with torch.no_grad():
            q, h2 = net(torch.from_numpy(z[tag + "_x"]), torch.from_numpy(z[tag + "_h"]))
        np.testing.assert_allclose(q.numpy(), z[tag + "_q"], rtol=0, atol=2e-6)     # weights stored as fp32
        np.testing.assert_allclose(h2.numpy(), z[tag + "_h2"], rtol=0, atol=2e-6)


def test_batched_choose_action_semantics():
    args = types.SimpleNamespace(conv=False, map_size=50, rnn_hidden_dim=64, n_actions=3, n_agents=3, obs_shape=4,
                                 last_action=True, reuse_network=True)
    torch.manual_seed(0)
    ag = cs.BatchedAgents(args, batch=5, device="cpu")
    obs, last = torch.rand(5, 3, 4), torch.zeros(5, 3, 3)
    a = ag.choose_action(obs, last, evaluate=True)
    assert a.shape == (5, 3) and a.dtype == torch.int64
    # one batched forward == the reference's per-agent batch-1 calls
    ag.init_hidden()
    q_rows = []
    for b in range(5):
        for i in range(3):
            x = torch.cat([obs[b, i], last[b, i], torch.eye(3)[i]]).unsqueeze(0)
            q, _ = ag.net(x, torch.zeros(1, 64))
            q_rows.append(int(q.argmax()))
    assert a.reshape(-1).tolist() == q_rows
    avail = torch.tensor([[1.0, 0.0, 0.0]] * 5)
    ag.init_hidden()
    assert (ag.choose_action(obs, last, avail=avail, evaluate=True) == 0).all()
    ag.init_hidden()
    assert (ag.choose_action(obs, last, avail=avail, epsilon=1.0, evaluate=False) == 0).all()


def test_replay_buffer_rejects_more_episodes_than_slots():
    """`inc > size` would hand out duplicate ring slots (several episodes racing for one entry); the
    reference's own index arithmetic fails on it as well (replay_buffer.py:84-101)."""
    import pytest
    buf = cs.DeviceReplayBuffer(_rb_args(), 4, device="cpu")
    assert list(buf._get_storage_idx(3)) == [0, 1, 2]
    with pytest.raises(ValueError):
        buf._get_storage_idx(5)
    assert list(buf._get_storage_idx(4)) == [3, 0, 1, 2]


def test_render_drawing_is_the_reference_scatter(tmp_path):
    """Row a13's optional host render (flight_env_easy.py:324-343): found targets orange, the others black, size 7; agents red
    triangles; title 'target_find:k/m'; axes 0..map_size -- drawn from plain host data, saved to a file on a headless box."""
    import pytest
    pytest.importorskip("matplotlib")
    from cooperative_search_amd.env import draw_env
    tgt = np.array([[5.0, 6.0], [20.0, 30.0], [44.0, 2.5]])
    ag = np.array([[0.0, 0.0], [25.0, 0.0]])
    out = tmp_path / "env0.png"
    ax = draw_env(tgt, 0b101, ag, 2, 3, 50, path=str(out))
    assert ax is not None and out.exists() and out.stat().st_size > 1000
    assert ax.get_title() == "target_find:2/3" and ax.get_xlim() == (0.0, 50.0) and ax.get_ylim() == (0.0, 50.0)
    cols = ax.collections
    assert len(cols) == 5   # one scatter per target and per agent, like the reference's loops
    import matplotlib.colors as mc
    face = [tuple(np.round(c.get_facecolor()[0][:3], 3)) for c in cols]
    want = [mc.to_rgb("orange"), mc.to_rgb("black"), mc.to_rgb("orange"), mc.to_rgb("red"), mc.to_rgb("red")]
    assert face == [tuple(np.round(w, 3)) for w in want]
    assert [float(c.get_sizes()[0]) for c in cols[:3]] == [7.0, 7.0, 7.0]
    assert [tuple(c.get_offsets()[0]) for c in cols] == [(5.0, 6.0), (20.0, 30.0), (44.0, 2.5), (0.0, 0.0), (25.0, 0.0)]
