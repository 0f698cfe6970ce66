"""Batched action selection: the caller of the env path.

The reference picks actions one agent at a time with batch 1 and a host->device copy per call
(/root/reference/agent/agent.py:33-75); once the env is batched that loop is the bottleneck.  `BatchedAgents` runs ONE
forward of the shared recurrent Q-network over all B*n (env, agent) rows and does the epsilon-greedy choice on the
device.  `AgentRNN` has the reference architecture and parameter names (network/base_net.py:5-46: [conv ->] fc1 ->
GRUCell -> fc2), so the reference's `*_rnn_net_params.pkl` state_dicts load unchanged.  Stock torch modules: nothing
custom is needed on ROCm for a 64-wide GRU.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F


class AgentRNN(nn.Module):
    def __init__(self, input_shape, args):
        super().__init__()
        self.args = args
        self.input_shape = input_shape
        if args.conv:
            self.conv_size = int((args.map_size - args.kernel_size_1) / args.stride_1 + 1)
            self.conv = nn.Sequential(nn.Conv2d(1, args.dim_1, args.kernel_size_1, args.stride_1), nn.ReLU(),
                                      nn.Conv2d(args.dim_1, args.dim_2, args.kernel_size_2, args.stride_2, args.padding_2),
                                      nn.ReLU())
            self.linear = nn.Linear(args.dim_2 * self.conv_size ** 2, args.conv_out_dim)
        self.fc1 = nn.Linear(input_shape, args.rnn_hidden_dim)
        self.rnn = nn.GRUCell(args.rnn_hidden_dim, args.rnn_hidden_dim)
        self.fc2 = nn.Sequential(nn.Linear(args.rnn_hidden_dim, args.rnn_hidden_dim), nn.ReLU(),
                                 nn.Linear(args.rnn_hidden_dim, args.n_actions))

    def forward(self, obs, hidden_state):
        if self.args.conv:
            cells = self.args.map_size ** 2
            prob = obs[:, :cells].reshape(-1, 1, self.args.map_size, self.args.map_size)
            feat = self.linear(self.conv(prob).reshape(-1, self.args.dim_2 * self.conv_size ** 2))
            obs = torch.cat([feat, obs[:, cells:]], 1)
        x = F.relu(self.fc1(obs))
        h = self.rnn(x, hidden_state.reshape(-1, self.args.rnn_hidden_dim))
        return self.fc2(h), h


def rnn_input_shape(args):
    """policy/qmix.py:19-25: obs_shape (+ n_actions if last_action) (+ n_agents if reuse_network) (+ conv_out_dim)."""
    shape = args.obs_shape
    if getattr(args, "last_action", True):
        shape += args.n_actions
    if getattr(args, "reuse_network", True):
        shape += args.n_agents
    if getattr(args, "conv", False):
        shape += args.conv_out_dim
    return shape


class BatchedAgents:
    """choose_action for every (env, agent) at once.  policy(obs, state, last_onehot, t) signature of collector.py."""

    def __init__(self, args, batch, device="cuda", net=None):
        self.args, self.batch, self.device = args, int(batch), torch.device(device)
        self.n_agents, self.n_actions = args.n_agents, args.n_actions
        self.net = (net or AgentRNN(rnn_input_shape(args), args)).to(self.device)
        self.agent_ids = torch.eye(self.n_agents, device=self.device).expand(self.batch, -1, -1)
        self.init_hidden()

    def init_hidden(self):
        self.hidden = torch.zeros(self.batch * self.n_agents, self.args.rnn_hidden_dim, device=self.device)

    @torch.no_grad()
    def choose_action(self, obs, last_onehot, avail=None, epsilon=0.0, evaluate=False, generator=None):
        """obs [B, n, obs_w], last_onehot [B, n, A] -> int64 actions [B, n] (agent/agent.py:33-75 semantics:
        inputs = obs ++ last_action ++ agent_id; q masked to -inf where unavailable; greedy when evaluating or with
        probability 1 - epsilon, otherwise uniform over the available actions)."""
        B, n, A = self.batch, self.n_agents, self.n_actions
        parts = [obs]
        if getattr(self.args, "last_action", True):
            parts.append(last_onehot)
        if getattr(self.args, "reuse_network", True):
            parts.append(self.agent_ids)
        x = torch.cat(parts, dim=2).reshape(B * n, -1)
        q, self.hidden = self.net(x, self.hidden)
        q = q.reshape(B, n, A)
        if avail is not None:
            q = q.masked_fill(avail.reshape(-1, 1, A).expand(B, n, A) == 0 if avail.dim() == 2 else avail == 0, float("-inf"))
        greedy = q.argmax(dim=2)
        if evaluate or epsilon <= 0.0:
            return greedy
        explore = torch.rand(B, n, device=self.device, generator=generator) < epsilon
        probs = torch.ones(B, n, A, device=self.device) if avail is None else (
            avail.reshape(-1, 1, A).expand(B, n, A) if avail.dim() == 2 else avail).to(torch.float32)
        rand_a = torch.multinomial(probs.reshape(B * n, A), 1, generator=generator).reshape(B, n)
        return torch.where(explore, rand_a, greedy)

    def policy(self, epsilon=0.0, evaluate=True, generator=None):
        def fn(obs, state, last_onehot, t):
            if t == 0:
                self.init_hidden()
            return self.choose_action(obs, last_onehot, None, epsilon, evaluate, generator)
        return fn


class FusedAgents:
    """`BatchedAgents` with the whole forward + choice in HIP (csrc/policy.hip): obs ++ one-hot(last action) ++
    one-hot(agent id) -> fc1 -> GRUCell -> fc2 -> argmax / epsilon-greedy in ONE launch (`cs_policy_forward`, fp32 on the
    matrix cores); for flight's conv network one more launch computes the 16 conv features of every env's probability map
    (`cs_policy_conv_features`, once per env: all its agents observe the same map).  Same parameters as `AgentRNN` (the
    reference's state_dict loads into `net`, then `load_weights()` repacks them).  Requires last_action and
    reuse_network (the reference's defaults, common/arguments.py:52-53), the reference's conv hyper-parameters
    (:256-265) and no availability mask (every action is always available in this env, flight_env_easy.py:184-188)."""

    CONV_HYPER = dict(map_size=50, dim_1=4, kernel_size_1=4, stride_1=2, dim_2=1, kernel_size_2=3, stride_2=1, padding_2=1,
                      conv_out_dim=16)

    def __init__(self, args, batch, device="cuda", net=None, seed=0, env_offset=0):
        """env_offset: global index of this shard's env 0 -- the exploration noise is keyed by the GLOBAL (env, agent)
        row, so a sharded batch picks the same actions as the whole one.  args.alg == 'reinforce' selects the softmax
        rule of agent/agent.py:77-97 instead of argmax / epsilon-greedy (:68-75)."""
        import ctypes as C

        from . import _lib
        self.conv = bool(getattr(args, "conv", False))
        if self.conv and any(getattr(args, k, None) != v for k, v in self.CONV_HYPER.items()):
            raise ValueError(f"FusedAgents' conv front end is built for {self.CONV_HYPER}; use BatchedAgents otherwise")
        if not (getattr(args, "last_action", True) and getattr(args, "reuse_network", True)):
            raise ValueError("FusedAgents needs last_action and reuse_network (the reference's defaults)")
        cells = args.map_size ** 2 if self.conv else 0
        # obs_shape is 4 for flight too (flight_env.py:34): the map is not counted, consumers add it when args.conv
        if args.rnn_hidden_dim != 64 or args.obs_shape != 4 or 4 + args.n_actions + args.n_agents > 16:
            raise ValueError("FusedAgents: rnn_hidden_dim must be 64, obs_shape 4 and 4 + n_actions + n_agents <= 16")
        self._C, self._lib = C, _lib
        self._L = _lib.load()
        # torch.ops.coopsearch.* (checks in C++, torch's stream) unless an experimental library is selected
        self._ops = _lib.pick_binding(None)[1]
        self.args, self.batch, self.device = args, int(batch), torch.device(device)
        self.n_agents, self.n_actions, self.cells = args.n_agents, args.n_actions, cells
        self.rows = self.batch * self.n_agents
        self.net = (net or AgentRNN(rnn_input_shape(args), args)).to(self.device)
        self.seed, self.calls = int(seed), 0
        self.row0 = int(env_offset) * self.n_agents
        self.softmax = getattr(args, "alg", None) == "reinforce"
        self.hidden = torch.zeros(self.rows, 64, device=self.device)
        # previous action on entry (-1 = none), chosen action on return: the kernel updates it in place
        self.actions = torch.full((self.batch, self.n_agents), -1, dtype=torch.int64, device=self.device)
        self.q = torch.zeros(self.batch, self.n_agents, self.n_actions, device=self.device)
        self.feat = torch.zeros(self.batch, 16, device=self.device) if self.conv else None
        self.load_weights()

    def load_weights(self):
        """Repack `self.net`'s parameters into MFMA fragment order (host side, once per weight update)."""
        import numpy as np
        C = self._C
        sd = {k: v.detach().to("cpu", torch.float32).contiguous().numpy() for k, v in self.net.state_dict().items()}
        order = ["fc1.weight", "fc1.bias", "rnn.weight_ih", "rnn.bias_ih", "rnn.weight_hh", "rnn.bias_hh",
                 "fc2.0.weight", "fc2.0.bias", "fc2.2.weight", "fc2.2.bias"]
        packed = np.zeros(self._L.cs_policy_packed_floats(), dtype=np.float32)
        rc = self._L.cs_policy_pack(*[C.c_void_p(sd[k].ctypes.data) for k in order], sd["fc1.weight"].shape[1],
                                    self.n_actions, C.c_void_p(packed.ctypes.data))
        if rc != 0:
            raise self._lib.CoopSearchError(self._L.cs_policy_last_error().decode())
        self.packed = torch.from_numpy(packed).to(self.device)
        if self.conv:  # the conv / linear tensors are used as they are (torch layout, fp32, contiguous device copies)
            self.conv_w = [self.net.state_dict()[k].detach().to(self.device, torch.float32).contiguous().clone()
                           for k in ("conv.0.weight", "conv.0.bias", "conv.2.weight", "conv.2.bias", "linear.weight",
                                     "linear.bias")]

    def _stream(self):
        return self._C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _check(self, rc):
        if rc != 0:
            raise self._lib.CoopSearchError(self._L.cs_policy_last_error().decode())

    def _on_device(self):
        """ctypes route: the launches go to the process's current device, so this object's device is made current."""
        return torch.cuda.device(self.device)

    def _conv_features(self, maps, map_stride, n_maps, feat):
        if self._ops is not None:
            self._ops.policy_conv_features(*self.conv_w, maps, int(map_stride), int(n_maps), feat)
            return
        vp = lambda t: self._C.c_void_p(t.data_ptr())
        with self._on_device():
            self._check(self._L.cs_policy_conv_features(*[vp(w) for w in self.conv_w], vp(maps), map_stride, n_maps,
                                                        vp(feat), self._stream()))

    def init_hidden(self):
        self.hidden.zero_()
        self.actions.fill_(-1)

    def selection(self, epsilon, evaluate):
        """(epsilon, CS_SELECT_* flags) of one choose_action call.  argmax rule (agent.py:68-75): greedy when evaluating.
        Softmax rule (:77-97): epsilon always enters prob; the draw is replaced by argmax only if epsilon == 0 and
        evaluate."""
        if not self.softmax:
            return (0.0 if evaluate else float(epsilon)), 0
        sample = not (float(epsilon) == 0.0 and evaluate)
        return float(epsilon), self._lib.SELECT_SOFTMAX | (self._lib.SELECT_SAMPLE if sample else 0)

    def choose_action(self, obs, epsilon=0.0, evaluate=False, want_q=False, last=None, out=None, eps_env=None):
        """obs: float32 [B, n, obs_shape] device tensor, contiguous (the env's live obs buffer works directly).
        Returns the int64 [B, n] action buffer (overwritten by the next call); it is also remembered as the
        next call's last action, like rollout.py:55-63.  A collector can pass `last` (int64 [B, n] previous actions,
        -1 = none) and `out` (int64 [B, n] destination, e.g. row t of its action table) to avoid any copy.
        eps_env (float64 [B] device tensor): every env explores with its OWN epsilon (the per-env exploration schedule,
        collector.EpsilonSchedule) instead of the scalar; ignored when evaluating, like rollout.py:35."""
        C = self._C
        width = self.cells + 4
        if obs.dtype != torch.float32 or obs.shape[-1] != width or obs.numel() != self.rows * width:
            raise ValueError(f"obs must be float32 [B, n, {width}]")
        if not obs.is_contiguous():
            obs = obs.contiguous()
        last = self.actions if last is None else last
        out = self.actions if out is None else out
        for t in (last, out):
            if t.dtype != torch.int64 or t.numel() != self.rows or not t.is_contiguous():
                raise ValueError("last / out must be contiguous int64 [B, n]")
        vp = lambda t: C.c_void_p(t.data_ptr())
        if self.conv:  # the map of an env's first row stands for all its rows (flight_env.py:223-230)
            self._conv_features(obs, self.n_agents * width, self.batch, self.feat)
        eps, sel = self.selection(epsilon, evaluate)
        if evaluate and not self.softmax:
            eps_env = None   # epsilon = 0 if evaluate (rollout.py:35)
        if eps_env is not None and (eps_env.dtype != torch.float64 or eps_env.numel() != self.batch or not eps_env.is_contiguous()):
            raise ValueError("eps_env must be a contiguous float64 [B] tensor")
        if self._ops is not None:
            self._ops.policy_forward(self.packed, obs, width, self.cells, last, self.feat if self.conv else None, self.n_agents,
                                     self.hidden, self.q if want_q else None, out, self.rows, self.n_agents, self.n_actions, eps,
                                     eps_env, self.seed, self.calls, self.row0, sel)
            self.calls += 1
            return out
        with self._on_device():
            self._check(self._L.cs_policy_forward(vp(self.packed), vp(obs), width, self.cells, vp(last),
                                                  vp(self.feat) if self.conv else None, self.n_agents, vp(self.hidden),
                                                  vp(self.q) if want_q else None, vp(out), self.rows, self.n_agents,
                                                  self.n_actions, eps, vp(eps_env) if eps_env is not None else None, self.seed,
                                                  self.calls, self.row0, sel, self._stream()))
        self.calls += 1
        return out

    def forward_raw(self, x, want_q=True):
        """Forward on caller-assembled input rows x [rows, (map_size^2 +) 4 + n_actions + n_agents] (greedy choice; conv
        features per ROW here, since raw rows need not share maps)."""
        C = self._C
        x = x.to(torch.float32).contiguous()
        vp = lambda t: C.c_void_p(t.data_ptr())
        feat = None
        if self.conv:
            feat = torch.empty(self.rows, 16, device=self.device)
            self._conv_features(x, x.stride(0), self.rows, feat)
        if self._ops is not None:
            self._ops.policy_forward(self.packed, x, x.stride(0), self.cells, None, feat, 1, self.hidden,
                                     self.q if want_q else None, self.actions, self.rows, self.n_agents, self.n_actions, 0.0,
                                     None, self.seed, self.calls, self.row0, 0)
            return self.actions
        with self._on_device():
            self._check(self._L.cs_policy_forward(vp(self.packed), vp(x), x.stride(0), self.cells, None,
                                                  vp(feat) if self.conv else None, 1, vp(self.hidden),
                                                  vp(self.q) if want_q else None, vp(self.actions), self.rows, self.n_agents,
                                                  self.n_actions, 0.0, None, self.seed, self.calls, self.row0, 0, self._stream()))
        return self.actions

    def policy(self, epsilon=0.0, evaluate=True):
        def fn(obs, state, last_onehot, t):
            if t == 0:
                self.init_hidden()
            return self.choose_action(obs, epsilon, evaluate)
        return fn
