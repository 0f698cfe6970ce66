"""Host-side mirror of the reference env protocol over the HIP path.

Reference boundary (duck-typed object consumed by common/rollout.py and main.py):
    Env(args, circle_dict); get_env_info(); reset(init=False); get_obs(); get_state();
    get_avail_agent_actions(i); step(act_list) -> (reward, terminated, win_flag); .target_find; render(); close()

`BatchedFlightEnv` keeps those names and argument meanings for B environments at once and returns device
tensors.  `FlightSearchEnvEasy` / `FlightSearchEnv` are B = 1 adapters that return exactly the reference's
Python types (int / bool / fresh float64 ndarrays) so a rollout.py-shaped loop runs unchanged.

All environment arithmetic happens in csrc/coopsearch.hip behind the C ABI of include/coopsearch.h; torch only
owns device memory and the stream.  There is no CPU fallback: without a GPU and the built library this raises.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from .targets import default_circle_dict

DEFAULT_BASE_SEED = 20240000  # env b of the global batch is seeded base + b
STEP_ADVANCE_EVERY = 32        # step(): hit tapes are refreshed every this many single steps ...
STEP_ADVANCE_MIN_AHEAD = 400   # ... for the rows with fewer twisted MT19937 words left than this (of 624)


def _cfg_from_args(args, circle_dict, batch, variant):
    cfg = _lib.CsConfig()
    cfg.variant = variant
    cfg.n_agents = int(args.n_agents)
    cfg.n_targets = int(args.target_num)
    cfg.map_size = int(args.map_size)
    cfg.view_range = int(args.view_range)
    cfg.time_limit = int(args.time_limit)
    cfg.agent_mode = int(args.agent_mode)
    cfg.target_mode = int(args.target_mode)
    cfg.velocity = float(args.agent_velocity)
    cfg.safe_dist = float(args.safe_dist)
    cfg.detect_prob = float(args.detect_prob)
    cfg.force_dist = float(args.force_dist)
    cfg.force_factor = 0.8  # POTENTIAL_FORCE_FACTOR, flight_env_easy.py:65
    if cfg.target_mode == 0:
        if len(circle_dict["x"]) < cfg.n_targets:
            raise Exception("target file has fewer rows than target_num")
        for j in range(min(cfg.n_targets, _lib.MAX_TARGETS)):
            cfg.cx[j], cfg.cy[j] = float(circle_dict["x"][j]), float(circle_dict["y"][j])
            cfg.dx[j], cfg.dy[j] = float(circle_dict["dx"][j]), float(circle_dict["dy"][j])
            cfg.deter[j] = 1 if circle_dict["deter"][j] == "t" else 0
    cfg.batch = int(batch)
    return cfg


class BatchedFlightEnv:
    """B independent flight_easy / flight environments resident in HBM, stepped by HIP kernels.

    args        namespace with the reference's fields (common/arguments.py:27-34, :233-284); `args.env`
                selects the variant ('flight' -> probability-map env, anything else -> flight_easy)
    circle_dict dict from load_targets (main.py:19-32); defaults to the shipped flight_targets.txt
    batch       environments on this device
    seeds       per-env np.random.seed values (uint32 array-like), default DEFAULT_BASE_SEED + env_offset + b
    env_offset  global index of env 0 (multi-GPU sharding: results do not depend on how the batch is split)
    freeze_done terminated envs ignore step() (reward 0, terminated 1); False reproduces the reference, which has
                no terminal guard
    auto_reset  terminated envs are reset(init=False) at the start of the next step()
    binding     "torch" (default): the calls go through torch.ops.coopsearch.* (csrc/torch_ops.cpp: tensor checks in C++,
                torch's current HIP stream); "ctypes": straight to the C ABI with data_ptr()s -- same library, same
                kernels, no torch in the call path
    kernel      flight_easy only: "group" (16 lanes per env: lowest step latency; a rollout is T launches of the step
                kernel), "oct" (rollout only: 8 lanes per env, lane t owns agent t and targets t, t + 8), "od"
                (rollout only: the octet layout with a kinematics wavefront running steps ahead of a detection
                wavefront), "ode" ("od" with a third wavefront per 8 envs that writes the outputs), "lane" (one env per
                lane: no replicated arithmetic, for large batches) or "lanev" (the second-generation lane-per-env kernel, teams of up to 5) or "auto" (rollout: "ode" up to
                8192 envs, "od" up to 16384, "oct" below 65536 envs (teams of 6 to 8: below 2^20), "lanev" from 65536 for teams of up
                to 5, "lane" from 2^20 for larger ones; single steps: the 16-lane step kernel, the lane-per-env kernel from 32768;
                the one table is DESIGN.md section 4).
                All produce bit-identical results.
    check_actions  validate every action on the device before stepping (CS_CHECK_ACTIONS): a value outside 0..2 raises the
                reference's IndexError ("list index out of range", dyaw[act], flight_env_easy.py:262) and leaves the envs
                untouched; costs a stream synchronisation per call.  None (default): on for batches of up to 64 envs.
                Without it the kernels treat any value other than 1 / 2 as action 0.
    step_advance  step(): refresh the hit tapes every STEP_ADVANCE_EVERY single steps (default).  False leaves every
                MT19937 word to be twisted on demand by the step kernel itself -- same results, one more dependent load per
                launch.
    """

    def __init__(self, args, circle_dict=None, batch=1, device="cuda", seeds=None, env_offset=0, freeze_done=True,
                 auto_reset=False, variant=None, kernel="auto", binding=None, step_advance=True, check_actions=None):
        if not torch.cuda.is_available():
            raise RuntimeError("BatchedFlightEnv needs a GPU: the HIP path has no CPU fallback")
        self._L = _lib.load()
        self.binding, self._ops = _lib.pick_binding(binding)
        self.args = args
        if variant is None:
            variant = "flight" if getattr(args, "env", "flight_easy") == "flight" else "flight_easy"
        self.variant = variant
        self.flight = variant == "flight"
        self.map_size = int(args.map_size)
        self.target_num = int(args.target_num)
        self.target_mode = int(args.target_mode)
        self.agent_mode = int(args.agent_mode)
        self.n_agents = int(args.n_agents)
        self.view_range = args.view_range
        self.time_limit = int(args.time_limit)
        self.detect_prob = args.detect_prob
        self.safe_dist = args.safe_dist
        self.velocity = args.agent_velocity
        self.force_dist = args.force_dist
        self.n_actions = 3
        self.state_shape = self.n_agents * 4 + self.target_num * 3
        self.obs_shape = 4  # reported as 4 for flight too (quirk Q8); consumers add map_size**2 when args.conv
        self.batch = int(batch)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("BatchedFlightEnv: device must be a GPU")
        # same messages / same moment (construction runs reset) as flight_env_easy.py:136,180
        if self.target_mode not in (0, 1):
            raise Exception("No such target mode")
        if self.agent_mode not in (0, 1, 2, 3):
            raise Exception("No such agent mode")
        self.circle_dict = circle_dict if circle_dict is not None else default_circle_dict()
        self.cfg = _cfg_from_args(args, self.circle_dict, self.batch, 1 if self.flight else 0)
        self._cfgp = C.byref(self.cfg)
        self._cfg_t = torch.frombuffer(bytearray(bytes(self.cfg)), dtype=torch.uint8)   # the struct's bytes, for the ops
        lay = _lib.CsLayout()
        self._call(self._L.cs_state_layout, self._cfgp, C.byref(lay))
        self.layout = lay
        B, n, m = self.batch, self.n_agents, self.target_num
        self.cells = self.map_size * self.map_size
        self.obs_width = self.cells + 4 if self.flight else 4
        with torch.cuda.device(self.device):
            self._blob = torch.empty(lay.total_bytes, dtype=torch.uint8, device=self.device)
            self._reward = torch.zeros(B, dtype=torch.float32, device=self.device)
            self._terminated = torch.zeros(B, dtype=torch.uint8, device=self.device)
            self._win = torch.zeros(B, dtype=torch.uint8, device=self.device)
            self.step_advance = bool(step_advance)
            # CS_CHECK_ACTIONS: None = the library's rule (batches of up to 64 envs, COOPSEARCH_CHECK_ACTIONS overrides)
            self.check_actions = _lib.check_actions_default(int(batch)) if check_actions is None else bool(check_actions)
            self._steps_since_advance = STEP_ADVANCE_EVERY   # the first step() refreshes the tapes
            self._obs = torch.zeros(B, n, self.obs_width, dtype=torch.float32, device=self.device)
            self._state = torch.zeros(B, self.state_shape, dtype=torch.float32, device=self.device)
            self._avail = torch.ones(B, self.n_actions, dtype=torch.float32, device=self.device)
            self._metrics = torch.zeros(4, dtype=torch.float64, device=self.device)
        if kernel not in ("auto", "group", "lane", "lanev", "oct", "od", "ode"):   # ("solo" / "duo", the 16-lane rollout kernels of rounds 1-2, were removed in round 6)
            raise ValueError("kernel must be 'auto', 'group', 'lane', 'lanev', 'oct', 'od' or 'ode'")
        self.kernel = kernel
        self.freeze_done = bool(freeze_done)
        self.auto_reset = bool(auto_reset)
        if self.freeze_done and self.auto_reset:
            self.freeze_done = False
        self.env_offset = int(env_offset)
        if self._ops is not None:
            assert int(self._ops.state_bytes(self._cfg_t)) == lay.total_bytes
            self._ops.env_init(self._cfg_t, self._blob)
        else:
            self._call(self._L.cs_init, self._cfgp, self._blob.data_ptr(), self._stream())
        if seeds is None:
            seeds = (DEFAULT_BASE_SEED + self.env_offset + np.arange(B, dtype=np.int64)) % (1 << 32)
        self.seed(seeds)
        self.reset(init=True)  # FlightSearchEnvEasy.__init__ ends with self.reset(init=True), :68

    # ------------------------------------------------------------------------------------------------ plumbing
    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def _call(self, fn, *args):
        """ctypes route: the cs_* entry points launch on the process's current device, so the env's device is made current
        for the call (the torch ops do the same in C++ with a HIPGuard); an env on cuda:1 works while cuda:0 is current."""
        with torch.cuda.device(self.device):
            _lib.check(fn(*args))

    def _view(self, off, count, dtype, shape):
        itemsize = torch.empty((), dtype=dtype).element_size()
        return self._blob[off:off + count * itemsize].view(dtype).view(*shape)

    def raw(self):
        """Zero-copy typed views of the state blob (layout: include/coopsearch.h, cs_layout)."""
        B, lay = self.batch, self.layout
        d = dict(
            tgt=self._view(lay.tgt_off, B * 32, torch.float64, (B, 16, 2)),
            agent=self._view(lay.agent_off, B * 32, torch.float64, (B, 8, 4)),
            hdr=self._view(lay.hdr_off, B * 16, torch.int32, (B, 16)),
            mt=self._view(lay.mt_off, B * _lib.MT_STRIDE, torch.int32, (B, _lib.MT_STRIDE)),
            ahead=self._view(lay.ahead_off, B, torch.int32, (B,)),
        )
        if self.flight:
            d["prob"] = self._view(lay.prob_off, B * self.cells, torch.float32, (B, self.map_size, self.map_size))
        return d

    def mt_advance(self, min_ahead=400):
        """cs_mt_advance: rows with fewer than `min_ahead` twisted words ahead of their cursor are twisted fully ahead and
        their hit tapes rebuilt (when, never what: the streams are untouched)."""
        if self._ops is not None:
            self._ops.mt_advance(self._cfg_t, self._blob, int(min_ahead))
        else:
            self._call(self._L.cs_mt_advance, self._cfgp, self._blob.data_ptr(), int(min_ahead), self._stream())
        self._steps_since_advance = 0

    def mt_canonical(self):
        """int32 [B, MT_STRIDE]: every env's MT19937 row in a form that depends only on the stream position (the kernels may
        leave different amounts of the row pre-twisted ahead of the cursor; see cs_mt_canonical)."""
        out = torch.empty(self.batch, _lib.MT_STRIDE, dtype=torch.int32, device=self.device)
        if self._ops is not None:
            self._ops.mt_canonical(self._cfg_t, self._blob, out)
        else:
            self._call(self._L.cs_mt_canonical, self._cfgp, self._blob.data_ptr(), out.data_ptr(), self._stream())
        return out

    def seed(self, seeds):
        """np.random.seed(seeds[b]) for env b's private NumPy-compatible stream."""
        s = np.ascontiguousarray(np.asarray(seeds, dtype=np.uint64) & 0xFFFFFFFF, dtype=np.uint32)
        if s.shape != (self.batch,):
            raise ValueError("seeds must have shape (batch,)")
        t = torch.from_numpy(s.view(np.int32)).to(self.device)
        if self._ops is not None:
            self._ops.env_seed(self._cfg_t, self._blob, t)
        else:
            self._call(self._L.cs_seed, self._cfgp, self._blob.data_ptr(), t.data_ptr(), self._stream())
        self._seeds_keepalive = t

    # ------------------------------------------------------------------------------------------- reference API
    def get_env_info(self):
        """flight_env_easy.py:71-77 (+ 'batch')."""
        return {"n_actions": self.n_actions, "state_shape": self.state_shape, "obs_shape": self.obs_shape,
                "episode_limit": self.time_limit, "batch": self.batch}

    def reset(self, init=False, mask=None):
        """env.reset(init) for every env (or those with mask[b] != 0).  flight: init=True clears the map."""
        mptr = None
        if mask is not None:
            mask = torch.as_tensor(mask, device=self.device)
            if mask.dtype != torch.uint8:
                mask = mask.to(torch.uint8)
            mask = mask.contiguous()
            if mask.shape != (self.batch,):
                raise ValueError("mask must have shape (batch,)")
            mptr = mask.data_ptr()
        if self._ops is not None:
            self._ops.env_reset(self._cfg_t, self._blob, mask, bool(init), self._obs, self._state)
            return
        self._call(self._L.cs_reset, self._cfgp, self._blob.data_ptr(), mptr, 1 if init else 0,
                                    self._obs.data_ptr(), self._state.data_ptr(), self._stream())

    def _actions(self, actions, lead_shape):
        if not torch.is_tensor(actions):
            actions = torch.as_tensor(np.asarray(actions), device=self.device)
        if actions.device != self.device:
            actions = actions.to(self.device)
        if actions.dim() == 0 or actions.shape[-1] != self.n_agents:
            raise Exception("Act num mismatch agent")  # flight_env_easy.py:256-257
        if tuple(actions.shape) != tuple(lead_shape) + (self.n_agents,):
            raise ValueError(f"actions must have shape {tuple(lead_shape) + (self.n_agents,)}")
        if actions.dtype not in (torch.int32, torch.int64):
            actions = actions.to(torch.int64)
        return actions.contiguous()

    def _flags(self, actions):
        f = 0
        if self.freeze_done:
            f |= _lib.FREEZE_DONE
        if self.auto_reset:
            f |= _lib.AUTO_RESET
        if actions.dtype == torch.int64:
            f |= _lib.ACTIONS_I64
        if self.check_actions:
            f |= _lib.CHECK_ACTIONS
        elif self.binding == "torch":   # the op layer applies its own default unless told: off means off (csrc/torch_ops.cpp)
            f |= _lib.OP_NO_CHECK_ACTIONS
        if self.kernel == "group":
            f |= _lib.KERNEL_GROUP
        elif self.kernel == "lane":   # one env per lane, first generation (k_rollout_lane)
            f |= _lib.KERNEL_LANE
        elif self.kernel == "lanev":  # one env per lane, targets in registers (k_rollout_lanev; teams of up to 5)
            f |= _lib.KERNEL_LANEV
        elif self.kernel == "oct":   # rollout(): 8 lanes per env; step() has no octet variant and uses the 16-lane kernel
            f |= _lib.KERNEL_OCT
        elif self.kernel == "od":    # rollout(): the octet layout, kinematics and detection wavefronts pipelined
            f |= _lib.KERNEL_OD
        elif self.kernel == "ode":   # ... plus an emitting wavefront
            f |= _lib.KERNEL_ODE
        return f

    def step(self, actions, out=None):
        """env.step(act_list) for every env: actions [B, n] in {0,1,2} -> (reward[B] f32, terminated[B] bool,
        win[B] bool).  The returned tensors (and get_obs/get_state) are overwritten by the next step.
        `out` (dict with any of reward/terminated/win/obs/state, contiguous tensors of the right shape) makes the
        kernel write those results straight into caller buffers (a collector's [T, ...] rows) instead of the live
        views; the live get_obs()/get_state() then lag until the next plain step() or refresh()."""
        a = self._actions(actions, (self.batch,))
        dst = dict(reward=self._reward, terminated=self._terminated, win=self._win, obs=self._obs, state=self._state)
        if out:
            for k, v in out.items():
                if v is None:
                    continue
                if k not in dst or v.numel() * v.element_size() != dst[k].numel() * dst[k].element_size() \
                        or not v.is_contiguous() or v.device != dst[k].device:
                    raise ValueError(f"step(out=): bad destination for {k!r}")
                dst[k] = v
        # single steps read their draws from the env's hit tape while it is valid; every STEP_ADVANCE_EVERY steps the rows
        # that are running low are twisted ahead again (cs_mt_advance: a coalesced pass that skips the others), which keeps
        # the MT19937 window load off the critical path of every cs_step launch.  Never changes a stream, only when its
        # words are regenerated.
        if self.step_advance and self._steps_since_advance >= STEP_ADVANCE_EVERY and self.n_agents <= 5:
            self.mt_advance(STEP_ADVANCE_MIN_AHEAD)
        self._steps_since_advance += 1
        if self._ops is not None:
            self._ops.env_step(self._cfg_t, self._blob, a, self._flags(a), dst["reward"], dst["terminated"].view(torch.uint8),
                               dst["win"].view(torch.uint8), dst["obs"], dst["state"])
        else:
            self._call(self._L.cs_step, self._cfgp, self._blob.data_ptr(), a.data_ptr(), self._flags(a),
                                       dst["reward"].data_ptr(), dst["terminated"].data_ptr(), dst["win"].data_ptr(),
                                       dst["obs"].data_ptr(), dst["state"].data_ptr(), self._stream())
        return dst["reward"], dst["terminated"].view(torch.bool), dst["win"].view(torch.bool)

    def rollout(self, actions, emit=True, out=None, update_views=True):
        """T steps from one call (flight_easy: ONE launch; flight: one launch per step, in which the map sweep of step t runs
        beside the kinematics / detection of step t + 1): actions [T, B, n] -> dict of
        [T, B, ...] tensors.
        `out` reuses caller buffers (keys reward/terminated/win/obs/state); update_views=False skips refreshing
        the live get_obs()/get_state() buffers afterwards (they then lag until the next step/refresh)."""
        T = int(actions.shape[0])
        a = self._actions(actions, (T, self.batch))
        B, n = self.batch, self.n_agents
        if out is None:
            out = dict(
                reward=torch.empty(T, B, dtype=torch.float32, device=self.device),
                terminated=torch.empty(T, B, dtype=torch.uint8, device=self.device),
                win=torch.empty(T, B, dtype=torch.uint8, device=self.device),
                obs=torch.empty(T, B, n, self.obs_width, dtype=torch.float32, device=self.device) if emit else None,
                state=torch.empty(T, B, self.state_shape, dtype=torch.float32, device=self.device) if emit else None,
            )
        else:
            emit = out.get("obs") is not None and out.get("state") is not None
            out = dict(out)
        term = out["terminated"].view(torch.uint8)
        win = out["win"].view(torch.uint8)
        if self._ops is not None:
            self._ops.env_rollout(self._cfg_t, self._blob, a, self._flags(a), out["reward"], term, win,
                                  out["obs"] if emit else None, out["state"] if emit else None)
        else:
            self._call(self._L.cs_rollout, self._cfgp, self._blob.data_ptr(), a.data_ptr(), T, self._flags(a),
                                          out["reward"].data_ptr(), term.data_ptr(), win.data_ptr(),
                                          out["obs"].data_ptr() if emit else None,
                                          out["state"].data_ptr() if emit else None, self._stream())
        if update_views:
            if emit:
                self._obs.copy_(out["obs"][-1])
                self._state.copy_(out["state"][-1])
            else:
                self.refresh()
        out["terminated"] = term.view(torch.bool)
        out["win"] = win.view(torch.bool)
        return out

    def epsilon_step(self, eps_env, anneal, min_epsilon, trace_row=None):
        """One step of the exploration schedule for a loop that calls choose_action / step itself (cs_epsilon_step): the envs
        the NEXT step() will execute anneal -- eps = eps - anneal if eps > min_epsilon else eps, common/rollout.py:75-76 --;
        trace_row (float64 [B]) receives the values before the anneal.  Call between choose_action and step."""
        flags = (_lib.FREEZE_DONE if self.freeze_done else 0) | (_lib.AUTO_RESET if self.auto_reset else 0)
        if self._ops is not None:
            self._ops.epsilon_step(self._cfg_t, self._blob, flags, eps_env, float(anneal), float(min_epsilon), trace_row)
            return
        self._call(self._L.cs_epsilon_step, self._cfgp, self._blob.data_ptr(), flags, eps_env.data_ptr(), float(anneal),
                   float(min_epsilon), trace_row.data_ptr() if trace_row is not None else None, self._stream())

    def rollout_policy(self, agents, T, epsilon=0.0, evaluate=True, emit=True, out=None, update_views=True, eps_env=None,
                       anneal=0.0, min_epsilon=0.0, per_step=False, eps_trace=None):
        """T closed-loop steps in ONE call: each step runs `agents`' network (a `FusedAgents`) on the current observation,
        picks the actions and steps the envs -- exactly what T x
        `env.step(agents.choose_action(env.get_obs(), epsilon, evaluate))` computes.  flight_easy (n <= 5): one launch,
        with the hidden state, the actions and the envs resident on chip in between.  flight: three kernels per step
        enqueued by one call (cs_rollout_policy_flight): the conv front end reads every env's map where it lives, and
        with emit=False the n observation copies of the map are never written.  Returns the `rollout` dict plus `actions`
        (int64 [T, B, n]); `agents.hidden` / `agents.actions` / `agents.calls` advance as if the T calls had been made.
        Exploration schedule (common/rollout.py:35-41, 75-76, 133-135; include/coopsearch.h cs_epsilon): eps_env (float64 [B],
        in / out) gives every env its own epsilon, annealed ON THE DEVICE after every step the env executes when per_step
        (`epsilon_anneal_scale == 'step'`: eps = eps - anneal if eps > min_epsilon else eps) and carried back in the same
        tensor; eps_trace (float64 [T, B]) records what every step's selection used.  Ignored when evaluating."""
        T = int(T)
        B, n = self.batch, self.n_agents
        if agents.rows != B * n or bool(getattr(agents, "conv", False)) != self.flight:
            raise ValueError("rollout_policy: `agents` must be a FusedAgents for this env's batch "
                             "(with the conv front end for flight, without it for flight_easy)")
        out = dict(out) if out else {}
        dev = self.device
        spec = dict(reward=((T, B), torch.float32), terminated=((T, B), torch.uint8), win=((T, B), torch.uint8),
                    actions=((T, B, n), torch.int64))
        if emit or out.get("obs") is not None:
            spec.update(obs=((T, B, n, self.obs_width), torch.float32), state=((T, B, self.state_shape), torch.float32))
        for k, (shape, dt) in spec.items():
            if out.get(k) is None:
                out[k] = torch.empty(shape, dtype=dt, device=dev)
            elif out[k].numel() * out[k].element_size() != torch.Size(shape).numel() * dt.itemsize or not out[k].is_contiguous():
                raise ValueError(f"rollout_policy(out=): bad destination for {k!r}")
        has_obs = out.get("obs") is not None and out.get("state") is not None
        flags = (_lib.FREEZE_DONE if self.freeze_done else 0) | (_lib.AUTO_RESET if self.auto_reset else 0)
        sel_eps, sel_flags = agents.selection(epsilon, evaluate)
        if evaluate and not agents.softmax:
            eps_env = eps_trace = None   # epsilon = 0 if evaluate (rollout.py:35): no schedule
        if eps_trace is not None and eps_env is None:
            raise ValueError("rollout_policy: eps_trace needs eps_env")
        for name, tns, numel in (("eps_env", eps_env, B), ("eps_trace", eps_trace, T * B)):
            if tns is not None and (tns.dtype != torch.float64 or tns.numel() != numel or not tns.is_contiguous() or not tns.is_cuda
                                    or (tns.device.index or 0) != (dev.index or 0)):
                raise ValueError(f"rollout_policy: {name} must be a contiguous float64 device tensor of {numel} elements")
        sched_t = (sel_eps, eps_env, float(anneal), float(min_epsilon), bool(per_step), eps_trace)
        if self._ops is None:
            sched_c = _lib.CsEpsilon(sel_eps, float(anneal), float(min_epsilon), 1 if per_step else 0, 0,
                                     eps_env.data_ptr() if eps_env is not None else None,
                                     eps_trace.data_ptr() if eps_trace is not None else None)
        if self.flight:
            scratch = getattr(agents, "_flight_scratch", None)
            if scratch is None or scratch.numel() != B * (16 + 4 * n):
                scratch = agents._flight_scratch = torch.empty(B, 16 + 4 * n, dtype=torch.float32, device=dev)
            if self._ops is not None:
                self._ops.rollout_policy_flight(self._cfg_t, self._blob, agents.packed, *agents.conv_w, agents.hidden,
                                                agents.actions, scratch, T, flags, *sched_t, agents.seed, agents.calls,
                                                agents.row0, sel_flags, out["actions"], out["reward"],
                                                out["terminated"].view(torch.uint8), out["win"].view(torch.uint8),
                                                out["obs"] if has_obs else None, out["state"] if has_obs else None)
            else:
                self._call(self._L.cs_rollout_policy_flight,
                    self._cfgp, self._blob.data_ptr(), agents.packed.data_ptr(), *[w.data_ptr() for w in agents.conv_w],
                    agents.hidden.data_ptr(), agents.actions.data_ptr(), scratch.data_ptr(), T, flags, C.byref(sched_c), agents.seed,
                    agents.calls, agents.row0, sel_flags, out["actions"].data_ptr(), out["reward"].data_ptr(),
                    out["terminated"].data_ptr(), out["win"].data_ptr(), out["obs"].data_ptr() if has_obs else None,
                    out["state"].data_ptr() if has_obs else None, self._stream())
        elif self._ops is not None:
            self._ops.rollout_policy(self._cfg_t, self._blob, agents.packed, agents.hidden, agents.actions, T, flags, *sched_t,
                                     agents.seed, agents.calls, agents.row0, sel_flags, out["actions"], out["reward"],
                                     out["terminated"].view(torch.uint8), out["win"].view(torch.uint8),
                                     out["obs"] if has_obs else None, out["state"] if has_obs else None)
        else:
            self._call(self._L.cs_rollout_policy,
                self._cfgp, self._blob.data_ptr(), agents.packed.data_ptr(), agents.hidden.data_ptr(),
                agents.actions.data_ptr(), T, flags, C.byref(sched_c), agents.seed, agents.calls, agents.row0, sel_flags,
                out["actions"].data_ptr(), out["reward"].data_ptr(), out["terminated"].data_ptr(), out["win"].data_ptr(),
                out["obs"].data_ptr() if has_obs else None, out["state"].data_ptr() if has_obs else None, self._stream())
        agents.calls += T
        agents.actions.copy_(out["actions"][-1])
        if update_views:
            if has_obs:
                self._obs.copy_(out["obs"][-1])
                self._state.copy_(out["state"][-1])
            else:
                self.refresh()
        out["terminated"] = out["terminated"].view(torch.bool)
        out["win"] = out["win"].view(torch.bool)
        return out

    def refresh(self):
        """Re-emit get_obs()/get_state() from the device state (after editing raw())."""
        if self._ops is not None:
            self._ops.env_emit(self._cfg_t, self._blob, self._obs, self._state)
            return
        self._call(self._L.cs_emit, self._cfgp, self._blob.data_ptr(), self._obs.data_ptr(), self._state.data_ptr(),
                                   self._stream())

    def get_obs(self):
        """[B, n, 4] (flight: [B, n, map*map + 4], map first) float32 -- live buffer."""
        return self._obs

    def get_state(self):
        """[B, 4n + 3m] float32 -- live buffer."""
        return self._state

    def get_avail_agent_actions(self, agent_id):
        if agent_id >= self.n_agents:
            raise Exception("Agent id out of range")  # flight_env_easy.py:185-186
        return self._avail

    @property
    def target_find(self):
        """int32 [B]: running count of found targets (flight_env_easy.py:42)."""
        return self.raw()["hdr"][:, _lib.H_TARGET_FIND]

    @property
    def time_step(self):
        return self.raw()["hdr"][:, _lib.H_TIME_STEP]

    @property
    def total_reward(self):
        return self.raw()["hdr"][:, _lib.H_TOTAL_REWARD]

    @property
    def win_flag(self):
        return (self.raw()["hdr"][:, _lib.H_FLAGS] & 1).to(torch.bool)

    def _done_mask(self):
        h = self.raw()["hdr"]
        return ((h[:, _lib.H_TARGET_FIND] >= self.target_num) | (h[:, _lib.H_TIME_STEP] >= self.time_limit)).to(torch.uint8)

    def metric_partials(self):
        """float64[4] on device: sum total_reward, sum win, sum target_find, env count (runner.py:86-96)."""
        self._metrics.zero_()
        if self._ops is not None:
            self._ops.env_metrics(self._cfg_t, self._blob, self._metrics)
            return self._metrics
        self._call(self._L.cs_metrics, self._cfgp, self._blob.data_ptr(), self._metrics.data_ptr(), self._stream())
        return self._metrics

    def render(self, env=0, path=None, pause=None):
        """Host-side scatter of ONE env of the batch (default env 0), the picture of flight_env_easy.py:324-343: targets as dots
        (found: orange, others black), agents as red triangles, title 'target_find:k/m', axes 0..map_size.  Copies that env's
        256 + 256 + 64 bytes to the host; not on the compute path.  `path`: save the figure there instead of showing it (headless
        boxes); `pause`: seconds for plt.pause (the reference: 3 after a full find, else 0.1).  Without matplotlib: a no-op,
        as the batched path's render has been so far."""
        raw = self.raw()
        b = int(env)
        if not 0 <= b < self.batch:
            raise IndexError(f"render: env {env} out of range for a batch of {self.batch}")
        hdr = raw["hdr"][b].cpu()
        return draw_env(raw["tgt"][b, :self.target_num].cpu().numpy(), int(hdr[_lib.H_FOUND]),
                        raw["agent"][b, :self.n_agents, :2].cpu().numpy(), int(hdr[_lib.H_TARGET_FIND]), self.target_num,
                        self.map_size, path=path, pause=pause)

    def close(self):
        pass


def draw_env(target_pos, found_mask, agent_pos, target_find, target_num, map_size, path=None, pause=None):
    """The drawing of flight_env_easy.py:324-343 from plain host data (no device, no env object): returns the matplotlib
    Axes, or None when matplotlib is missing.  Colours, marker and sizes are the reference's."""
    try:
        import matplotlib
        if path is not None:
            matplotlib.use("Agg", force=False)
        import matplotlib.pyplot as plt
    except Exception:   # noqa: BLE001  (no matplotlib, or no usable backend: rendering is optional)
        return None
    plt.cla()
    for j, (x, y) in enumerate(target_pos):
        plt.scatter(x, y, c="orange" if (found_mask >> j) & 1 else "black", s=7)
    for x, y in agent_pos:
        plt.scatter(x, y, c="red", marker="^")
    plt.title("target_find:{}/{}".format(target_find, target_num))
    plt.xlim(0, map_size)
    plt.ylim(0, map_size)
    ax = plt.gca()
    if path is not None:
        plt.savefig(path)
    else:
        plt.draw()
        plt.pause(pause if pause is not None else (3 if target_find == target_num else 0.1))
    return ax


class _SingleEnvAdapter:
    """B = 1 adapter with the reference's exact return types (BASELINE config 1 plumbing)."""

    _variant = "flight_easy"

    def __init__(self, args, circle_dict, seed=None, device="cuda"):
        self.args = args
        if seed is None:
            seed = int(np.random.randint(0, 2 ** 31 - 1))  # the reference rides numpy's global stream
        self._env = BatchedFlightEnv(args, circle_dict, batch=1, device=device, seeds=[seed], freeze_done=False,
                                     variant=self._variant)
        e = self._env
        self.map_size, self.target_num, self.n_agents = e.map_size, e.target_num, e.n_agents
        self.target_mode, self.agent_mode = e.target_mode, e.agent_mode
        self.view_range, self.time_limit = e.view_range, e.time_limit
        self.n_actions, self.state_shape, self.obs_shape = e.n_actions, e.state_shape, e.obs_shape
        self.circle_dict = e.circle_dict

    def seed(self, s):
        self._env.seed([int(s)])

    def get_env_info(self):
        info = self._env.get_env_info()
        info.pop("batch")
        return info

    def reset(self, init=False):
        self._env.reset(init=init)

    def get_avail_agent_actions(self, agent_id):
        if agent_id >= self.n_agents:
            raise Exception("Agent id out of range")
        return np.ones(self.n_actions)

    def get_obs(self):
        return self._env.get_obs()[0].to(torch.float64).cpu().numpy()

    def get_state(self):
        return self._env.get_state()[0].to(torch.float64).cpu().numpy()

    def step(self, act_list):
        if len(act_list) != self.n_agents:
            raise Exception("Act num mismatch agent")
        acts = [int(a) for a in act_list]  # python ints, np.int64 or 0-dim torch.LongTensor (agent/agent.py:36,72)
        for a in acts:
            if a not in (0, 1, 2, -1, -2, -3):
                raise IndexError("list index out of range")  # dyaw[act] in the reference
        acts = [a % 3 for a in acts]
        r, t, w = self._env.step(torch.tensor([acts], dtype=torch.int32))
        return int(r.item()), bool(t.item()), bool(w.item())

    # attributes the reference exposes (read by rollout.py:79,190,201 and by debugging code)
    def _hdr(self):
        return self._env.raw()["hdr"][0].cpu().numpy()

    @property
    def target_find(self):
        return int(self._hdr()[_lib.H_TARGET_FIND])

    @property
    def time_step(self):
        return int(self._hdr()[_lib.H_TIME_STEP])

    @property
    def total_reward(self):
        return int(self._hdr()[_lib.H_TOTAL_REWARD])

    @property
    def curr_reward(self):
        return int(self._hdr()[_lib.H_CURR_REWARD])

    @property
    def win_flag(self):
        return bool(self._hdr()[_lib.H_FLAGS] & 1)

    @property
    def out_flag(self):
        f = int(self._hdr()[_lib.H_FLAGS]) >> 8
        return [(f >> i) & 1 for i in range(self.n_agents)]

    @property
    def agent_pos(self):
        a = self._env.raw()["agent"][0, :self.n_agents, :2].cpu().numpy()
        return [[float(x), float(y)] for x, y in a]

    @property
    def agent_yaw(self):
        return [float(v) for v in self._env.raw()["agent"][0, :self.n_agents, 2].cpu().numpy()]

    @property
    def target_pos(self):
        t = self._env.raw()["tgt"][0, :self.target_num].cpu().numpy()
        return [[float(x), float(y)] for x, y in t]

    @property
    def prob_map(self):
        return self._env.raw()["prob"][0].to(torch.float64).cpu().numpy()

    def render(self):
        """flight_env_easy.py:324-343: the reference's interactive scatter (plt.draw + plt.pause) of this env."""
        self._env.render(0)

    def close(self):
        pass


class FlightSearchEnvEasy(_SingleEnvAdapter):
    """Drop-in for /root/reference/env/flight_env_easy.py:14 FlightSearchEnvEasy(args, circle_dict)."""
    _variant = "flight_easy"


class FlightSearchEnv(_SingleEnvAdapter):
    """Drop-in for /root/reference/env/flight_env.py:14 FlightSearchEnv(args, circle_dict)."""
    _variant = "flight"
