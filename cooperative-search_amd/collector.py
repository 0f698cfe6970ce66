"""Batched episode collection and evaluation: the callers of the env path.

`EpisodeCollector.generate_episodes` is the batched counterpart of `RolloutWorker.generate_episode`
(/root/reference/common/rollout.py:22-140): same 11 keys, same padding semantics (after an episode terminates its
remaining time steps are all-zero with padded = 1 and terminated = 1, rollout.py:105-116), one leading batch
dimension of B episodes instead of 1.  `evaluate` is `Runner.evaluate` (runner.py:86-96) and
`collect_experiment_data` the found-fraction curve of runner.py:139-171 / rollout.py:143-204, both reduced over
ranks with the all-gather of dist.py.

The env arithmetic stays in the HIP kernels; the episode batch itself (masking, padding, the [T, B] -> [B, T]
transposition) is assembled by `cs_store_episodes` (csrc/episodes.hip) in one pass, either into fresh tensors or
straight into the ring of a `DeviceReplayBuffer`.
"""
import ctypes as C
import os

import torch

from . import _lib
from . import dist as _dist
from .replay import KEYS


def assemble_episodes(o, s, u, r, term, n_actions, out=None, slots=None):
    """Step-major tables -> the reference's 11-key episode batch (rollout.py:66-76,105-132) with one kernel pass.
    o [T+1, B, n, w], s [T+1, B, S], u int64 [T, B, n], r [T, B], term bool/uint8 [T, B] (device, contiguous).
    out: dict of float32 [slots, T, ...] destinations (default: fresh [B, T, ...] tensors); slots: int64 [B] destination
    slot of each env's episode (default: its own index)."""
    T, B, n = u.shape
    w, S, A = o.shape[-1], s.shape[-1], int(n_actions)
    dev = o.device
    if out is None:
        shapes = {"o": (B, T, n, w), "u": (B, T, n, 1), "s": (B, T, S), "r": (B, T, 1), "o_next": (B, T, n, w),
                  "s_next": (B, T, S), "avail_u": (B, T, n, A), "avail_u_next": (B, T, n, A), "u_onehot": (B, T, n, A),
                  "padded": (B, T, 1), "terminated": (B, T, 1)}
        out = {k: torch.empty(shapes[k], dtype=torch.float32, device=dev) for k in KEYS}
    for t in (o, s, u, r, term) + tuple(out[k] for k in KEYS):
        if not t.is_contiguous() or t.device != dev:
            raise ValueError("assemble_episodes: tensors must be contiguous and on one device")
    if any(out[k].dtype != torch.float32 or out[k].shape[1] != T for k in KEYS) or u.dtype != torch.int64:
        raise ValueError("assemble_episodes: destinations must be float32 [slots, T, ...] and u int64")
    ops = _lib.pick_binding(None)[1]
    if ops is not None:   # torch.ops.coopsearch.store_episodes: checks in C++, torch's stream
        ops.store_episodes(o, s, u, r, term.view(torch.uint8), slots, A, [out[k] for k in KEYS])
        return out
    L = _lib.load()
    eo = _lib.CsEpisodeOut(**{k: out[k].data_ptr() for k in KEYS})
    with torch.cuda.device(dev):   # the launch goes to the process's current device
        rc = L.cs_store_episodes(B, T, n, A, w, S, o.data_ptr(), s.data_ptr(), u.data_ptr(), r.data_ptr(),
                                 term.view(torch.uint8).data_ptr(), slots.data_ptr() if slots is not None else None,
                                 C.byref(eo), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    if rc != 0:
        raise _lib.CoopSearchError(L.cs_episodes_last_error().decode())
    return out


def assemble_episodes_torch(o, s, u, r, term, n_actions):
    """The same batch with stock torch ops (the definition the kernel is tested against)."""
    T, B, n = u.shape
    A = int(n_actions)
    done_before = torch.zeros(T, B, dtype=torch.bool, device=o.device)
    done_before[1:] = term[:-1].to(torch.bool)
    real = ~done_before  # a step is real if the env had not terminated before it
    rf = real.to(torch.float32)

    def bt(x):  # [T, B, ...] -> [B, T, ...]
        return x.transpose(0, 1).contiguous()

    m4 = rf[:, :, None, None]
    onehot = torch.nn.functional.one_hot(u, A).to(torch.float32)
    ones = torch.ones(T, B, n, A, dtype=torch.float32, device=o.device)
    return dict(
        o=bt(o[:-1] * m4), s=bt(s[:-1] * rf[:, :, None]), u=bt((u.to(torch.float32) * rf[:, :, None])[..., None]),
        r=bt((r * rf)[..., None]), avail_u=bt(ones * m4), o_next=bt(o[1:] * m4), s_next=bt(s[1:] * rf[:, :, None]),
        avail_u_next=bt(ones * m4), u_onehot=bt(onehot * m4), padded=bt((1.0 - rf)[..., None]),
        terminated=bt(torch.where(real, term.to(torch.float32), torch.ones_like(rf))[..., None]))


class EpsilonSchedule:
    """The exploration schedule a RolloutWorker carries (common/rollout.py:17-19, 35-41, 75-76, 133-135; defaults of
    common/arguments.py:78-82: epsilon 1 -> 0.05 over 10000 steps, scale 'step').  The reference has one worker and one scalar;
    a batch of B envs is B workers, each with its OWN epsilon (float64 [B] on the device), annealed by its own executed steps
    with the reference's rule `epsilon = epsilon - anneal_epsilon if epsilon > min_epsilon else epsilon`:
        'step'     after every executed env step -- on the device, inside the fused rollout (cs_epsilon) or by
                   cs_epsilon_step in a per-step loop;
        'episode'  once before every episode (rollout.py:36-38);
        'epoch'    once before the episode with episode_num == 0 (:39-41).
    The values persist across generate_episodes calls (rollout.py:133-135)."""

    def __init__(self, args, batch, device="cuda"):
        self.anneal = float(args.anneal_epsilon)
        self.min_epsilon = float(args.min_epsilon)
        self.scale = getattr(args, "epsilon_anneal_scale", "step")
        if self.scale not in ("step", "episode", "epoch"):
            raise ValueError("epsilon_anneal_scale must be 'step', 'episode' or 'epoch'")
        self.values = torch.full((int(batch),), float(args.epsilon), dtype=torch.float64, device=device)

    @property
    def per_step(self):
        return self.scale == "step"

    def anneal_once(self):
        v = self.values
        self.values = torch.where(v > self.min_epsilon, v - self.anneal, v)   # IEEE double subtraction, like the reference's floats
        return self.values

    def begin_episode(self, episode_num=None):
        """What generate_episode does to epsilon before its loop (rollout.py:36-41)."""
        if self.scale == "episode" or (self.scale == "epoch" and episode_num == 0):
            self.anneal_once()
        return self.values


class EpisodeCollector:
    def __init__(self, env, schedule=None):
        """schedule (an `EpsilonSchedule`): the exploration schedule this collector carries across generate_episodes calls,
        as RolloutWorker carries self.epsilon; without one, `epsilon` of each call is used as a constant."""
        self.env = env
        self.schedule = schedule

    def generate_episodes(self, policy=None, actions=None, init=False, agents=None, epsilon=0.0, evaluate=True,
                          one_launch=True, into=None, episode_num=None, eps_trace=None):
        """One episode per env.  Either `actions` (open-loop table, int [T, B, n]; flight_easy runs it as ONE fused
        rollout launch), `policy(obs[B,n,obs], state[B,S], last_onehot[B,n,A], t) -> int actions [B, n]`, or `agents`
        (a `FusedAgents`: flight_easy with n <= 5 and one_launch: the WHOLE episode -- T x (network forward, env step)
        -- is one kernel launch, `env.rollout_policy`; otherwise two launches per step; both write straight into the
        [T, ...] episode tables, no copies).
        `into` (a `DeviceReplayBuffer`): the batch is written straight into the buffer's next B ring slots
        (store_episode without the intermediate copy) and the returned episode dict is None.
        With a schedule (see __init__) and evaluate=False the envs explore with their own, annealing epsilon
        (episode_num: the reference's argument, only the 'epoch' scale looks at it; eps_trace: float64 [T, B] that receives the
        epsilon of every step's selection).
        Returns (episode dict of float32 [B, T, ...] tensors, episode_reward[B], win_tag[B] bool, targets_find[B])."""
        env = self.env
        sched = self.schedule if (self.schedule is not None and not evaluate and agents is not None) else None
        if sched is not None:
            sched.begin_episode(episode_num)
        kw = {} if sched is None else dict(eps_env=sched.values, anneal=sched.anneal, min_epsilon=sched.min_epsilon,
                                           per_step=sched.per_step, eps_trace=eps_trace)
        B, n, T, A = env.batch, env.n_agents, env.time_limit, env.n_actions
        dev = env.device
        saved = (env.freeze_done, env.auto_reset)
        env.freeze_done, env.auto_reset = True, False  # finished envs must stay put; their steps become padding
        try:
            env.reset(init=init)  # rollout.py:26
            obs_w, S = env.obs_width, env.state_shape
            o = torch.empty(T + 1, B, n, obs_w, dtype=torch.float32, device=dev)
            s = torch.empty(T + 1, B, S, dtype=torch.float32, device=dev)
            u = torch.empty(T, B, n, dtype=torch.int64, device=dev)
            r = torch.empty(T, B, dtype=torch.float32, device=dev)
            term = torch.empty(T, B, dtype=torch.bool, device=dev)
            o[0].copy_(env.get_obs())
            s[0].copy_(env.get_state())
            if actions is not None and (not env.flight or env.batch * env.time_limit * env.n_agents * env.obs_width * 4 < (8 << 30)):
                acts = torch.as_tensor(actions, device=dev)
                out = env.rollout(acts)
                o[1:].copy_(out["obs"])
                s[1:].copy_(out["state"])
                u.copy_(acts.to(torch.int64))
                r.copy_(out["reward"])
                term.copy_(out["terminated"])
            elif agents is not None and one_launch and not env.flight and n <= 5:
                agents.init_hidden()
                env.rollout_policy(agents, T, epsilon, evaluate,
                                   out=dict(actions=u, reward=r, terminated=term, obs=o[1:], state=s[1:]), **kw)
            elif agents is not None:
                agents.init_hidden()
                none = torch.full((B, n), -1, dtype=torch.int64, device=dev)
                for t in range(T):
                    agents.choose_action(o[t], epsilon, evaluate, last=u[t - 1] if t else none, out=u[t],
                                         eps_env=sched.values if sched is not None else None)
                    if sched is not None and (sched.per_step or eps_trace is not None):   # rollout.py:75-76, before the step it follows
                        env.epsilon_step(sched.values, sched.anneal if sched.per_step else 0.0,
                                         sched.min_epsilon if sched.per_step else 1e300,
                                         eps_trace[t] if eps_trace is not None else None)
                    env.step(u[t], out=dict(reward=r[t], terminated=term[t], obs=o[t + 1], state=s[t + 1]))
                env.refresh()
            else:
                last = torch.zeros(B, n, A, dtype=torch.float32, device=dev)
                for t in range(T):
                    a = actions[t] if actions is not None else policy(o[t], s[t], last, t)
                    u[t].copy_(torch.as_tensor(a, device=dev))
                    env.step(u[t], out=dict(reward=r[t], terminated=term[t], obs=o[t + 1], state=s[t + 1]))
                    last = torch.nn.functional.one_hot(u[t], A).to(torch.float32)
                env.refresh()
            if into is not None:
                slots = torch.as_tensor(into._get_storage_idx(inc=B), device=dev)
                assemble_episodes(o, s, u, r, term, A, out=into.buffers, slots=slots)
                episode = None
            else:
                episode = assemble_episodes(o, s, u, r, term, A)
            episode_reward = env.total_reward.to(torch.float32).clone()
            win_tag = env.win_flag.clone()  # terminated and win_flag (rollout.py:64): a win always terminates
            targets_find = env.target_find.clone()
            return episode, episode_reward, win_tag, targets_find
        finally:
            env.freeze_done, env.auto_reset = saved


def _run_episodes(env, policy, init):
    """One batch of episodes with frozen termination; yields (t, target_find) after every step."""
    B, n, T, A = env.batch, env.n_agents, env.time_limit, env.n_actions
    saved = (env.freeze_done, env.auto_reset)
    env.freeze_done, env.auto_reset = True, False
    try:
        env.reset(init=init)
        last = torch.zeros(B, n, A, dtype=torch.float32, device=env.device)
        for t in range(T):
            a = torch.as_tensor(policy(env.get_obs(), env.get_state(), last, t), device=env.device).to(torch.int64)
            env.step(a)
            last = torch.nn.functional.one_hot(a, A).to(torch.float32)
            yield t, env.target_find
    finally:
        env.freeze_done, env.auto_reset = saved


def random_policy(generator=None):
    """The reference's alg='random' (agent/agent.py:34-36): uniform over the available actions."""
    def policy(obs, state, last, t):
        return torch.randint(0, 3, obs.shape[:2], device=obs.device, generator=generator)
    return policy


def evaluate(env, policy, batches=1):
    """Runner.evaluate (runner.py:86-96): mean win_tag, episode_reward, targets_find over batches * B * world
    evaluation episodes (reset() without init, like generate_episode).  One all-gather per call."""
    part = torch.zeros(4, dtype=torch.float64, device=env.device)
    for _ in range(batches):
        for _t, _tf in _run_episodes(env, policy, init=False):
            pass
        part += env.metric_partials()
    m = _dist.reduce_metrics(part)
    return m["win_rate"], m["episode_reward"], m["targets_find"]


def collect_experiment_data(env, policy, batches=1, num=None, result_path=None, return_stats=False):
    """Runner.collect_experiment_data (runner.py:139-171): percent of targets found by step t (float64[episode_limit]),
    reset(init=True) per episode as generate_replay does (rollout.py:143-204), averaged over batches * B * world episodes,
    plus the three means the reference prints -- targets found, episode reward, episode length (runner.py:163-165).
    num + result_path: rank 0 writes `average_res_<num>.npy` into result_path, the file the reference saves (:171).
    Returns the curve, or (curve, {'targets_find', 'episode_reward', 'steps', 'episodes'}) with return_stats."""
    curve = _dist.FoundCurve(env.time_limit, env.target_num, env.device)
    part = torch.zeros(4, dtype=torch.float64, device=env.device)   # sum target_find, sum reward, sum steps, episodes
    for _ in range(batches):
        for t, tf in _run_episodes(env, policy, init=True):
            curve.add_step(t, tf)
        curve.end_episodes(env.batch)
        # (a finished env is frozen: its counters are those of its last executed step, i.e. generate_replay's return values)
        part += torch.stack([env.target_find.to(torch.float64).sum(), env.total_reward.to(torch.float64).sum(),
                             env.time_step.to(torch.float64).sum(),
                             torch.tensor(float(env.batch), dtype=torch.float64, device=env.device)])
    res = curve.result()
    tot = _dist.all_gather_sum(part)
    n = float(tot[3].item())
    stats = {"targets_find": float(tot[0].item()) / n, "episode_reward": float(tot[1].item()) / n,
             "steps": float(tot[2].item()) / n, "episodes": int(n)}
    if num is not None and result_path is not None and _dist.rank() == 0:
        import numpy as np
        os.makedirs(result_path, exist_ok=True)
        np.save(os.path.join(result_path, "average_res_{}".format(num)), res)   # runner.py:171 (np.save appends .npy)
    return (res, stats) if return_stats else res
