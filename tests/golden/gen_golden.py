#!/usr/bin/env python3
"""Generate golden input/output vectors from the *imported* reference env.

Runs ONLY in the build container (needs /root/reference, which never travels to
the GPU box).  It contains no reference code: it imports the reference's
`env/flight_env_easy.py` / `env/flight_env.py` / `main.load_targets`, drives them
with recorded actions under the seeding protocol of SURVEY.md §8c, and dumps what
they produced as small compressed .npz fixtures next to this script.

    python tests/golden/gen_golden.py            # regenerate every fixture

Seeding protocol (the reference never seeds NumPy; this is the harness's):
    env = Env(args, circle_dict)          # ctor's reset(init=True) eats RNG
    np.random.seed(seed)
    env.reset(init=...)                   # episode 0
    for t: env.step(actions[t])           # actions pre-drawn from RandomState(aseed)
    [env.reset(init=...) ; steps ...]     # further episodes continue the stream

Every fixture stores, per episode, the post-reset snapshot and, per step, the
inputs (actions) and everything observable afterwards, plus the uniform draws
the detection pass consumed (np.random.rand is wrapped by a recorder).
"""
import os
import sys
import types
import json

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True
REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))

import numpy as np


def import_reference():
    if not os.path.isdir(REF):
        raise SystemExit("reference tree not present; fixtures can only be regenerated in the build container")
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))       # imported, never used
    sys.modules.setdefault("pynvml", types.ModuleType("pynvml"))  # dead code in main.py
    sys.path.insert(0, REF)
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        from env.flight_env_easy import FlightSearchEnvEasy
        from env.flight_env import FlightSearchEnv
        from main import load_targets
    finally:
        os.chdir(cwd)
    return FlightSearchEnvEasy, FlightSearchEnv, load_targets


def make_args(env_name, n_agents, agent_mode, target_mode=0):
    # fields of common/arguments.py:27-34 and :233-284
    return types.SimpleNamespace(
        env=env_name, map_size=50, target_num=15, target_mode=target_mode, agent_mode=agent_mode,
        n_agents=n_agents, view_range=7, agent_velocity=1, time_limit=200, turn_limit=np.pi / 4,
        flight_height=8000, safe_dist=1, detect_prob=0.9, wrong_alarm_prob=0.1, force_dist=3,
        search_env=True, conv=(env_name == "flight"))


class DrawRecorder:
    """Wraps numpy.random.rand so each scalar draw of the env is logged."""

    def __init__(self):
        self.orig = np.random.rand
        self.log = []

    def __enter__(self):
        def rec(*a):
            v = self.orig(*a)
            if not a:
                self.log.append(float(v))
            return v
        np.random.rand = rec
        return self

    def __exit__(self, *exc):
        np.random.rand = self.orig

    def take(self):
        out, self.log = self.log, []
        return out


def yaw_index(yaw):
    return int(np.rint(float(yaw) / (np.pi / 18.0))) % 36


def snapshot(env, flight):
    d = dict(
        agent_pos=np.array([[float(p[0]), float(p[1])] for p in env.agent_pos], dtype=np.float64),
        yaw=np.array([float(y) for y in env.agent_yaw], dtype=np.float64),
        yaw_idx=np.array([yaw_index(y) for y in env.agent_yaw], dtype=np.int32),
        out_flag=np.array(env.out_flag, dtype=np.int32),
        found=np.array([1 if t.find else 0 for t in env.target_list], dtype=np.int32),
        target_find=np.int32(env.target_find),
        win=np.int32(1 if env.win_flag else 0),
        time_step=np.int32(env.time_step),
        obs=np.asarray(env.get_obs(), dtype=np.float64)[:, -4:],
        state=np.asarray(env.get_state(), dtype=np.float64),
    )
    return d


def run_trace(name, env_name, n, agent_mode, seed, aseed, episodes, target_mode=0,
              pokes=None, map_every=0, scripted=None):
    """episodes: list of dicts {init: bool, max_steps: int, stop_on_done: bool}.
    pokes: {(episode, step): {'agent_pos': [[x,y]...], 'yaw_idx': [...]}} applied BEFORE that step.
    scripted: optional {episode: actions[T,n]} overriding the random table."""
    Easy, Flight, load_targets = import_reference()
    flight = env_name == "flight"
    circle = load_targets(os.path.join(REF, "flight_targets.txt"))
    args = make_args(env_name, n, agent_mode, target_mode)
    import io, contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        env = (Flight if flight else Easy)(args, circle)
    arng = np.random.RandomState(aseed)
    out = {}
    meta = dict(name=name, env=env_name, n_agents=n, agent_mode=agent_mode, target_mode=target_mode,
                seed=seed, aseed=aseed, n_targets=15, map_size=50, view_range=7, time_limit=200,
                velocity=1, safe_dist=1, detect_prob=0.9, force_dist=3, force_factor=0.8,
                episodes=[], map_every=map_every)
    np.random.seed(seed)
    with DrawRecorder() as rec:
        for e, ep in enumerate(episodes):
            T = ep["max_steps"]
            if scripted and e in scripted:
                actions = np.asarray(scripted[e], dtype=np.int32)
                assert actions.shape == (T, n)
            else:
                actions = arng.randint(0, 3, size=(T, n)).astype(np.int32)
            env.reset(init=ep["init"])
            reset_draws = rec.take()
            pre = f"e{e}_"
            snap = snapshot(env, flight)
            out[pre + "target_pos"] = np.array(env.target_pos, dtype=np.float64)
            for k, v in snap.items():
                out[pre + "reset_" + k] = v
            out[pre + "reset_draws"] = np.array(reset_draws, dtype=np.float64)
            if flight:
                out[pre + "reset_prob_map"] = env.prob_map.copy()
            steps = dict(reward=[], terminated=[], win=[], target_find=[], found=[], agent_pos=[], yaw=[],
                         yaw_idx=[], out_flag=[], obs=[], state=[], n_draws=[], time_step=[])
            draws_flat = []
            maps, map_steps = [], []
            poke_rows = []
            used = 0
            for t in range(T):
                if pokes and (e, t) in pokes:
                    pk = pokes[(e, t)]
                    for i, (x, y) in enumerate(pk["agent_pos"]):
                        env.agent_pos[i] = [np.float64(x), np.float64(y)]
                    for i, m in enumerate(pk["yaw_idx"]):
                        env.agent_yaw[i] = m * (np.pi / 18.0)
                    poke_rows.append((t, np.array(pk["agent_pos"], dtype=np.float64),
                                      np.array(pk["yaw_idx"], dtype=np.int32)))
                r, term, win = env.step([int(a) for a in actions[t]])
                d = rec.take()
                s = snapshot(env, flight)
                steps["reward"].append(int(r))
                steps["terminated"].append(1 if term else 0)
                steps["win"].append(1 if win else 0)
                for k in ("target_find", "found", "agent_pos", "yaw", "yaw_idx", "out_flag", "obs", "state", "time_step"):
                    steps[k].append(s[k])
                steps["n_draws"].append(len(d))
                draws_flat.extend(d)
                used = t + 1
                if flight and map_every and ((t + 1) % map_every == 0 or (term and ep["stop_on_done"]) or t == T - 1):
                    maps.append(env.prob_map.copy())
                    map_steps.append(t + 1)
                if term and ep["stop_on_done"]:
                    break
            out[pre + "actions"] = actions[:used]
            for k, v in steps.items():
                dt = np.float64 if k in ("agent_pos", "yaw", "obs", "state") else np.int32
                out[pre + k] = np.array(v, dtype=dt)
            out[pre + "draws"] = np.array(draws_flat, dtype=np.float64)
            if flight and maps:
                out[pre + "prob_maps"] = np.array(maps, dtype=np.float64)
                out[pre + "prob_map_steps"] = np.array(map_steps, dtype=np.int32)
            if poke_rows:
                out[pre + "poke_steps"] = np.array([p[0] for p in poke_rows], dtype=np.int32)
                out[pre + "poke_agent_pos"] = np.array([p[1] for p in poke_rows], dtype=np.float64)
                out[pre + "poke_yaw_idx"] = np.array([p[2] for p in poke_rows], dtype=np.int32)
            meta["episodes"].append(dict(init=bool(ep["init"]), steps=used, stop_on_done=bool(ep["stop_on_done"]),
                                         sum_reward=int(np.sum(steps["reward"])), n_draws=len(draws_flat),
                                         n_reset_draws=len(reset_draws)))
    out["meta"] = np.array(json.dumps(meta))
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: " + "; ".join(
        f"ep{e}: {m['steps']} steps sumR={m['sum_reward']} draws={m['n_draws']} (+{m['n_reset_draws']} at reset)"
        for e, m in enumerate(meta["episodes"])), f"-> {os.path.getsize(path)/1024:.0f} KiB")
    return meta


EP = lambda init=True, T=200, stop=True: dict(init=init, max_steps=T, stop_on_done=stop)


def wall_script(n, T):
    """Hand-written action table: straight for a while, then hard turns, so agents hit the top wall, slide,
    turn into the side walls and come back down (all four walls get touched with AM0/AM1 starts)."""
    a = np.zeros((T, n), dtype=np.int32)
    for t in range(T):
        for i in range(n):
            if t < 45:
                a[t, i] = 0
            elif t < 45 + 9 * (1 + (i % 2)):
                a[t, i] = 1 if i % 2 == 0 else 2      # quarter/half turns, opposite senses
            elif t < 120:
                a[t, i] = 0
            elif t < 120 + 18:
                a[t, i] = 2 if i % 2 == 0 else 1
            else:
                a[t, i] = (t + i) % 3
    return a


def capture_episode(name, env_name, n, agent_mode, seed, aseed):
    """Golden episode dict from the reference's own RolloutWorker.generate_episode (common/rollout.py:22-140),
    driven by a stub Agents object that replays a pre-drawn action table (so the env is the only consumer of
    numpy's global stream, as in the seeding protocol above)."""
    Easy, Flight, load_targets = import_reference()
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        from common.rollout import RolloutWorker
    finally:
        os.chdir(cwd)
    import io, contextlib
    flight = env_name == "flight"
    circle = load_targets(os.path.join(REF, "flight_targets.txt"))
    args = make_args(env_name, n, agent_mode)
    with contextlib.redirect_stdout(io.StringIO()):
        env = (Flight if flight else Easy)(args, circle)
    info = env.get_env_info()
    args.n_actions, args.state_shape, args.obs_shape = info["n_actions"], info["state_shape"], info["obs_shape"]
    args.episode_limit = info["episode_limit"]
    args.epsilon, args.anneal_epsilon, args.min_epsilon, args.epsilon_anneal_scale = 0.0, 0.0, 0.0, "step"
    args.alg, args.evaluate_epoch = "scripted", 20
    actions = np.random.RandomState(aseed).randint(0, 3, size=(args.episode_limit, n)).astype(np.int32)

    class StubPolicy:
        def init_hidden(self, k):
            pass

    class StubAgents:
        def __init__(self):
            self.policy = StubPolicy()
            self.calls = 0

        def choose_action(self, obs, last_action, agent_num, avail_actions, epsilon, evaluate=False):
            t, self.calls = self.calls // n, self.calls + 1
            return int(actions[t][agent_num])

    with contextlib.redirect_stdout(io.StringIO()):
        worker = RolloutWorker(env, StubAgents(), args)
    np.random.seed(seed)
    episode, episode_reward, win_tag, targets_find = worker.generate_episode(1, evaluate=True)
    out = {k: np.asarray(v) for k, v in episode.items()}
    out["actions_table"] = actions
    out["meta"] = np.array(json.dumps(dict(name=name, env=env_name, n_agents=n, agent_mode=agent_mode, seed=seed,
                                           aseed=aseed, episode_reward=int(episode_reward), win_tag=bool(win_tag),
                                           targets_find=int(targets_find),
                                           steps=int((out["padded"][0, :, 0] == 0).sum()))))
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: reward={episode_reward} win={win_tag} found={targets_find} "
          f"steps={(out['padded'][0, :, 0] == 0).sum()} shapes=" +
          ",".join(f"{k}{tuple(v.shape)}" for k, v in out.items() if k not in ("meta", "actions_table")),
          f"-> {os.path.getsize(path)/1024:.0f} KiB")


def dump_target_table():
    """The dict the reference's load_targets (main.py:19-32) parses out of flight_targets.txt."""
    _, _, load_targets = import_reference()
    d = load_targets(os.path.join(REF, "flight_targets.txt"))
    with open(os.path.join(HERE, "flight_targets_parsed.json"), "w") as f:
        json.dump(d, f, indent=1)


def main():
    metas = []
    dump_target_table()
    # --- flight_easy: seeded random-action traces (known answers quoted in SURVEY.md §8c) -------------
    metas.append(run_trace("easy_n3_am0_s0_a1", "flight_easy", 3, 0, 0, 1, [EP()]))
    metas.append(run_trace("easy_n5_am0_s0_a1", "flight_easy", 5, 0, 0, 1, [EP()]))
    metas.append(run_trace("easy_n3_am2_s7_a1", "flight_easy", 3, 2, 7, 1, [EP()]))
    metas.append(run_trace("easy_n3_am3_s3_a2", "flight_easy", 3, 3, 3, 2, [EP()]))          # reset-time draws
    metas.append(run_trace("easy_n3_am1_s11_a3", "flight_easy", 3, 1, 11, 3, [EP()]))
    metas.append(run_trace("easy_n1_am0_s5_a4", "flight_easy", 1, 0, 5, 4, [EP()]))            # n == 1 branch
    metas.append(run_trace("easy_n5_am3_s9_a5", "flight_easy", 5, 3, 9, 5, [EP()]))
    # stream continuity over three episodes, and stepping PAST termination (no terminal guard, quirk Q10)
    metas.append(run_trace("easy_n3_am0_s21_a6_3ep", "flight_easy", 3, 0, 21, 6,
                           [EP(False), EP(False, 120), EP(True, 200, False)]))
    metas.append(run_trace("easy_n5_am0_s2_a7_past_done", "flight_easy", 5, 0, 2, 7, [EP(True, 200, False)]))
    # target_mode 1 (uniform random targets: 30 rand() per reset)
    metas.append(run_trace("easy_n3_am0_tm1_s4_a8", "flight_easy", 3, 0, 4, 8, [EP(), EP(False, 60)], target_mode=1))
    # scripted walls
    metas.append(run_trace("easy_n3_am0_s13_walls", "flight_easy", 3, 0, 13, 0, [EP()],
                           scripted={0: wall_script(3, 200)}))
    metas.append(run_trace("easy_n5_am1_s14_walls", "flight_easy", 5, 1, 14, 0, [EP()],
                           scripted={0: wall_script(5, 200)}))
    # --- poked states: force range, coincident agents, exact-boundary wall tests ----------------------
    pokes = {
        (0, 0): dict(agent_pos=[[10.0, 10.0], [11.5, 10.5], [30.0, 30.0]], yaw_idx=[0, 18, 9]),      # head-on, in force range
        (0, 3): dict(agent_pos=[[20.0, 20.0], [20.0, 20.0], [21.0, 20.0]], yaw_idx=[5, 5, 20]),      # coincident pair + neighbour
        (0, 6): dict(agent_pos=[[49.0, 25.0], [25.0, 49.0], [1.0, 25.0]], yaw_idx=[0, 9, 18]),       # land exactly on x=50 / y=50 / x=0
        (0, 8): dict(agent_pos=[[0.2, 49.9], [0.3, 49.8], [25.0, 0.5]], yaw_idx=[13, 14, 27]),       # two agents into the (0,50) corner
        (0, 9): dict(agent_pos=[[0.0, 50.0], [0.0, 50.0], [25.0, 0.0]], yaw_idx=[9, 9, 27]),         # coincident in the corner
        (0, 12): dict(agent_pos=[[25.0, 25.0], [27.9, 25.0], [25.0, 27.99]], yaw_idx=[35, 1, 19]),   # force_dist edge (<9 strict), yaw wrap both ways
        (0, 14): dict(agent_pos=[[40.0, 45.0], [22.0, 33.0], [26.5, 33.0]], yaw_idx=[36 % 36, 17, 19]),
    }
    poke_actions = np.array([[0, 0, 0], [1, 2, 0], [0, 0, 0], [0, 0, 0], [2, 1, 1], [0, 0, 0], [0, 0, 0], [0, 1, 2],
                             [0, 0, 0], [0, 0, 0], [1, 1, 1], [2, 2, 2], [2, 1, 0], [1, 2, 0], [0, 0, 0], [1, 1, 2],
                             [0, 2, 1], [0, 0, 0], [2, 2, 2], [1, 1, 1]], dtype=np.int32)
    metas.append(run_trace("easy_n3_am0_s17_pokes", "flight_easy", 3, 0, 17, 0, [EP(True, 20, False)],
                           pokes=pokes, scripted={0: poke_actions}))
    metas.append(run_trace("flight_n3_am0_s17_pokes", "flight", 3, 0, 17, 0, [EP(True, 20, False)],
                           pokes=pokes, scripted={0: poke_actions}, map_every=5))       # `>=` wall test differs here
    # --- flight (probability map) ----------------------------------------------------------------------
    metas.append(run_trace("flight_n3_am0_s0_a1", "flight", 3, 0, 0, 1, [EP(True, 200), EP(False, 40)], map_every=20))
    metas.append(run_trace("flight_n3_am3_s3_a2", "flight", 3, 3, 3, 2, [EP(True, 60, False)], map_every=15))
    metas.append(run_trace("flight_n5_am2_s8_a9", "flight", 5, 2, 8, 9, [EP(True, 50, False)], map_every=25))
    with open(os.path.join(HERE, "MANIFEST.json"), "w") as f:
        json.dump(metas, f, indent=1)
    capture_all_episodes()
    capture_replay_indices()     # replay_indices.json
    capture_rnn_forward()        # rnn_forward.npz
    capture_random_curves()      # random_curves.json
    # (the trained-checkpoint fixtures trained_*.npz have their own, slower script: gen_trained.py)


def capture_replay_indices():
    """FIFO storage-index rule and sampling helpers of the reference ReplayBuffer (common/replay_buffer.py:84-101,
    70-82): a sequence of store sizes -> the indices it hands out."""
    import_reference()
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        from common.replay_buffer import ReplayBuffer
    finally:
        os.chdir(cwd)
    import io, contextlib
    args = types.SimpleNamespace(n_actions=3, n_agents=3, state_shape=57, obs_shape=4, episode_limit=4, conv=False,
                                 map_size=50)
    rng = np.random.RandomState(5)
    cases = []
    for size in (7, 16, 100):
        with contextlib.redirect_stdout(io.StringIO()):
            rb = ReplayBuffer(args, size)
        incs = [int(v) for v in rng.randint(1, max(2, size // 2), size=40)]
        steps = []
        for inc in incs:
            idx = rb._get_storage_idx(inc)
            latest = None
            k = min(3, rb.current_size)
            if k > 0:
                # sample_latest returns buffer rows; recover the indices it used via a marker array
                rb.buffers["r"][:, 0, 0] = np.arange(size)
                latest = [int(v) for v in rb.sample_latest(k)["r"][:, 0, 0]]
            steps.append(dict(inc=inc, idx=[int(v) for v in np.atleast_1d(idx)], current_idx=int(rb.current_idx),
                              current_size=int(rb.current_size), latest3=latest))
        cases.append(dict(size=size, steps=steps))
    with open(os.path.join(HERE, "replay_indices.json"), "w") as f:
        json.dump(cases, f)
    print("replay_indices.json:", sum(len(c["steps"]) for c in cases), "steps")


def capture_rnn_forward():
    """Reference RNN (network/base_net.py:5-46) forward on seeded random weights and inputs, for flight_easy
    (input 4 + 3 + n) and flight (conv front end).  Weights are ours (torch.manual_seed), not the shipped models."""
    import_reference()
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        from network.base_net import RNN
    finally:
        os.chdir(cwd)
    import torch
    out = {}
    for tag, conv, n in (("easy", False, 3), ("flight", True, 3)):
        args = types.SimpleNamespace(conv=conv, map_size=50, rnn_hidden_dim=64, n_actions=3, dim_1=4, kernel_size_1=4,
                                     stride_1=2, dim_2=1, kernel_size_2=3, stride_2=1, padding_2=1, conv_out_dim=16)
        in_shape = 4 + 3 + n + (16 if conv else 0)
        torch.manual_seed(1234)
        net = RNN(in_shape, args).double()
        rows = 6
        width = (2500 if conv else 0) + 4 + 3 + n
        x = torch.rand(rows, width, dtype=torch.float64)
        h = torch.randn(rows, 64, dtype=torch.float64)
        with torch.no_grad():
            q, h2 = net(x, h)
        for k, v in net.state_dict().items():
            out[f"{tag}_w_{k}"] = v.numpy().astype(np.float32)
        out[f"{tag}_x"], out[f"{tag}_h"] = x.numpy(), h.numpy()
        out[f"{tag}_q"], out[f"{tag}_h2"] = q.numpy(), h2.numpy()
    path = os.path.join(HERE, "rnn_forward.npz")
    np.savez_compressed(path, **out)
    print("rnn_forward.npz", f"{os.path.getsize(path)/1024:.0f} KiB")


RANDOM_CURVE_CONFIGS = [(3, 0), (5, 0), (3, 3), (3, 2)]   # (n_agents, agent_mode), flight_easy, target_mode 0
RANDOM_CURVE_EPISODES = 1500
RANDOM_CURVE_INDEX = [10, 20, 40, 60, 80, 100, 150, 199]   # the reference's own print indices (runner.py:168)


def capture_random_curves():
    """Found-fraction curves of the imported reference env under an iid uniform random policy, the protocol of
    `RolloutWorker.generate_replay` (common/rollout.py:143-204: reset(init=True), res[t] = target_find / target_num after
    step t + 1, padded with 1.0 after an early termination) averaged like `collect_experiment_data` (runner.py:163-171),
    over RANDOM_CURVE_EPISODES episodes instead of the reference's 100 -- the tight statistical pin for the batched env's
    random-policy curves (the shipped average_res_529.npy files are 100-episode samples: s.e. 2-3 points).
    Deterministic: env stream np.random.seed(777000 + 10 n + agent_mode), actions from RandomState(888000 + ...)."""
    Easy, _Flight, load_targets = import_reference()
    circle = load_targets(os.path.join(REF, "flight_targets.txt"))
    import io, contextlib
    out = {"episodes": RANDOM_CURVE_EPISODES, "index": RANDOM_CURVE_INDEX, "protocol":
           "flight_easy, target_mode 0, iid uniform actions, reset(init=True) per episode, pad 1.0 after termination",
           "curves": []}
    for n, am in RANDOM_CURVE_CONFIGS:
        with contextlib.redirect_stdout(io.StringIO()):
            env = Easy(make_args("flight_easy", n, am), circle)
        seed, aseed = 777000 + 10 * n + am, 888000 + 10 * n + am
        np.random.seed(seed)
        arng = np.random.RandomState(aseed)
        T = 200
        acc = np.zeros(T)
        found_sum = 0
        for _ep in range(RANDOM_CURVE_EPISODES):
            env.reset(init=True)
            acts = arng.randint(0, 3, size=(T, n))
            res, terminated, step = [], False, 0
            while not terminated and step < T:
                _r, terminated, _w = env.step([int(a) for a in acts[step]])
                step += 1
                res.append(env.target_find / 15)
            res += [1.0] * (T - len(res))
            acc += np.array(res)
            found_sum += env.target_find
        curve = acc / RANDOM_CURVE_EPISODES * 100.0
        shipped = np.load(os.path.join(REF, "result", f"flight_easy_Seed0_random_{n}a15t(AM{am}TM0)", "average_res_529.npy"))
        out["curves"].append({"n_agents": n, "agent_mode": am, "seed": seed, "aseed": aseed,
                              "curve": [round(float(v), 6) for v in curve],
                              "at_index": [round(float(curve[i]), 6) for i in RANDOM_CURVE_INDEX],
                              "shipped_100_episodes_at_index": [round(float(shipped[i]), 6) for i in RANDOM_CURVE_INDEX],
                              "mean_targets_found": found_sum / RANDOM_CURVE_EPISODES})
        print(f"random curve n={n} AM{am}:", np.round(curve[RANDOM_CURVE_INDEX], 2), "shipped",
              np.round(shipped[RANDOM_CURVE_INDEX], 2), flush=True)
    with open(os.path.join(HERE, "random_curves.json"), "w") as f:
        json.dump(out, f)
    print("random_curves.json written")


def capture_all_episodes():
    # episode dicts of common/rollout.py (the "next" row f1 of SURVEY.md section 8)
    capture_episode("episode_easy_n3_am0_s0_a1", "flight_easy", 3, 0, 0, 1)      # 200 steps, time-limit termination
    capture_episode("episode_easy_n5_am0_s0_a1", "flight_easy", 5, 0, 0, 1)      # 76 steps, win + zero padding
    capture_episode("episode_flight_n3_am3_s3_a2", "flight", 3, 3, 3, 2)         # conv obs (2504 wide)


if __name__ == "__main__":
    main()
