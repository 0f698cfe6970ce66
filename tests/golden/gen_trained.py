#!/usr/bin/env python3
"""Fixtures for policy-in-the-loop parity with the reference's SHIPPED checkpoints.

Runs ONLY in the build container (needs /root/reference).  Contains no reference code: it torch.load()s the shipped
`model/<run>/<N>_rnn_net_params.pkl` state_dicts (weights are data), instantiates the reference's own
`network.base_net.RNN` and `env.flight_env_easy.FlightSearchEnvEasy` / `env.flight_env.FlightSearchEnv`, and replays
the loop of `RolloutWorker.generate_replay` (common/rollout.py:143-209: reset(init=True), epsilon = 0, evaluate = True,
per agent inputs = obs ++ one-hot(last action) ++ one-hot(agent id) as agent/agent.py:41-52, action = argmax q) for a
number of seeded episodes, exactly what `Runner.collect_experiment_data` (runner.py:139-171) averages into the
`result/<run>/average_res_<N>.npy` files.  Stored per run in tests/golden/trained_<tag>.npz:

    w_<param>        the checkpoint's tensors (float32)
    ref_curve        percent of targets found by step t, mean over `episodes` replays of the reference loop  [200]
    ref_reward, ref_found, ref_steps   per-episode sums, for the record
    shipped_curve    the result file the reference ships for this run (checkpoint number in `shipped_num`; for some of
                     the runs the shipped curve belongs to an EARLIER checkpoint than the shipped weights)
    traj_*           EXACT trajectories: the first TRAJ episodes are each started from their own np.random.seed value
                     (traj_seeds), and their per-step actions, rewards, found counts and the smallest top-1 / top-2 gap of
                     the network outputs over the agents (traj_qgap) are recorded, so that the HIP closed loop can be compared
                     step for step -- not only statistically -- wherever the argmax is not a near-tie

DOP runs keep their policy in `<N>_actor_net_params.pkl`: the actor IS the same RNN (policy/dop.py:40) and the greedy
choice the same argmax (agent/agent.py:61-62, 70-73); REINFORCE goes through the softmax rule, which for epsilon = 0 and
evaluate is argmax(prob) (agent.py:92-93).

    python tests/golden/gen_trained.py [tag ...]
"""
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_golden import REF, import_reference, make_args  # noqa: E402

RUNS = {
    # tag: (run directory, env, n_agents, agent_mode, checkpoint, shipped result number (None: none shipped), episodes)
    "easy3_qmix": ("flight_easy_Seed22322107_qmix_3a15t(AM0TM0)", "flight_easy", 3, 0, 121, 60, 100),
    "easy5_qmix": ("flight_easy_Seed59818301_qmix_5a15t(AM0TM0)", "flight_easy", 5, 0, 91, 60, 100),
    "easy3_reinforce": ("flight_easy_Seed18818508_reinforce_3a15t(AM0TM0)", "flight_easy", 3, 0, 198, 198, 100),
    "flight3_qmix": ("flight_Seed74853802_qmix_3a15t(AM0TM0)", "flight", 3, 0, 70, 70, 30),
    # round 3: the DOP actors north_star names, the start modes with reset-time draws (AM2 / AM3), the other flight teams
    "easy3_dop": ("flight_easy_Seed10525772_dop_3a15t(AM0TM0)", "flight_easy", 3, 0, 68, 68, 100),
    "easy5_dop": ("flight_easy_Seed75130173_dop_5a15t(AM0TM0)", "flight_easy", 5, 0, 51, 25, 100),
    "easy3_qmix_am3": ("flight_easy_Seed94841538_qmix_3a15t(AM3TM0)", "flight_easy", 3, 3, 115, 115, 100),
    "easy3_qmix_am2": ("flight_easy_Seed25913373_qmix_3a15t(AM2TM0)", "flight_easy", 3, 2, 108, 108, 100),
    "flight1_qmix": ("flight_Seed90192445_qmix_1a15t(AM0TM0)", "flight", 1, 0, 145, 145, 30),
    "flight5_qmix": ("flight_Seed94367982_qmix_5a15t(AM0TM0)", "flight", 5, 0, 36, 36, 20),
    "flight3_reinforce": ("flight_Seed12339614_reinforce_3a15t(AM0TM0)", "flight", 3, 0, 24, None, 20),
}
TRAJ = 4   # exact trajectories recorded per run


def capture(tag):
    import torch
    run, env_name, n, am, num, shipped_num, episodes = RUNS[tag]
    Easy, Flight, load_targets = import_reference()
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        from network.base_net import RNN
        circle = load_targets("flight_targets.txt") if os.path.exists("flight_targets.txt") else None
    finally:
        os.chdir(cwd)
    if circle is None:
        raise SystemExit("flight_targets.txt not found in the reference tree")
    conv = env_name == "flight"
    nargs = types.SimpleNamespace(conv=conv, map_size=50, rnn_hidden_dim=64, n_actions=3, dim_1=4, kernel_size_1=4,
                                  stride_1=2, dim_2=1, kernel_size_2=3, stride_2=1, padding_2=1, conv_out_dim=16)
    kind = "actor" if "_dop_" in run else "rnn"   # policy/dop.py:185-188 saves the actor RNN under this name
    sd = torch.load(os.path.join(REF, "model", run, f"{num}_{kind}_net_params.pkl"), map_location="cpu", weights_only=True)
    net = RNN(4 + 3 + n + (16 if conv else 0), nargs)
    net.load_state_dict(sd)
    net.eval()
    env = (Flight if conv else Easy)(make_args(env_name, n, am), circle)
    np.random.seed(20240000 + num)
    T = 200
    curves, rewards, founds, steps = [], [], [], []
    eye = np.eye(n)
    traj_seeds = [(20240000 + 7919 * num + 104729 * (k + 1)) % (1 << 31) for k in range(TRAJ)]
    traj_actions = np.full((TRAJ, T, n), -1, dtype=np.int32)
    traj_rewards = np.zeros((TRAJ, T), dtype=np.int32)
    traj_found = np.zeros((TRAJ, T), dtype=np.int32)
    traj_qgap = np.zeros((TRAJ, T), dtype=np.float32)
    traj_len = np.zeros(TRAJ, dtype=np.int32)
    with torch.no_grad():
        for ep in range(episodes):
            if ep < TRAJ:
                np.random.seed(traj_seeds[ep])   # its own stream start: the HIP env can be seeded identically
            env.reset(init=True)
            hidden = torch.zeros(n, 64)
            last = np.zeros((n, 3))
            res, terminated, step, total = [], False, 0, 0
            while not terminated and step < T:
                obs = env.get_obs()
                actions = []
                gap = np.inf
                for i in range(n):
                    x = np.hstack((obs[i], last[i], eye[i]))
                    q, h = net(torch.tensor(x, dtype=torch.float32).unsqueeze(0), hidden[i:i + 1])
                    hidden[i] = h[0]
                    a = int(torch.argmax(q))
                    top2 = torch.topk(q.reshape(-1), 2).values
                    gap = min(gap, float(top2[0] - top2[1]))
                    actions.append(a)
                    last[i] = 0.0
                    last[i, a] = 1.0
                reward, terminated, _info = env.step(actions)
                if ep < TRAJ:
                    traj_actions[ep, step] = actions
                    traj_rewards[ep, step] = reward
                    traj_found[ep, step] = env.target_find
                    traj_qgap[ep, step] = gap
                    traj_len[ep] = step + 1
                total += reward
                step += 1
                res.append(env.target_find / 15)
            res += [1.0] * (T - len(res))
            curves.append(res)
            rewards.append(total)
            founds.append(env.target_find)
            steps.append(step)
            if ep % 10 == 9:
                print(f"  {tag}: {ep + 1}/{episodes} episodes", flush=True)
    out = {f"w_{k}": v.numpy().astype(np.float32) for k, v in sd.items()}
    out["ref_curve"] = np.mean(np.array(curves), axis=0) * 100.0
    out["ref_reward"], out["ref_found"], out["ref_steps"] = np.array(rewards), np.array(founds), np.array(steps)
    out["episodes"], out["checkpoint"], out["shipped_num"] = episodes, num, shipped_num
    out["n_agents"], out["agent_mode"] = n, am
    out["shipped_num"] = -1 if shipped_num is None else shipped_num
    out["shipped_curve"] = (np.load(os.path.join(REF, "result", run, f"average_res_{shipped_num}.npy"))
                            if shipped_num is not None else np.full(T, np.nan))
    out["traj_seeds"] = np.array(traj_seeds, dtype=np.int64)
    out["traj_actions"], out["traj_rewards"], out["traj_found"] = traj_actions, traj_rewards, traj_found
    out["traj_qgap"], out["traj_len"] = traj_qgap, traj_len
    path = os.path.join(HERE, f"trained_{tag}.npz")
    np.savez_compressed(path, **out)
    idx = [10, 20, 40, 60, 80, 100, 150, 199]
    print(tag, f"{os.path.getsize(path) / 1024:.0f} KiB", "ref", np.round(out["ref_curve"][idx], 2), "shipped",
          np.round(out["shipped_curve"][idx], 2), "mean found", np.mean(founds))


if __name__ == "__main__":
    for tag in (sys.argv[1:] or list(RUNS)):
        capture(tag)
