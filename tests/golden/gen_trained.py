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
    shipped_curve    the result file the reference ships for this run (checkpoint number in `shipped_num`; for two of
                     the runs the shipped curve belongs to an EARLIER checkpoint than the shipped weights)

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
    # tag: (run directory, env, n_agents, agent_mode, checkpoint, shipped result number, episodes)
    "easy3_qmix": ("flight_easy_Seed22322107_qmix_3a15t(AM0TM0)", "flight_easy", 3, 0, 121, 60, 100),
    "easy5_qmix": ("flight_easy_Seed59818301_qmix_5a15t(AM0TM0)", "flight_easy", 5, 0, 91, 60, 100),
    "easy3_reinforce": ("flight_easy_Seed18818508_reinforce_3a15t(AM0TM0)", "flight_easy", 3, 0, 198, 198, 100),
    "flight3_qmix": ("flight_Seed74853802_qmix_3a15t(AM0TM0)", "flight", 3, 0, 70, 70, 30),
}


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
    sd = torch.load(os.path.join(REF, "model", run, f"{num}_rnn_net_params.pkl"), map_location="cpu", weights_only=True)
    net = RNN(4 + 3 + n + (16 if conv else 0), nargs)
    net.load_state_dict(sd)
    net.eval()
    env = (Flight if conv else Easy)(make_args(env_name, n, am), circle)
    np.random.seed(20240000 + num)
    T = 200
    curves, rewards, founds, steps = [], [], [], []
    eye = np.eye(n)
    with torch.no_grad():
        for ep in range(episodes):
            env.reset(init=True)
            hidden = torch.zeros(n, 64)
            last = np.zeros((n, 3))
            res, terminated, step, total = [], False, 0, 0
            while not terminated and step < T:
                obs = env.get_obs()
                actions = []
                for i in range(n):
                    x = np.hstack((obs[i], last[i], eye[i]))
                    q, h = net(torch.tensor(x, dtype=torch.float32).unsqueeze(0), hidden[i:i + 1])
                    hidden[i] = h[0]
                    a = int(torch.argmax(q))
                    actions.append(a)
                    last[i] = 0.0
                    last[i, a] = 1.0
                reward, terminated, _info = env.step(actions)
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
    out["shipped_curve"] = np.load(os.path.join(REF, "result", run, f"average_res_{shipped_num}.npy"))
    path = os.path.join(HERE, f"trained_{tag}.npz")
    np.savez_compressed(path, **out)
    idx = [10, 20, 40, 60, 80, 100, 150, 199]
    print(tag, f"{os.path.getsize(path) / 1024:.0f} KiB", "ref", np.round(out["ref_curve"][idx], 2), "shipped",
          np.round(out["shipped_curve"][idx], 2), "mean found", np.mean(founds))


if __name__ == "__main__":
    for tag in (sys.argv[1:] or list(RUNS)):
        capture(tag)
