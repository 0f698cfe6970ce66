#!/usr/bin/env python3
"""Fixture for the exploration schedule: the epsilon the reference's OWN RolloutWorker hands to
Agents.choose_action at every env step, over several consecutive training episodes.

Runs ONLY in the build container (needs /root/reference); contains no reference code.  The reference's argument setters
(common/arguments.py: get_mixer_args -> epsilon = 1, min_epsilon = 0.05, anneal_epsilon = 0.95 / 10000,
epsilon_anneal_scale = 'step'), its Agents with alg = 'qmix' (policy/qmix.py, random-init networks on the CPU), its
FlightSearchEnvEasy and its RolloutWorker.generate_episode (common/rollout.py:22-140) run as they are; choose_action is
wrapped only to RECORD the epsilon argument of every call.  Cases:
    default      the shipped defaults, 6 episodes (epsilon persists across episodes, rollout.py:133-135)
    fast         anneal_epsilon = 0.95 / 150: the floor min_epsilon is crossed inside the run (the `epsilon > min_epsilon`
                 rule leaves the value one step BELOW... whatever the reference's float arithmetic leaves: recorded)
    episode      epsilon_anneal_scale = 'episode': one anneal before each episode (rollout.py:36-38)
    epoch        epsilon_anneal_scale = 'epoch': an anneal only when episode_num == 0 (rollout.py:39-41); 2 epochs x 3 episodes
Stored in tests/golden/epsilon_schedule.json: per case the settings, the per-episode step counts and the epsilon of every
step as a Python float repr (exact doubles).

    python tests/golden/gen_epsilon.py
"""
import contextlib
import io
import json
import os
import sys

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_golden import REF, import_reference  # noqa: E402


def run_case(scale, anneal_steps, episodes, epochs=1, seed=123):
    Easy, Flight, load_targets = import_reference()
    cwd = os.getcwd()
    os.chdir(REF)
    argv = sys.argv
    try:
        sys.argv = ["main.py"]
        from common.arguments import get_common_args, get_mixer_args, get_flight_easy_args
        from common.rollout import RolloutWorker
        from agent.agent import Agents
        args = get_common_args()
        args.alg, args.cuda, args.show, args.load_model, args.seed_idx = "qmix", False, True, False, 0
        args = get_flight_easy_args(get_mixer_args(args))
        circle = load_targets("flight_targets.txt")
    finally:
        sys.argv = argv
        os.chdir(cwd)
    args.seed = 19990227
    args.epsilon_anneal_scale = scale
    args.anneal_epsilon = (args.epsilon - args.min_epsilon) / anneal_steps
    with contextlib.redirect_stdout(io.StringIO()):
        env = Easy(args, circle)
        info = env.get_env_info()
        args.n_actions, args.state_shape, args.obs_shape = info["n_actions"], info["state_shape"], info["obs_shape"]
        args.episode_limit = info["episode_limit"]
        agents = Agents(env, args)
        worker = RolloutWorker(env, agents, args)
    used = []
    inner = agents.choose_action

    def recording(obs, last_action, agent_num, avail_actions, epsilon, evaluate=False):
        if agent_num == 0:
            used.append(float(epsilon))
        return inner(obs, last_action, agent_num, avail_actions, epsilon, evaluate)

    agents.choose_action = recording
    np.random.seed(seed)
    steps, carried = [], []
    with contextlib.redirect_stdout(io.StringIO()):
        for _epoch in range(epochs):
            for k in range(episodes):
                n0 = len(used)
                worker.generate_episode(k)
                steps.append(len(used) - n0)
                carried.append(float(worker.epsilon))
    return {"scale": scale, "epsilon0": float(args.epsilon), "anneal_epsilon": repr(float(args.anneal_epsilon)),
            "min_epsilon": repr(float(args.min_epsilon)), "episodes_per_epoch": episodes, "epochs": epochs,
            "steps": steps, "carried_after_episode": [repr(v) for v in carried], "used": [repr(v) for v in used]}


def main():
    out = {"default": run_case("step", 10000, 6), "fast": run_case("step", 150, 4),
           "episode": run_case("episode", 10, 14), "epoch": run_case("epoch", 10, 3, epochs=2)}
    for k, v in out.items():
        print(k, "episodes", v["steps"], "first", v["used"][0], "last", v["used"][-1], "carried", v["carried_after_episode"][-1])
    with open(os.path.join(HERE, "epsilon_schedule.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("wrote", os.path.join(HERE, "epsilon_schedule.json"))


if __name__ == "__main__":
    main()
