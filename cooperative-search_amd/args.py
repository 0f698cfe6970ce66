"""Environment constants: restatement of the env-related setters of /root/reference/common/arguments.py.

`get_flight_easy_args` (:269-284) and `get_flight_args` (:233-267) mutate an argparse-style namespace exactly
like the reference's (same field names and values); `make_env_args` builds a namespace with the env fields of
`get_common_args` (:27-34) for callers that have no argparse namespace of their own.
"""
import math
import types


def _flight_common(args):
    args.agent_velocity = 1
    args.time_limit = 200
    args.turn_limit = math.pi / 4     # stored by the reference, never used
    args.flight_height = 8000         # never used
    args.safe_dist = 1
    args.detect_prob = 0.9
    args.wrong_alarm_prob = 0.1       # never used
    args.force_dist = 3
    args.search_env = True
    return args


def get_flight_easy_args(args):
    _flight_common(args)
    args.conv = False
    return args


def get_flight_args(args):
    _flight_common(args)
    args.conv = True
    args.dim_1, args.kernel_size_1, args.stride_1 = 4, 4, 2
    args.dim_2, args.kernel_size_2, args.stride_2, args.padding_2 = 1, 3, 1, 1
    args.conv_out_dim = 16
    return args


def make_env_args(env="flight_easy", n_agents=3, agent_mode=0, target_mode=0, map_size=50, target_num=15,
                  view_range=7):
    args = types.SimpleNamespace(env=env, map_size=map_size, target_num=target_num, target_mode=target_mode,
                                 agent_mode=agent_mode, n_agents=n_agents, view_range=view_range)
    return get_flight_args(args) if env == "flight" else get_flight_easy_args(args)


def apply_env_info(args, env):
    """main.py:113-118: copy get_env_info() into the namespace; plus the algorithm-side fields the agent network and the
    replay buffer read (common/arguments.py:37-39 last_action / reuse_network, get_mixer_args :58 rnn_hidden_dim)."""
    info = env.get_env_info()
    args.n_actions = info["n_actions"]
    args.state_shape = info["state_shape"]
    args.obs_shape = info["obs_shape"]
    args.episode_limit = info["episode_limit"]
    for k, v in (("last_action", True), ("reuse_network", True), ("rnn_hidden_dim", 64)):
        if not hasattr(args, k):
            setattr(args, k, v)
    return args
