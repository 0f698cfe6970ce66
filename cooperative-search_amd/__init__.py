"""cooperative-search_amd: MI355X-native batched flight_easy / flight environment path.

Scope: the environment reset/step/reward/obs/state path of
WZN1ng/Cooperative-Search, advanced by hand-written gfx950 HIP kernels behind the C ABI of
include/coopsearch.h.  Nothing here falls back to a CPU implementation: importing works anywhere, but
constructing an environment without the built HIP library and a GPU raises.
"""
from .targets import load_targets, DEFAULT_TARGETS_FILE, default_circle_dict  # noqa: F401
from .args import get_flight_easy_args, get_flight_args, make_env_args, apply_env_info  # noqa: F401
from .env import BatchedFlightEnv, FlightSearchEnvEasy, FlightSearchEnv  # noqa: F401
from . import _lib as lib  # noqa: F401
from . import dist  # noqa: F401
from .replay import DeviceReplayBuffer  # noqa: F401
from .agents import AgentRNN, BatchedAgents, FusedAgents, rnn_input_shape  # noqa: F401
from .collector import EpisodeCollector, EpsilonSchedule, evaluate, collect_experiment_data, random_policy  # noqa: F401

__all__ = ["BatchedFlightEnv", "FlightSearchEnvEasy", "FlightSearchEnv", "load_targets", "default_circle_dict",
           "get_flight_easy_args", "get_flight_args", "make_env_args", "lib"]
