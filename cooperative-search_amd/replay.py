"""HBM-resident episode replay buffer: the storage between collector and learner.

Counterpart of /root/reference/common/replay_buffer.py:5-101, which keeps `[size, episode_limit, ...]` float64 arrays
in host memory.  Here the ring lives on the device as float32 (a 3000-episode flight buffer with its 2504-wide obs is
~36 GB: sized for 288 GB of HBM, not for host RAM), whole batches of episodes are stored with one indexed copy per
key, and sampling never leaves the GPU.  The FIFO index rule (`_get_storage_idx`, :84-101), `can_sample`, uniform
sampling with replacement (`sample`, :63-68) and `sample_latest` (:70-82) follow the reference exactly; the index
arithmetic is plain host integers, as there.
"""
import numpy as np
import torch

KEYS = ("o", "u", "s", "r", "o_next", "s_next", "avail_u", "avail_u_next", "u_onehot", "padded", "terminated")


class DeviceReplayBuffer:
    def __init__(self, args, buffer_size, device="cuda", dtype=torch.float32):
        self.args = args
        self.n_actions, self.n_agents = args.n_actions, args.n_agents
        self.state_shape, self.obs_shape = args.state_shape, args.obs_shape
        self.size, self.episode_limit = int(buffer_size), args.episode_limit
        self.current_idx = 0
        self.current_size = 0
        obs = self.obs_shape + (args.map_size ** 2 if getattr(args, "conv", False) else 0)  # replay_buffer.py:18-21
        S, T, n, A = self.size, self.episode_limit, self.n_agents, self.n_actions
        shapes = {"o": (S, T, n, obs), "u": (S, T, n, 1), "s": (S, T, self.state_shape), "r": (S, T, 1),
                  "o_next": (S, T, n, obs), "s_next": (S, T, self.state_shape), "avail_u": (S, T, n, A),
                  "avail_u_next": (S, T, n, A), "u_onehot": (S, T, n, A), "padded": (S, T, 1), "terminated": (S, T, 1)}
        self.device = torch.device(device)
        self.buffers = {k: torch.empty(shapes[k], dtype=dtype, device=self.device) for k in KEYS}

    def _get_storage_idx(self, inc=None):
        """Ring positions for the next `inc` episodes.  Same sequence as the reference's three-case rule
        (replay_buffer.py:84-101), written as modular arithmetic: writing starts at the cursor (or at 0 when the cursor
        sits exactly at `size`, which the reference leaves un-wrapped after an exact fill), wraps modulo `size`, and the
        cursor ends one past the last written slot -- again left at `size` rather than 0 after an exact fill."""
        inc = inc or 1
        if inc > self.size:
            # the reference's index arithmetic breaks here too (it raises on the assignment); duplicated slots would
            # make several episodes race for one ring entry
            raise ValueError(f"cannot store {inc} episodes at once in a buffer of {self.size}")
        start = 0 if self.current_idx >= self.size else self.current_idx
        idx = (start + np.arange(inc)) % self.size
        end = start + inc
        self.current_idx = end if end <= self.size else end - self.size
        self.current_size = min(self.size, self.current_size + inc)
        return idx

    def store_episode(self, episode_batch):
        """episode_batch: dict of [k, T, ...] tensors (EpisodeCollector.generate_episodes) or ndarrays."""
        k = int(episode_batch["o"].shape[0])
        idx = torch.as_tensor(self._get_storage_idx(inc=k), device=self.device)
        for key in KEYS:
            src = torch.as_tensor(episode_batch[key], device=self.device).to(self.buffers[key].dtype)
            self.buffers[key].index_copy_(0, idx, src)

    def can_sample(self, batch_size):
        return self.current_size >= batch_size

    def sample(self, batch_size, generator=None):
        """Uniform with replacement over the filled part (replay_buffer.py:63-68), drawn on the device."""
        idx = torch.randint(0, self.current_size, (batch_size,), device=self.device, generator=generator)
        return {k: v.index_select(0, idx) for k, v in self.buffers.items()}

    def latest_indices(self, batch_size):
        """The `batch_size` most recently stored slots, oldest first (replay_buffer.py:70-79)."""
        assert self.can_sample(batch_size)
        return [(self.current_idx - batch_size + k) % self.current_size for k in range(batch_size)]

    def sample_latest(self, batch_size):
        idx = torch.as_tensor(self.latest_indices(batch_size), device=self.device)
        return {k: v.index_select(0, idx) for k, v in self.buffers.items()}
