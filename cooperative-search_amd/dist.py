"""Multi-GPU sharding and the path's only collective.

Environments are independent, so the batch shards by env with no data-path collective: rank r owns the global envs
[offset, offset + count); seeds derive from the GLOBAL env index, so results do not depend on the sharding.  The
single exchange step is the evaluation-metric reduction of the reference's `Runner.evaluate` (runner.py:86-96:
mean win_tag, episode_reward, targets_find over the evaluation episodes) and of the found-fraction curve of
`RolloutWorker.generate_replay` / `collect_experiment_data` (rollout.py:190-198, runner.py:163-171): every rank
contributes a small vector of partial sums, all-gathered (RCCL over xGMI when the backend is nccl; latency-bound:
<= 1.7 KB per rank) and summed locally.
"""
import numpy as np
import torch
import torch.distributed as dist

from .env import DEFAULT_BASE_SEED


def shard(global_batch, rank, world):
    """Contiguous shard of the global batch: (offset, count); the first `global_batch % world` ranks get one more."""
    base, extra = divmod(int(global_batch), int(world))
    count = base + (1 if rank < extra else 0)
    offset = rank * base + min(rank, extra)
    return offset, count


def seeds_for(offset, count, base_seed=DEFAULT_BASE_SEED):
    """np.random.seed values of the global envs [offset, offset + count)."""
    return ((int(base_seed) + int(offset) + np.arange(count, dtype=np.int64)) % (1 << 32)).astype(np.uint32)


def rank():
    return dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0


def all_gather_sum(partial):
    """Sum of `partial` (any-shape tensor of partial sums) over all ranks via ONE all-gather; identity when
    torch.distributed is not initialised.  Works on CPU tensors (gloo) and GPU tensors (nccl = RCCL).  A process group
    of ONE rank still runs the collective (RCCL at world_size 1 is the same code path as at 8: tests/test_gpu_rccl.py)."""
    if not (dist.is_available() and dist.is_initialized()):
        return partial.clone()
    parts = [torch.zeros_like(partial) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, partial.contiguous())
    return torch.stack(parts).sum(0)


def reduce_metrics(partials4):
    """[sum total_reward, sum win, sum target_find, count] (cs_metrics) -> the three numbers Runner.evaluate returns
    (win_rate, mean episode_reward, mean targets_find) over ALL ranks, plus the env count."""
    tot = all_gather_sum(partials4.to(torch.float64))
    n = float(tot[3].item())
    return {"win_rate": float(tot[1].item()) / n, "episode_reward": float(tot[0].item()) / n,
            "targets_find": float(tot[2].item()) / n, "episodes": int(n)}


class FoundCurve:
    """Per-rank accumulator of the found-fraction curve: res[t] = target_find / target_num after step t + 1,
    averaged over episodes (rollout.py:190-198; after an early termination the reference pads with 1.0, which is
    exactly what a frozen env's target_find / target_num keeps reporting, since early termination means all found)."""

    def __init__(self, episode_limit, target_num, device):
        self.sum = torch.zeros(episode_limit, dtype=torch.float64, device=device)
        self.count = torch.zeros(1, dtype=torch.float64, device=device)
        self.target_num = float(target_num)

    def add_step(self, t, target_find):
        self.sum[t] += target_find.to(torch.float64).sum() / self.target_num

    def end_episodes(self, n_envs):
        self.count += float(n_envs)

    def result(self):
        """average_res * 100 as runner.py:171 saves it (percent of targets found by step t), over all ranks."""
        tot = all_gather_sum(torch.cat([self.sum, self.count]))
        return (tot[:-1] / tot[-1] * 100.0).cpu().numpy()
