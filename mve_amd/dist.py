"""Multi-GPU plumbing for the embarrassingly parallel reference-view axis.

One process per GPU (torchrun / torch.distributed.run).  Reference views are independent
(apps/dmrecon/dmrecon.cc:285-318 runs them under `omp parallel for schedule(dynamic,1)`), so
the data path has NO collective: torch.distributed is used only for the start/stop barrier
and the max-over-ranks of the elapsed time.  Backend "nccl" (= RCCL) on GPUs, "gloo" in the
CPU tests.
"""
from __future__ import annotations

import os
from typing import List, Sequence


def rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def shard_views(view_ids: Sequence[int], rank: int, world: int) -> List[int]:
    """Strong-scaling partition of one scene's reference views: round-robin, so that ranks differ
    by at most one view (the static analogue of the reference's dynamic OpenMP schedule)."""
    return [v for i, v in enumerate(view_ids) if i % world == rank]


def rank_views(view_ids: Sequence[int], rank: int, world: int, scaling: str) -> List[int]:
    """The reference views rank `rank` reconstructs per step.
    weak: all of them, of its own scene replica (per-GPU work fixed as N grows);
    strong (BASELINE config 4): its share of ONE scene (total work fixed) -- apps/dmrecon/dmrecon.cc:285-318
    deals the views of one scene over its workers the same way."""
    if scaling == "weak" or world <= 1:
        return list(view_ids)
    if scaling != "strong":
        raise ValueError("scaling must be 'weak' or 'strong'")
    return shard_views(view_ids, rank, world)


class Collective:
    """Barrier + max-reduce; a no-op for a single process (no torch import at N = 1)."""

    def __init__(self, backend: str = "nccl", device_index: int = 0):
        self.rank, self.world, self.local_rank = rank_world()
        self.dist = None
        self.torch = None
        self.device = None
        if self.world > 1 or os.environ.get("MI_FORCE_DIST"):      # MI_FORCE_DIST: exercise the RCCL path with one rank
            import torch
            import torch.distributed as dist
            self.torch, self.dist = torch, dist
            if backend == "nccl":
                torch.cuda.set_device(device_index)
                self.device = torch.device("cuda", device_index)
            else:
                self.device = torch.device("cpu")
            if not dist.is_initialized():
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world)

    def barrier(self):
        if self.dist is not None:
            if self.device.type == "cuda":
                self.torch.cuda.synchronize()
            self.dist.barrier()
            if self.device.type == "cuda":
                self.torch.cuda.synchronize()

    def max(self, value: float) -> float:
        if self.dist is None:
            return float(value)
        t = self.torch.tensor([float(value)], dtype=self.torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum(self, value: float) -> float:
        if self.dist is None:
            return float(value)
        t = self.torch.tensor([float(value)], dtype=self.torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def close(self):
        if self.dist is not None and self.dist.is_initialized():
            self.dist.destroy_process_group()
