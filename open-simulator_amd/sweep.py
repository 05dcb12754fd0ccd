"""Multi-GPU node-count sweep: shard independent scenarios over ranks, one all-gather for the global plan.

The reference runs one Simulate per candidate size, serially (pkg/apply/apply.go:203-259).  Scenarios are
independent, so rank r of W evaluates scenarios r, r+W, r+2W, ... of the (node count x pod order) grid -- with
the grid ordered count-major, every rank sees every node count (balanced: cost grows with the count).  Inputs
are replicated, per-scenario outputs stay on the rank that produced them; the ONLY exchange is an all-gather of
one 32-byte record per rank (RCCL over xGMI when the backend is "nccl"; the payload is latency-bound), after
which every rank holds the same global minimum-node plan.

Backend-agnostic: `tests/test_sweep_gloo.py` drives it with gloo on CPU and a stub engine.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np

NO_PLAN = 1 << 40     # "no scenario of this rank schedules every pod"


def shard(scen: np.ndarray, rank: int, world: int) -> np.ndarray:
    """Scenarios of `rank`: every world-th one, so each rank gets every node count."""
    return np.ascontiguousarray(np.asarray(scen)[rank::world])


def global_index(local_index: int, rank: int, world: int) -> int:
    """Index into the unsharded scenario list of a rank-local scenario index."""
    return local_index * world + rank


@dataclass
class GlobalPlan:
    found: bool
    n_nodes: int
    rank: int            # rank that owns the winning scenario (its placement row lives there)
    scenario: int        # index into the UNSHARDED scenario list
    local_scenario: int
    order_id: int

    def as_list(self):
        return [self.n_nodes if self.found else -1, self.rank, self.scenario, self.order_id]


def plan_record(found: bool, n_nodes: int, local_scenario: int, order_id: int, rank: int, world: int):
    """The 4 x int64 record a rank contributes; lexicographic min over ranks = minimum node count, ties to the
    lowest global scenario index (the rule of simon_min_plan, extended across ranks)."""
    if not found:
        return [NO_PLAN, NO_PLAN, local_scenario, order_id]
    return [int(n_nodes), global_index(int(local_scenario), rank, world), int(local_scenario), int(order_id)]


def reduce_records(records) -> GlobalPlan:
    best_rank, best = min(enumerate(records), key=lambda t: (t[1][0], t[1][1]))
    if best[0] >= NO_PLAN:
        return GlobalPlan(False, -1, -1, -1, -1, -1)
    return GlobalPlan(True, int(best[0]), best_rank, int(best[1]), int(best[2]), int(best[3]))


def all_gather_plan(record, device: Optional[str] = None, group=None, force: bool = False) -> GlobalPlan:
    """All-gather the per-rank records and take the global minimum.  Without an initialised process group the
    call degenerates to the single-rank answer (`force`: run the collective even in a group of one rank -- the way a one-GPU box
    executes the RCCL calls of the multi-GPU launch, bench.py's SIMON_BENCH_FORCE_DIST)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force):
        return reduce_records([record])
    world = dist.get_world_size(group)
    mine = torch.tensor(record, dtype=torch.int64, device=device)
    out = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(out, mine, group=group)
    return reduce_records([t.tolist() for t in out])
