"""Worker of tests/test_sweep_gloo.py: one rank of a world-size-N gloo group, oracle as the (stub) engine."""
import json
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_lib as O  # noqa: E402
from open_simulator_amd import sweep, synth  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    caps = json.loads(os.environ.get("SWEEP_CAPS", "[100, 100]"))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    prob, scen, orders = synth.config3(n_counts=12, n_orders=3, n_pods=400, n_het=14)
    mine = sweep.shard(scen, rank, world)
    res = O.run(prob, mine, orders, want_placement=False)            # stub engine: the CPU oracle
    plan = O.min_plan(prob, mine, res, *caps)
    rec = sweep.plan_record(bool(plan.found), plan.n_nodes, plan.scenario, plan.order_id, rank, world)
    g = sweep.all_gather_plan(rec)
    with open(os.environ["SWEEP_OUT"] + f".{rank}", "w") as f:
        json.dump({"plan": g.as_list(), "found": g.found, "local": g.local_scenario, "n_local": len(mine)}, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
