"""Multi-rank path on CPU: world size 2 (and 3) over gloo, sharded sweep + all-gather of the best plan must
equal the single-process answer over the unsharded scenario list."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import oracle_lib as O
from open_simulator_amd import sweep, synth

HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_world(world, caps, tmp_path):
    out = str(tmp_path / f"sweep_w{world}")
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   SWEEP_OUT=out, SWEEP_CAPS=json.dumps(caps))
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_sweep_worker.py")], env=env))
    for p in procs:
        assert p.wait(timeout=300) == 0
    return [json.load(open(f"{out}.{r}")) for r in range(world)]


@pytest.mark.parametrize("world,caps", [(2, [100, 100]), (3, [100, 100]), (2, [60, 100]), (2, [1, 1])])
def test_sharded_sweep_matches_single_process(world, caps, tmp_path):
    prob, scen, orders = synth.config3(n_counts=12, n_orders=3, n_pods=400, n_het=14)
    res = O.run(prob, scen, orders, want_placement=False)
    ref = O.min_plan(prob, scen, res, *caps)
    outs = run_world(world, caps, tmp_path)
    assert sum(o["n_local"] for o in outs) == len(scen)
    assert all(o["plan"] == outs[0]["plan"] for o in outs)            # every rank holds the same global plan
    if ref.found:
        n_nodes, rank, scenario, order_id = outs[0]["plan"]
        assert (n_nodes, scenario, order_id) == (ref.n_nodes, ref.scenario, ref.order_id)
        assert scenario % world == rank and sweep.global_index(outs[0]["local"], rank, world) == scenario
    else:
        assert not outs[0]["found"]


def test_shard_covers_every_node_count():
    _, scen, _ = synth.config3(n_counts=16, n_orders=4, n_pods=50, n_het=10)
    for world in (1, 2, 4, 8):
        counts = [set(sweep.shard(scen, r, world)[:, 0].tolist()) for r in range(world)]
        if world <= 4:                                                # 4 orders per count: up to 4 ranks see every count
            assert all(c == counts[0] for c in counts)
        assert sum(len(sweep.shard(scen, r, world)) for r in range(world)) == len(scen)


def test_reduce_records_rules():
    recs = [sweep.plan_record(True, 700, 5, 1, 0, 2), sweep.plan_record(True, 650, 9, 0, 1, 2)]
    g = sweep.reduce_records(recs)
    assert (g.found, g.n_nodes, g.rank, g.scenario) == (True, 650, 1, 19)
    recs = [sweep.plan_record(False, 0, -1, 0, 0, 2), sweep.plan_record(False, 0, -1, 0, 1, 2)]
    assert not sweep.reduce_records(recs).found
    recs = [sweep.plan_record(True, 650, 3, 0, 0, 2), sweep.plan_record(True, 650, 2, 0, 1, 2)]   # tie: lowest global index
    assert sweep.reduce_records(recs).scenario == 5
