#!/usr/bin/env python3
"""What does the REST instantiation of the score-table kernel cost the pods that do not need it?  Config 3 (no GPU pod, no term)
run three ways on one box: one-level summary (generation 4), two-level (5, SIMON_TABLE_COARSE=1), and generation 6 forced by an
all-zero gpu_cnt array and one GPU pod (every other pod stays table-only).  python profiles/gpu_rest_overhead.py [n_counts] [n_orders]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from open_simulator_amd import capi, synth  # noqa: E402


def run(prob, scen, orders, env):
    for k, v in env.items():
        os.environ[k] = v
    try:
        with capi.Context(0) as ctx:
            ctx.load_problem(prob)
            ctx.load_scenarios(scen, orders)
            best = 1e9
            for _ in range(4):
                ctx.run_loaded(False)
                st = ctx.stats()
                best = min(best, st.kernel_ms)
            res = ctx.fetch(False)
            return best, st.kernel_generation, st.lds_bytes, res
    finally:
        for k in env:
            os.environ.pop(k, None)


def main():
    n_counts = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    n_orders = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    prob, scen, orders = synth.config3(n_counts=n_counts, n_orders=n_orders)
    a = run(prob, scen, orders, {})
    b = run(prob, scen, orders, {"SIMON_TABLE_COARSE": "1"})
    prob.gpu_cnt = np.zeros(len(prob.alloc_cpu), np.int32)
    prob.gpu_mem_total = np.zeros(len(prob.alloc_cpu), np.int64)
    prob.gpu_mem = np.zeros(len(prob.req_cpu), np.int64)
    prob.pod_gpu_cnt = np.zeros(len(prob.req_cpu), np.int32)
    prob.gpu_mem[-1] = 1 << 30                      # ONE pod asks for GPU memory (no node has any): the batch takes the REST instantiation
    prob.pod_gpu_cnt[-1] = 1
    c = run(prob, scen, orders, {})
    for name, r in (("one-level", a), ("two-level", b), ("REST instantiation", c)):
        print(f"{name:20s} generation {r[1]} lds {r[2]:6d} kernel_ms {r[0]:.3f} unscheduled sum {int(r[3].unscheduled.sum())}")


if __name__ == "__main__":
    main()
