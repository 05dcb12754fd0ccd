#!/usr/bin/env python3
"""bench.py -- capacity-plan scenarios/sec on the BASELINE.json workload.

A "step" = one pass of the hot path over ONE scenario batch already resident in HBM:
config 3 of BASELINE.json (10k pods x 488..1511 nodes, 1024 node counts x 4 pod orders = 4096
scenarios) per GPU.  With N ranks (one process per GPU, launched by torch.distributed.run) every
rank owns 4096 scenarios of the 1024 x (4N) grid, interleaved so that each rank sees every node
count (N = 8 is config 4: 32k scenarios); scenarios are independent, so there is no data-path
collective -- only the per-rank best plan (48 B) is all-gathered over RCCL each step to pick the
global minimum-node plan.  value = scenarios all ranks completed / max-over-ranks time.

Prints ONE JSON line on rank 0.  The oracle (oracle/) is used only for the cpu_baseline leg.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s HBM3E spec peak
DTYPE = {1: "u32 (gcd-normalised int64 quantities) + f64 (BalancedAllocation)",
         2: "int64 quantities + f64 (BalancedAllocation / normalisations) + u8 (signature, node) score table",
         3: "f64-resident exact integers (gcd-normalised int64 quantities) + f64 (BalancedAllocation)",
         4: "u8 score table + u32 (gcd-normalised int64 quantities) + f64 (BalancedAllocation)"}
_N = ("achieved = algorithmic bytes per step (SURVEY 8d: sum over scenarios of P*(56*n+108), resp. P*((56+64+4)*n+4*G+108) "
      "with Open-Gpu-Share slots and anti-affinity domains) / HIP-event time of the scenario kernels of one step. ")
NOTE = {4: _N + "The (signature, node) score table replaces re-reading node state, so the ratio is not bounded by 1 (it is the "
                "speed-up over a state-streaming formulation); traffic = PMC-measured HBM-side bytes per step (profiles/); the "
                "binding limit is VALU issue + L2 latency of one wave per scenario, DESIGN.md section 5.3 / profiles/README.md",
        2: _N + "The all-feature kernel reads one table byte per node per cycle instead of the node's state and refreshes only "
                "the touched node's column, so the ratio can exceed a state-streaming formulation; binding limit: VALU issue of "
                "the per-node loop (2 waves per SIMD at 512 threads per scenario), DESIGN.md section 5.4",
        3: _N + "Node state is register-resident; binding limit: VALU issue, DESIGN.md section 5.1",
        1: _N + "Node state is register-resident; binding limit: VALU issue, DESIGN.md section 5.1"}


def algorithmic_bytes(scen, n_pods, workload="config3", n_groups=50):
    """per pod-placement step on an n-node scenario 56*n + 108 bytes (cpu+mem only), resp.
    (56+64+4)*n + 4*G + 108 with Open-Gpu-Share device slots and anti-affinity domains (config 5)."""
    n = scen[:, 0].astype(np.int64)
    per_node, extra = (124, 4 * n_groups + 108) if workload == "config5" else (56, 108)
    return int((n_pods * (per_node * n + extra)).sum())


VALU_ISSUE_PEAK = 256 * 4 * 2.4e9 / 4   # wave64 instructions / s: 256 CUs x 4 SIMDs, one wave64 VALU op per 4 cycles at 2.4 GHz


def measured_issue(kernel, scenarios, pods, kernel_ms):
    """The roofline that actually binds these integer / compare / fp64 kernels: VALU issue.  Instruction counts per step
    come from the committed PMC profile (profiles/traffic.json, SQ_INSTS_VALU), the time is measured live."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            for row in json.load(f)["entries"]:
                if row["kernel"] == kernel and row["scenarios_per_gpu"] == scenarios and row["pods"] == pods:
                    insts = row["insts_valu_per_dispatch"] * row["dispatches_per_step"]
                    ach = insts / (kernel_ms * 1e-3)
                    return {"bound": "valu_issue", "achieved": round(ach / 1e9, 1), "peak": round(VALU_ISSUE_PEAK / 1e9, 1),
                            "unit": "G wave64-instr/s", "frac": round(ach / VALU_ISSUE_PEAK, 4),
                            "wave_time_split": {"parked_on_waitcnt_or_barrier": round(row["wait_any_per_dispatch"] / row["wave_cycles_per_dispatch"], 3),
                                                "executing": round(row["active_inst_any_per_dispatch"] / row["wave_cycles_per_dispatch"], 3)},
                            "source": "SQ_INSTS_VALU / SQ_WAIT_ANY / SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES of profiles/r01e_*_summary.txt"}
    except (OSError, KeyError, ValueError):
        pass
    return None


def measured_traffic(kernel, scenarios, pods):
    """HBM-side bytes per step from the committed PMC profile of this workload (profiles/traffic.json: FETCH_SIZE x 2
    per MI355X_MICROARCH.md + WRITE_SIZE, summed over the kernel's launches of one step); None when no profile of
    this kernel / workload size is committed."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            for row in json.load(f)["entries"]:
                if row["kernel"] == kernel and row["scenarios_per_gpu"] == scenarios and row["pods"] == pods:
                    return row["hbm_bytes_per_step"]
    except (OSError, KeyError, ValueError):
        pass
    return None


def cpu_baseline(prob, scen, orders, budget_s=12.0):
    """Time the C oracle on a bounded sample of the same scenarios (rank 0 only): one scenario per task on every host
    core this process may use (the oracle call releases the GIL; a scenario is sequential, scenarios are independent --
    the same parallelism the GPU path uses).  The single-thread rate of the calibration run is reported alongside."""
    from concurrent.futures import ThreadPoolExecutor
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    oracle_lib.load()
    S = len(scen)
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for qf, pf in (("/sys/fs/cgroup/cpu.max", None), ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us")):
        try:                                               # a container's CPU quota (cgroup v2 / v1) caps the useful thread count
            if pf is None:
                quota, period = open(qf).read().split()
            else:
                quota, period = open(qf).read().strip(), open(pf).read().strip()
            if quota not in ("max", "-1"):
                cores = max(1, min(cores, -(-int(quota) // int(period))))
            break
        except (OSError, ValueError):
            continue
    cores = max(1, int(os.environ.get("SIMON_BENCH_CPU_THREADS", cores)))
    t0 = time.perf_counter()
    oracle_lib.run(prob, scen[[S // 2]], orders, want_placement=False)   # a mid-sized scenario: calibrate the sample size
    per = max(time.perf_counter() - t0, 1e-6)
    k = int(max(2, min(S, budget_s * cores / per)))
    pick = np.linspace(0, S - 1, k).astype(int)            # spread evenly over node counts and orders
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as pool:
        list(pool.map(lambda i: oracle_lib.run(prob, scen[[i]], orders, want_placement=False), pick.tolist()))
    dt = time.perf_counter() - t0
    return {"value": round(k / dt, 4), "unit": "scenarios/s", "cores": cores, "kind": "port",
            "sample": f"{k} of the {S} scenarios of this rank's batch (evenly spaced over node counts/orders), {dt:.1f} s on "
                      f"{cores} threads, one scenario per task (speed-up over one thread {k / dt * per:.1f}x; one thread alone: "
                      f"{1.0 / per:.2f} scenarios/s; C oracle: restated CPU baseline, not the Go reference binary)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--counts", type=int, default=1024, help="node counts in the sweep (1024 = BASELINE config 3/4)")
    ap.add_argument("--pods", type=int, default=10000)
    ap.add_argument("--orders-per-gpu", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", choices=["config3", "config5"], default="config3",
                    help="config3 = the BASELINE metric's workload (default); config5 = gpushare-style 50k pods x 5k nodes "
                         "(GPU share + anti-affinity + taints) on the all-feature kernel, 256 scenarios per GPU")
    ap.add_argument("--placement", type=int, default=1, help="store the [S][P] placement matrix in HBM (default on)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from open_simulator_amd import capi, sweep, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    # test hook (tests/test_gpu_parity.py::test_bench_two_ranks_share_one_gpu): several ranks on ONE device over gloo, to
    # exercise the world > 1 code path on a single-GPU box; the real launch is one rank per GPU over RCCL ("nccl")
    backend = os.environ.get("SIMON_BENCH_BACKEND", "nccl")
    if os.environ.get("SIMON_BENCH_SHARE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    n_orders = args.orders_per_gpu * world
    seed = synth.SEED + (3 if world == 1 else 4)
    if args.workload == "config5":
        prob, scen_all, orders = synth.config5(n_scen=int(os.environ.get("SIMON_BENCH_C5_SCEN", "256")) * world, n_orders=n_orders)
    else:
        prob, scen_all, orders = synth.config3(n_counts=args.counts, n_orders=n_orders, n_pods=args.pods, seed=seed)
    scen = sweep.shard(scen_all, rank, world)                 # every rank gets every node count
    S_local, S_total = len(scen), len(scen_all)

    ctx = capi.Context(local_rank)
    ctx.load_problem(prob)
    ctx.load_scenarios(scen, orders)                          # inputs resident in HBM before timing
    def step():
        ctx.run_loaded(want_placement=bool(args.placement))
        plan = ctx.min_plan()                                 # device-side reduction of this rank's batch
        rec = sweep.plan_record(bool(plan.found), plan.n_nodes, plan.scenario, plan.order_id, rank, world)
        return sweep.all_gather_plan(rec, device="cuda" if backend == "nccl" else "cpu").as_list()   # RCCL all-gather, 32 B per rank (no-op at N=1)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    kernel_ms = []
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        best = step()
        kernel_ms.append(ctx.stats().kernel_ms)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        st = ctx.stats()
        k_ms = float(np.mean(kernel_ms))
        alg = algorithmic_bytes(scen, prob.n_pods, args.workload)
        achieved = alg / (k_ms * 1e-3) / 1e9
        kname = {1: "simon::narrow_kernel", 2: "simon::wide_kernel", 3: "simon::fast_kernel", 4: "simon::cache_kernel"}.get(st.kernel_variant)
        traffic = measured_traffic(kname, S_local, prob.n_pods)
        value = S_total * args.steps / dt
        out = {
            "metric": "capacity-plan scenarios/sec (10k pods x ~1k nodes)", "value": round(value, 3),
            "unit": "scenarios/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": DTYPE.get(st.kernel_variant, "int64 + f64"), "data": "synthetic",
            "pods_placed_per_sec": round(value * prob.n_pods, 1),
            "config": {"workload": (f"BASELINE config 5-style (gpushare): {prob.n_pods} pods x " if args.workload == "config5" else
                                    f"BASELINE config {'3' if world == 1 else '4-style'}: {prob.n_pods} pods x ") + 
                                   f"{int(scen_all[:, 0].min())}..{int(scen_all[:, 0].max())} nodes, {len(set(scen_all[:, 0].tolist()))} node counts x "
                                   f"{n_orders} pod orders = {S_total} scenarios ({S_local} per GPU)",
                       "scenarios_per_gpu": S_local, "pods": prob.n_pods, "node_pool": prob.n_nodes,
                       "placement_matrix": bool(args.placement),
                       "kernel": {1: "narrow_v1", 2: "wide", 3: "narrow_fast", 4: "narrow_cache"}.get(st.kernel_variant, "?"), "workgroup": st.workgroup_size,
                       "slots_per_lane": st.slots_per_lane, "plan": best},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "kernel": kname, "kernel_ms": round(k_ms, 3), "launches_per_step": st.n_launches,
                         "algorithmic_bytes_per_launch": alg,
                         "binding": measured_issue(kname, S_local, prob.n_pods, k_ms),
                         "note": NOTE.get(st.kernel_variant, NOTE[4])},
        }
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(prob, scen, orders)
        print(json.dumps(out), flush=True)
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
