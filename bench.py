#!/usr/bin/env python3
"""bench.py -- capacity-plan scenarios/sec on the BASELINE.json workload.

A "step" = one pass of the hot path over ONE scenario batch already resident in HBM:
config 3 of BASELINE.json (10k pods x 488..1511 nodes, 1024 node counts x 4 pod orders = 4096
scenarios) per GPU.  With N ranks (one process per GPU) every rank owns 4096 scenarios of the
1024 x (4N) grid, interleaved so that each rank sees every node count (N = 8 is config 4: 32k
scenarios); scenarios are independent, so there is no data-path collective -- only the per-rank
best plan (32 B) is all-gathered over RCCL each step to pick the global minimum-node plan.
value = scenarios all ranks completed / max-over-ranks time.

Launch: `python bench.py --gpus N` starts the N ranks itself (re-exec through torch.distributed.run,
127.0.0.1 rendezvous) when WORLD_SIZE is not set; under an external torch.distributed.run the ranks
read RANK / LOCAL_RANK / WORLD_SIZE from the environment as usual.

`python bench.py --group N` is the one-process variant: N devices behind simon_group_* (what a Go host does).

Output on rank 0: the LAST stdout line is ONE compact JSON record (<= 3 KB: the driver keeps the last ~8 KB of stdout and parses
the last line; round 3's 24.8 KB line lost its head) -- the headline record (config 3) with `roofline` + `cpu_baseline` +
`parity_sample` and a `digest` with one short row per other workload.  Everything else -- notes, peaks, instruction mixes, the
`end_to_end` leg from host buffers, `other_workloads` in full (config 2, 200 signatures, the Service workloads, config 5 at 256
scenarios and at the saturating batch size, the all-feature kernel's random mix; each with its OWN live-PMC roofline, CPU
baseline and parity sample) -- goes to the sidecar `bench_detail.json` (path in the line's `detail`; SIMON_BENCH_DETAIL overrides)
and to EARLIER stdout lines prefixed `#detail `.  The oracle (oracle/) is only the checker: `cpu_baseline` times it and
`parity_sample` compares what it computed with the GPU's results of the timed batch (exit code 3 on a mismatch).  Nothing under
oracle/ is on the timed path.
"""
import argparse
import glob
import json
import os
import shutil
import socket
import sqlite3
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# ---- chip constants, /opt/skills/guides/MI355X_MICROARCH.md ------------------------------------------------
HBM_PEAK_GBS = 8000.0            # 8 TB/s HBM3E spec peak
N_CU, N_SIMD, CLOCK_HZ = 256, 4, 2.4e9
VALU_ISSUE_PEAK = N_CU * N_SIMD * CLOCK_HZ / 2   # wave64 VALU instructions / s: SIMDs are 32 wide, a wave64 op issues in 2 cycles
LDS_BYTES_PER_CLK_CU = 128       # ds_read_b32 / b16 rate (the accesses of these kernels are <= 4 B per lane)

DTYPE = {1: "u32 (gcd-normalised int64 quantities) + f64 (BalancedAllocation)",
         2: "int64 quantities + f64 (BalancedAllocation / normalisations) + u8 (signature, node) score table",
         3: "f64-resident exact integers (gcd-normalised int64 quantities) + f64 (BalancedAllocation)",
         4: "u8 score table + u32 (gcd-normalised int64 quantities) + f64 (BalancedAllocation)"}
KERNEL_NAME = {1: "simon::narrow_kernel", 2: "simon::wide_kernel", 3: "simon::fast_kernel", 4: "simon::table_kernel"}
KERNEL_SHORT = {1: "narrow_v1", 2: "wide (all-feature kernel)", 3: "narrow_fast", 4: "score_table"}
SERVICE_PREF = 60                 # services of the `config3_service_pref` sub-record whose pods carry preferred self anti-affinity
SERVICE_ANTI = 20                 # services of the `config3_service_anti` sub-record that also require anti-affinity to their own pods (hostname key)
SERVICE_GPU = 20                  # services of the `config3_service_gpu20` sub-record whose pods ask for GPU memory (30 % of the nodes carry devices)
SERVICE_SHAPES = 30               # node shapes of the `config3_service_shapes30` sub-record: x 3 zones = 90 internal node classes (generation 7, two classes per lane)
SMALL_COUNTS = 16                 # node counts of the `service_small` sub-record (x 4 pod orders = 64 scenarios: what a sweep of candidate sizes looks like)
SIG_CLIFF = 300                   # ... and of the `config3_sigs300` row: three groups of 128 signatures per wave (the table's last regime before 384)
CLASS_RECORD = 80                 # distinct node shapes of the `config3_classes80` row: more than 64 internal node classes (two per lane on the score table)
CLASS_CLIFF = 160                 # ... and of `config3_classes160`: 129 .. 256 classes run generation 4's kernels of simon_table_cls4.hip since the end of round 6 (before: generation 1, the priced cliff)
SIG_RECORD = 200                  # request signatures of the `config3_sigs` sub-record (beyond the 128 two registers per lane hold; the table takes 384)
C5_SATURATING = 2048              # config-5 scenarios per GPU at which generation 6 saturates the chip (8 resident waves per CU; profiles/README.md)

PMC_GROUPS = [                    # one rocprofv3 pass each (FETCH_SIZE and WRITE_SIZE do not fit one pass; 8 SQ slots)
    ["FETCH_SIZE"],
    ["WRITE_SIZE"],
    ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY",
     "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY"],
    ["SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU",
     "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM"],
]


def algorithmic_bytes(scen, n_pods, workload="config3", n_groups=50):
    """per pod-placement step on an n-node scenario 56*n + 108 bytes (cpu+mem only), resp.
    (56+64+4)*n + 4*G + 108 with Open-Gpu-Share device slots and anti-affinity domains (config 5)."""
    n = scen[:, 0].astype(np.int64)
    per_node, extra = (124, 4 * n_groups + 108) if workload == "config5" else (56, 108)
    return int((n_pods * (per_node * n + extra)).sum())


# ---------------------------------------------------------------------------------------------------------------
# PMC counters of the dominant kernel: measured live (rocprofv3 passes over a child run of this script), else
# replayed from the committed profile -- the JSON says which.
# ---------------------------------------------------------------------------------------------------------------
def _pmc_pass(counters, child_args, timeout_s):
    """One `rocprofv3 --kernel-trace --pmc ...` pass over `bench.py --pmc-child`; returns {kernel: {counter: (dispatches,
    avg per dispatch)}} from the rocpd database, or raises."""
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        raise RuntimeError("rocprofv3 not found")
    tmp = tempfile.mkdtemp(prefix="simon_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [rocprof, "--kernel-trace", "--pmc", *counters, "-d", tmp, "-o", "pmc", "--",
           sys.executable, os.path.join(ROOT, "bench.py"), "--pmc-child", *child_args]
    try:
        subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s, check=True)
        out = {}
        for db in glob.glob(os.path.join(tmp, "**", "*.db"), recursive=True):
            con = sqlite3.connect(db)
            q = "select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name"
            for k, c, n, avg in con.execute(q):
                out.setdefault(k, {})[c] = (int(n), float(avg))
            con.close()
        if not out:
            raise RuntimeError("no counters in the rocprofv3 output")
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def pmc_live(kernel_name, child_args, budget_s):
    """All PMC groups for `kernel_name`; None when rocprofv3 is unavailable or a pass fails / runs out of budget."""
    merged, t0 = {}, time.perf_counter()
    short = kernel_name.split("::")[-1]
    for grp in PMC_GROUPS:
        left = budget_s - (time.perf_counter() - t0)
        if left < 20:
            break
        try:
            res = _pmc_pass(grp, child_args, timeout_s=min(left, 150))
        except Exception as e:                                        # noqa: BLE001 -- any failure means "not measured"
            print(f"[bench] PMC pass {grp[0]}.. failed: {e}", file=sys.stderr)
            if not merged:
                return None
            continue
        for k, cs in res.items():
            if short in k and "unpermute" not in k:
                merged.update(cs)
    return merged or None


def pmc_replayed(kernel, scenarios, pods):
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            for row in json.load(f)["entries"]:
                if row["kernel"] == kernel and row["scenarios_per_gpu"] == scenarios and row["pods"] == pods:
                    d = row["dispatches_per_step"]
                    return {"FETCH_SIZE": (d, row["fetch_kb_per_dispatch"]), "WRITE_SIZE": (d, row["write_kb_per_dispatch"]),
                            "SQ_INSTS_VALU": (d, row["insts_valu_per_dispatch"]), "SQ_INSTS_SALU": (d, row["insts_salu_per_dispatch"]),
                            "SQ_INSTS_LDS": (d, row["insts_lds_per_dispatch"]), "SQ_WAVE_CYCLES": (d, row["wave_cycles_per_dispatch"]),
                            "SQ_WAIT_ANY": (d, row["wait_any_per_dispatch"]), "SQ_ACTIVE_INST_ANY": (d, row["active_inst_any_per_dispatch"])}, row.get("source", "profiles/traffic.json")
    except (OSError, KeyError, ValueError):
        pass
    return None, None


def roofline_record(kname, variant, k_ms, launches, alg_bytes, pmc, source, workload, lds_bytes=None, scenarios=None):
    """The roofline of the dominant kernel.  These kernels are integer / compare / fp64 work on an L2-resident score
    table: the binding resource is VALU issue (plus the dependent latency of one wave per scenario), so `frac` is the VALU
    issue fraction against the guide's peak (one wave64 VALU op per 2 cycles per SIMD).  The SURVEY 8(d) figure
    (algorithmic bytes of a state-streaming formulation / time / 8 TB/s) is reported as `algorithmic_ratio` -- it is not
    bounded by 1 because the table replaces the state sweep it prices -- next to the measured HBM and LDS fractions."""
    sec = k_ms * 1e-3
    rec = {"bound": "valu_issue", "achieved": None, "peak": round(VALU_ISSUE_PEAK / 1e9, 1), "unit": "G wave64-instr/s", "frac": None,
           "kernel": kname, "kernel_ms": round(k_ms, 3), "launches_per_step": launches,
           "algorithmic_bytes_per_step": alg_bytes,
           "algorithmic_ratio": round(alg_bytes / sec / 1e9 / HBM_PEAK_GBS, 4),
           "algorithmic_gbs": round(alg_bytes / sec / 1e9, 1),
           "traffic": None, "measured_hbm_gbs": None, "measured_hbm_frac": None, "lds_frac": None,
           "peaks": {"hbm_gbs": HBM_PEAK_GBS, "valu_wave64_instr_per_s": VALU_ISSUE_PEAK,
                     "valu_rule": "256 CUs x 4 SIMDs x 2.4 GHz / 2 cycles per wave64 op (MI355X_MICROARCH.md)",
                     "lds_bytes_per_clk_per_cu": LDS_BYTES_PER_CLK_CU},
           "counters_source": source}
    if lds_bytes:
        # one wave (workgroup) per scenario: the LDS it asks for bounds the resident waves per CU (profiles/micro/occupancy_probe.hip)
        fit = min(32, int(160 * 1024 // max(int(lds_bytes), 1)))
        rec["lds_bytes_per_workgroup"] = int(lds_bytes)
        rec["workgroups_per_cu_that_fit"] = fit
        if scenarios:
            rec["workgroups_per_cu_offered"] = round(scenarios / N_CU, 2)
    if not pmc:
        rec["note"] = "no PMC counters available (rocprofv3 absent and no committed profile of this workload size): frac unmeasured"
        return rec

    def tot(name):                      # counter summed over the kernel's dispatches of ONE step
        if name not in pmc:
            return None
        n, avg = pmc[name]
        return avg * launches

    valu, salu, ldsi = tot("SQ_INSTS_VALU"), tot("SQ_INSTS_SALU"), tot("SQ_INSTS_LDS")
    if valu:
        rec["achieved"] = round(valu / sec / 1e9, 1)
        rec["frac"] = round(valu / sec / VALU_ISSUE_PEAK, 4)
    fetch, write = tot("FETCH_SIZE"), tot("WRITE_SIZE")
    if fetch is not None and write is not None:
        hbm = int(fetch * 1024 * 2 + write * 1024)   # KB per dispatch; FETCH_SIZE doubled per the guide's gfx950 note
        rec["traffic"] = hbm
        rec["measured_hbm_gbs"] = round(hbm / sec / 1e9, 1)
        rec["measured_hbm_frac"] = round(hbm / sec / 1e9 / HBM_PEAK_GBS, 4)
        rec["traffic_over_algorithmic"] = round(hbm / alg_bytes, 5)
    lds_act = tot("SQ_LDS_IDX_ACTIVE")
    if lds_act is not None:
        # SQ_LDS_IDX_ACTIVE = LDS-array cycles summed over the CUs; the LDS of one CU serves one access group per cycle
        rec["lds_frac"] = round(lds_act / (N_CU * sec * CLOCK_HZ), 4)
        rec["lds_bank_conflict_cycles_frac"] = round((tot("SQ_LDS_BANK_CONFLICT") or 0.0) / max(lds_act, 1.0), 4)
    va = tot("SQ_ACTIVE_INST_VALU")
    if va is not None:
        # SQ_ACTIVE_INST_VALU: quad-cycles in which a wave has a VALU instruction executing; one VALU per SIMD
        rec["valu_pipe_busy_frac"] = round(va * 4 / (N_CU * N_SIMD * sec * CLOCK_HZ), 4)
        if valu:
            rec["cycles_per_valu_instruction"] = round(va * 4 / valu, 2)
    wc = tot("SQ_WAVE_CYCLES")
    if wc:
        split = {}
        for key, name in (("parked_on_waitcnt_or_barrier", "SQ_WAIT_ANY"), ("issue_stalled", "SQ_WAIT_INST_ANY"), ("executing", "SQ_ACTIVE_INST_ANY")):
            v = tot(name)
            if v is not None:
                split[key] = round(v / wc, 3)
        rec["wave_time_split"] = split
    rec["instructions_per_step"] = {"valu": valu, "salu": salu, "lds": ldsi, "vmem_rd": tot("SQ_INSTS_VMEM_RD"), "vmem_wr": tot("SQ_INSTS_VMEM_WR")}
    rec["note"] = ("frac = SQ_INSTS_VALU / kernel time / VALU issue peak (the guide's 2 cycles per wave64 op).  One wave runs one scenario and its "
                   "pods are strictly sequential; with every scenario of the batch resident (16 waves per CU) the VALU pipe is the busiest "
                   "resource (valu_pipe_busy_frac: the instruction mix -- fp64, DPP, readlane, v_perm -- averages cycles_per_valu_instruction), "
                   "the rest is the dependent chain of a scheduling cycle (DESIGN.md 5.3); HBM / Infinity Cache carry the part of the "
                   "per-scenario byte tables that does not stay in L2 (measured_hbm_frac).")
    return rec


# ---------------------------------------------------------------------------------------------------------------
# CPU side: the oracle as baseline AND as checker of the timed batch
# ---------------------------------------------------------------------------------------------------------------
def host_cores():
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
    return max(1, int(os.environ.get("SIMON_BENCH_CPU_THREADS", cores)))


def oracle_sample(prob, scen, orders, budget_s, min_k=2, max_k=None):
    """Run the C oracle on a bounded, evenly spaced sample of this rank's scenarios, one scenario per task on every host
    core this process may use (the oracle call releases the GIL).  Returns (pick, results, timing dict)."""
    from concurrent.futures import ThreadPoolExecutor
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    oracle_lib.load()
    S = len(scen)
    cores = host_cores()
    t0 = time.perf_counter()
    oracle_lib.run(prob, scen[[S // 2]], orders, want_placement=False)   # a mid-sized scenario: calibrate the sample size
    per = max(time.perf_counter() - t0, 1e-6)
    k = int(max(min_k, min(S, budget_s * cores / per)))
    if max_k:
        k = min(k, max_k)
    pick = np.unique(np.linspace(0, S - 1, k).astype(int))            # spread evenly over node counts and orders
    k = len(pick)
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as pool:
        results = list(pool.map(lambda i: oracle_lib.run(prob, scen[[i]], orders, want_placement=True), pick.tolist()))
    dt = time.perf_counter() - t0
    return pick, results, {"k": k, "dt": dt, "cores": cores, "per": per}


def parity_check(ctx, pick, results, want_rows):
    """Compare the GPU results of the batch that was just timed with the oracle's, scenario by scenario: unscheduled count,
    used cpu / memory, and the full placement row (every pod's node) when the run stored placements."""
    gpu = ctx.fetch(want_placement=False)
    bad, rows = [], 0
    for i, ref in zip(pick.tolist(), results):
        ok = (int(gpu.unscheduled[i]) == int(ref.unscheduled[0]) and int(gpu.used_cpu[i]) == int(ref.used_cpu[0])
              and int(gpu.used_mem[i]) == int(ref.used_mem[0]))
        if ok and want_rows:
            row = ctx.fetch_placement(i)
            ok = bool((row == ref.placement[0]).all())
            rows += 1
        if not ok:
            bad.append(int(i))
    return {"scenarios": len(pick), "placement_rows": rows, "mismatches": len(bad), "first_mismatches": bad[:8],
            "compared": "unscheduled, used_cpu, used_mem" + (", placement[S][P] rows" if want_rows else ""),
            "checker": "oracle/simon_oracle.c (C restatement of the determinised reference) on the scenarios of the timed batch"}


# ---------------------------------------------------------------------------------------------------------------
def c5_scenarios(args):
    return int(args.c5_scenarios or os.environ.get("SIMON_BENCH_C5_SCEN", "256"))


def build_workload(args, synth, world):
    # weak scaling (default: BASELINE config 4 at N = 8 is 32 orders): every GPU gets --orders-per-gpu orders x every node count;
    # --strong: the batch stays config 3's 4 096 scenarios whatever N (the ranks split them, s mod N)
    n_orders = args.orders_per_gpu * (1 if getattr(args, "strong", False) else world)
    if args.workload == "config5":
        return synth.config5(n_scen=c5_scenarios(args) * world, n_orders=n_orders), n_orders
    if args.workload == "config2":
        return synth.config2(), 1
    if args.workload == "service":             # config 3's pool and sweep, every pod selected by a Service (system-default soft spread)
        return synth.config_service(n_counts=args.counts, n_orders=n_orders, n_pods=args.pods, n_anti=args.anti, n_pref=args.pref, n_hard=args.hard, n_gpu=args.gpu_services, n_shapes=args.shapes), n_orders
    if args.workload == "config5asdrawn":      # BASELINE config 5 AS DRAWN (per-pod requests and GPU requests) with its anti-affinity groups behind Services: the walks over the mask rows
        return synth.config5(n_scen=c5_scenarios(args), n_orders=4, services=True), 4
    if args.workload == "config5service":      # config 5's shape as Deployments behind Services: GPU share + required self anti-affinity + taints + soft spread
        return synth.config5_service(n_scen=c5_scenarios(args), n_orders=4), 4
    if args.workload == "typical":             # Kubernetes objects of a typical cluster, 64 candidate sizes (the `simon apply` shape; team mode of generation 7)
        return synth.typical_cluster_sweep(), 1
    if args.workload == "widemix":             # the adversarial random object mix: every plugin, 464 node classes -> the all-feature kernel
        return wide_mix_sweep(), 1
    if args.workload == "config3classes":      # config 3 with `--classes` distinct node shapes (80: two classes per lane; 160: off the score table)
        return synth.config3_classes(args.classes, n_counts=args.counts, n_orders=n_orders, n_pods=args.pods), n_orders
    if args.workload == "config3sig":          # config 3 with `--sigs` distinct request signatures (the > 64-signature regime)
        return synth.config3(n_counts=args.counts, n_orders=n_orders, n_pods=args.pods, seed=synth.SEED + 3, n_sigs=args.sigs), n_orders
    seed = synth.SEED + (3 if world == 1 or getattr(args, "strong", False) else 4)
    return synth.config3(n_counts=args.counts, n_orders=n_orders, n_pods=args.pods, seed=seed), n_orders


def wide_mix_sweep(n_nodes=2500, new_nodes=2500, n_workloads=400, max_replicas=250, n_counts=64, seed=1):
    """profiles/e2e_sweep.py's random mix as arrays: tests/randk8s.py objects (required / preferred (anti-)affinity on several keys, hard
    and soft spread constraints, taints, host ports, node selectors, random pod capacities: 464 node classes) -> workloads.expand ->
    flatten -> one scenario per candidate number of new nodes.  What lands on the all-feature kernel (simon::wide_kernel)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import randk8s
    from open_simulator_amd import k8s, simulate as sim
    nodes, workloads, services = randk8s.rand_cluster(seed, n_nodes=n_nodes, n_workloads=n_workloads, max_replicas=max_replicas)
    for j, n in enumerate(nodes):
        n["metadata"]["labels"][randk8s.ZONE] = f"z{j % 3}"       # zones round robin by index: every size is a prefix of the pool's nodeTree order
    ds = [{"apiVersion": "apps/v1", "kind": "DaemonSet", "metadata": {"name": "agent0", "namespace": "kube-system"},
           "spec": {"selector": {"matchLabels": {"app": "agent0"}},
                    "template": {"metadata": {"labels": {"app": "agent0"}},
                                 "spec": {"containers": [{"name": "a", "image": "x", "resources": {"requests": {"cpu": "100m", "memory": "128Mi"}}}],
                                          "tolerations": [{"operator": "Exists"}]}}}}]
    cluster = k8s.group_resources(nodes + services + ds)
    apps = [sim.AppResource("app", k8s.group_resources(workloads))]
    template = {"apiVersion": "v1", "kind": "Node", "metadata": {"name": "tmpl", "labels": {"disk": "ssd", randk8s.ZONE: "z0"}},
                "status": {"allocatable": {"cpu": "32", "memory": "64Gi", "pods": "40"}, "capacity": {"cpu": "32", "memory": "64Gi"}}}
    counts = np.unique(np.linspace(0, new_nodes, n_counts).astype(int)).tolist()
    batch = sim.sweep_batch(cluster, apps, template, counts)
    return batch.flat.problem, batch.scen, batch.orders


def workload_name(args, prob, scen_all, n_orders, S_local, world):
    head = {"config5": "BASELINE config 5-style (gpushare): ", "config2": "BASELINE config 2: ",
            "config5asdrawn": "BASELINE config 5 as drawn, its anti-affinity groups behind Services (walks over the position-mask rows): ",
            "config5service": "config 5's shape behind Services (GPU share + required self anti-affinity + taints + soft spread constraints): ",
            "service": "config 3 with every pod selected by a Service (system-default soft PodTopologySpread constraints): ",
            "typical": "typical cluster (Kubernetes objects: Deployments behind Services, preferred / required self anti-affinity, hard zone constraints): ",
            "widemix": "random Kubernetes-object mix with every plugin (all-feature kernel): ",
            "config3classes": f"config 3 variant with {args.classes} distinct node shapes (internal node classes): ",
            "config3sig": f"config 3 variant with {args.sigs} request signatures: "}.get(
                args.workload, f"BASELINE config {'3' if world == 1 or getattr(args, 'strong', False) else '4-style'}{' (fixed batch split over the ranks)' if getattr(args, 'strong', False) and world > 1 else ''}: ")
    return (head + f"{prob.n_pods} pods x {int(scen_all[:, 0].min())}..{int(scen_all[:, 0].max())} nodes, "
            f"{len(set(scen_all[:, 0].tolist()))} node counts x {n_orders} pod orders = {len(scen_all)} scenarios ({S_local} per GPU)")


def time_steps(ctx, steps, warmup, placement, fence, after_step=None):
    for _ in range(warmup):
        ctx.run_loaded(want_placement=placement)
        ctx.min_plan()
    k_ms = []
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.run_loaded(want_placement=placement)
        plan = ctx.min_plan()
        if after_step:
            after_step(plan)
        k_ms.append(ctx.stats().kernel_ms)
    fence()
    return time.perf_counter() - t0, float(np.mean(k_ms))


def kernel_of(st):
    return "simon::table_kernel" if getattr(st, "kernel_generation", 0) in (4, 5, 6) else KERNEL_NAME.get(st.kernel_variant)


def cpu_baseline_record(tm, S_local, what):
    return {"value": round(tm["k"] / tm["dt"], 4), "unit": "scenarios/s", "cores": tm["cores"], "kind": "port",
            "sample": f"{tm['k']} of the {S_local} scenarios of {what} (evenly spaced over node counts/orders), {tm['dt']:.1f} s on "
                      f"{tm['cores']} threads, one scenario per task (one thread alone: {1.0 / tm['per']:.2f} scenarios/s; C oracle = restated "
                      f"CPU baseline of the naive per-pod loop over all nodes, not the Go reference binary -- no Go toolchain on this box; the GPU/CPU "
                      f"ratio is mostly algorithmic: the kernel re-evaluates one table column per cycle, the oracle every node)"}


def end_to_end(capi, torch, prob, scen, orders, device):
    """What a host that starts from HOST buffers waits for (Applier.Run's loop as one batch): simon_load_* (H2D + staging), then
    simon_load_scenarios + simon_run_loaded + the per-scenario counts + simon_min_plan + simon_fetch_placement of the WINNING
    scenario's row -- never the [S][P] matrix.  A fresh context; wall clock around the calls."""
    torch.cuda.synchronize()
    with capi.Context(device) as ctx:
        t0 = time.perf_counter()
        ctx.load_problem(prob)
        t1 = time.perf_counter()
        ctx.load_scenarios(scen, orders)
        ctx.run_loaded(want_placement=True)
        res = ctx.fetch(want_placement=False)
        plan = ctx.min_plan()
        row = ctx.fetch_placement(plan.scenario if plan.found else 0)
        t2 = time.perf_counter()
        st = ctx.stats()
    return {"load_problem_ms": round((t1 - t0) * 1e3, 3), "batch_ms": round((t2 - t1) * 1e3, 3), "kernel_ms": round(st.kernel_ms, 3),
            "h2d_ms": round(st.h2d_ms, 3), "d2h_ms": round(st.d2h_ms, 3), "scenarios": len(scen),
            "scenarios_per_s": round(len(scen) / (t2 - t0), 1), "plan_found": bool(plan.found),
            "placed_in_winning_row": int((row >= 0).sum()), "unscheduled_min": int(res.unscheduled.min()),
            "calls": "simon_load_nodes/pods/class_tables | simon_load_scenarios, simon_run_loaded, simon_fetch_results (counts), "
                     "simon_min_plan, simon_fetch_placement(winning scenario)"}


def sub_record(name, capi, synth, torch, steps, warmup, oracle_k, pmc_mode, c5_scen=256, cpu_budget_s=6.0):
    """Driver-timed record of another BASELINE configuration on the same GPU (rank 0, N = 1), measured like the headline one: HIP-event
    kernel time, live PMC roofline of ITS kernel (same code path: child runs of this script), the oracle as CPU baseline on a sample
    of ITS scenarios, and the parity of exactly that sample (every placement row)."""
    if name == "config2":
        prob, scen, orders = synth.config2()
        child = ["--workload", "config2"]
        wl, label = "config3", "BASELINE config 2"
    elif name == "service":                         # config 3 with every pod behind a Service: soft spread constraints (generation 7)
        prob, scen, orders = synth.config_service()
        child = ["--workload", "service"]
        wl, label = "config3", "config 3 with Service-selected pods"
    elif name == "service_anti":                    # ... 20 of the 60 services with required anti-affinity to their own pods on the hostname key
        prob, scen, orders = synth.config_service(n_anti=SERVICE_ANTI)
        child = ["--workload", "service", "--anti", str(SERVICE_ANTI)]
        wl, label = "config3", f"config 3 with Service-selected pods, {SERVICE_ANTI} services with required self anti-affinity (hostname)"
    elif name == "service_pref":                    # ... every service also PREFERS not to sit next to its own pods (hostname 100, zone 50)
        prob, scen, orders = synth.config_service(n_pref=SERVICE_PREF)
        child = ["--workload", "service", "--pref", str(SERVICE_PREF)]
        wl, label = "config3", f"config 3 with Service-selected pods that prefer not to sit next to their own kind ({SERVICE_PREF} services, hostname 100 + zone 50)"
    elif name == "service_shapes":                  # ... the existing nodes in 30 shapes: x 3 zones = 90 internal node classes (round 6: two classes per lane in the walks; was: the all-feature kernel)
        prob, scen, orders = synth.config_service(n_counts=SMALL_COUNTS * 4, n_shapes=SERVICE_SHAPES)
        child = ["--workload", "service", "--counts", str(SMALL_COUNTS * 4), "--shapes", str(SERVICE_SHAPES)]
        wl, label = "config3", f"config 3 with Service-selected pods, nodes in {SERVICE_SHAPES} shapes x 3 zones, {16 * SMALL_COUNTS} scenarios"
    elif name == "service_gpu":                     # a gpushare cluster behind Services: GPU share folded into the table, on generation 7 (was: the all-feature kernel)
        prob, scen, orders = synth.config_service(n_counts=SMALL_COUNTS * 4, n_gpu=SERVICE_GPU)
        child = ["--workload", "service", "--counts", str(SMALL_COUNTS * 4), "--gpu-services", str(SERVICE_GPU)]
        wl, label = "config3", f"config 3 with Service-selected pods, {SERVICE_GPU} of the 60 services asking for GPU share, {16 * SMALL_COUNTS} scenarios"
    elif name == "config5_asdrawn":                 # BASELINE config 5 as drawn, its anti-affinity groups behind Services: REST && SPREAD in one instantiation
        prob, scen, orders = synth.config5(n_scen=c5_scen, n_orders=4, services=True)
        child = ["--workload", "config5asdrawn", "--c5-scenarios", str(c5_scen)]
        wl, label = "config5", f"BASELINE config 5 as drawn, its anti-affinity groups behind Services, at {c5_scen} scenarios"
    elif name == "config5_service":                 # config 5's shape as Deployments behind Services: GPU fold + anti-affinity fold + generation 7's walk
        prob, scen, orders = synth.config5_service(n_scen=c5_scen, n_orders=4)
        child = ["--workload", "config5service", "--c5-scenarios", str(c5_scen)]
        wl, label = "config5", f"config 5's shape behind Services at {c5_scen} scenarios"
    elif name == "service_small":                   # the `simon apply` shape: 64 candidate scenarios -- generation 7 in team mode (four waves per scenario)
        prob, scen, orders = synth.config_service(n_counts=SMALL_COUNTS)
        child = ["--workload", "service", "--counts", str(SMALL_COUNTS)]
        wl, label = "config3", f"config 3 with Service-selected pods, {4 * SMALL_COUNTS} scenarios"
    elif name == "config3_small":                   # the `simon apply` shape without Services: 64 candidate scenarios -- generation 4, one wave per scenario, the workspace in LDS (5.3k)
        prob, scen, orders = synth.config3(n_counts=SMALL_COUNTS)
        child = ["--workload", "config3", "--counts", str(SMALL_COUNTS)]
        wl, label = "config3", f"BASELINE config 3's pool, {4 * SMALL_COUNTS} scenarios"
    elif name == "typical":                         # Kubernetes objects through the host mirror: 50 707 pods x 2 500..5 000 nodes, 64 candidate sizes
        prob, scen, orders = synth.typical_cluster_sweep()
        child = ["--workload", "typical"]
        wl, label = "config3", "typical cluster, 64 candidate sizes"
    elif name == "widemix":                         # what the all-feature kernel gets: the adversarial random mix, 64 candidate sizes
        prob, scen, orders = wide_mix_sweep()
        child = ["--workload", "widemix"]
        wl, label = "config5", "random object mix, 64 candidate sizes"
    elif name == "config3sig_cliff":                # the price of the cliffs as numbers: 300 signatures ...
        prob, scen, orders = synth.config3(n_counts=1024, n_orders=4, n_pods=10000, seed=synth.SEED + 3, n_sigs=SIG_CLIFF)
        child = ["--workload", "config3sig", "--sigs", str(SIG_CLIFF)]
        wl, label = "config3", f"config 3 with {SIG_CLIFF} request signatures"
    elif name in ("config3_classes", "config3_classes_cliff"):   # ... 80 node shapes (two classes per lane) and 160 (four, simon_table_cls4.hip; beyond 256 internal node classes the problem leaves the score table)
        ncl = CLASS_RECORD if name == "config3_classes" else CLASS_CLIFF
        prob, scen, orders = synth.config3_classes(ncl)
        child = ["--workload", "config3classes", "--classes", str(ncl)]
        wl, label = "config3", f"config 3 with {ncl} distinct node shapes"
    elif name == "config3sig":                      # config 3 with SIG_RECORD request signatures: the > 128-signature regime as a number
        prob, scen, orders = synth.config3(n_counts=1024, n_orders=4, n_pods=10000, seed=synth.SEED + 3, n_sigs=SIG_RECORD)
        child = ["--workload", "config3sig", "--sigs", str(SIG_RECORD)]
        wl, label = "config3", f"config 3 with {SIG_RECORD} request signatures"
    else:
        prob, scen, orders = synth.config5(n_scen=c5_scen, n_orders=4)
        child = ["--workload", "config5", "--c5-scenarios", str(c5_scen)]
        wl, label = "config5", f"BASELINE config 5 at {c5_scen} scenarios"
    device = torch.cuda.current_device()
    rec = {"workload": {"config2": "config2", "service": "config3_service", "service_small": f"config3_service_S{4 * SMALL_COUNTS}", "config3_small": f"config3_S{4 * SMALL_COUNTS}", "service_gpu": f"config3_service_gpu{SERVICE_GPU}_S{16 * SMALL_COUNTS}", "service_shapes": f"config3_service_shapes{SERVICE_SHAPES}_S{16 * SMALL_COUNTS}", "config5_service": f"config5_service_S{c5_scen}", "config5_asdrawn": f"config5_asdrawn_service_S{c5_scen}", "typical": "typical_cluster_x64", "widemix": "wide_mix_x64", "service_anti": f"config3_service_anti{SERVICE_ANTI}", "service_pref": f"config3_service_pref{SERVICE_PREF}",
                        "config3sig": f"config3_sigs{SIG_RECORD}", "config3sig_cliff": f"config3_sigs{SIG_CLIFF}",
                        "config3_classes": f"config3_classes{CLASS_RECORD}", "config3_classes_cliff": f"config3_classes{CLASS_CLIFF}"}.get(name, f"config5_S{c5_scen}")}
    with capi.Context(device) as ctx:
        ctx.load_problem(prob)
        ctx.load_scenarios(scen, orders)
        dt, k_ms = time_steps(ctx, steps, warmup, True, torch.cuda.synchronize)
        st = ctx.stats()
        rec.update({"scenarios": len(scen), "pods": prob.n_pods, "nodes": f"{int(scen[:, 0].min())}..{int(scen[:, 0].max())}",
                    "value": round(len(scen) * steps / dt, 3), "unit": "scenarios/s",
                    "pods_placed_per_sec": round(len(scen) * steps / dt * prob.n_pods, 1),
                    "ms_per_step": round(dt / steps * 1e3, 3), "steps": steps, "warmup": warmup,
                    "kernel": KERNEL_SHORT.get(st.kernel_variant, "?"), "kernel_generation": st.kernel_generation,
                    "kernel_ms": round(k_ms, 3), "workgroup": st.workgroup_size})
        if st.kernel_generation == 7 and st.workgroup_size > 64:   # several waves per scenario (team mode of generation 7): the same batch with ONE wave per scenario, same process
            os.environ["SIMON_TEAM"] = "0"
            try:
                with capi.Context(device) as ctx1:
                    ctx1.load_problem(prob)
                    ctx1.load_scenarios(scen, orders)
                    _, k1 = time_steps(ctx1, max(2, steps // 2), 1, True, torch.cuda.synchronize)
            finally:
                os.environ.pop("SIMON_TEAM", None)
            rec["team"] = {"waves_per_scenario": st.workgroup_size // 64, "one_wave_kernel_ms": round(k1, 3), "speedup": round(k1 / k_ms, 3)}
        if name in ("config2", "config3_small") and st.kernel_generation == 4:   # small batch of a small problem: the scenario's workspace lives in LDS (round 5) -- the same batch with the workspace in HBM, same process
            os.environ["SIMON_LDS_WS"] = "0"
            try:
                with capi.Context(device) as ctx1:
                    ctx1.load_problem(prob)
                    ctx1.load_scenarios(scen, orders)
                    _, k1 = time_steps(ctx1, max(2, steps // 2), 1, True, torch.cuda.synchronize)
                    lds0 = ctx1.stats().lds_bytes
            finally:
                os.environ.pop("SIMON_LDS_WS", None)
            rec["lds_workspace"] = {"lds_bytes": st.lds_bytes, "in_lds": st.lds_bytes > lds0, "workspace_in_hbm_kernel_ms": round(k1, 3), "speedup": round(k1 / k_ms, 3)}
        if oracle_k > 0:
            pick, refs, tm = oracle_sample(prob, scen, orders, budget_s=cpu_budget_s, min_k=min(oracle_k, len(scen)), max_k=max(oracle_k, 1))
            rec["cpu_baseline"] = cpu_baseline_record(tm, len(scen), label)
            rec["parity_sample"] = parity_check(ctx, pick, refs, True)
    alg = algorithmic_bytes(scen, prob.n_pods, wl)
    pmc, source = None, None
    if pmc_mode in ("auto", "live"):
        t0 = time.perf_counter()
        pmc = pmc_live(kernel_of(st), child + ["--steps", "1", "--warmup", "0"], budget_s=float(os.environ.get("SIMON_BENCH_PMC_BUDGET_S", "240")) / 2)
        if pmc:
            source = (f"measured live in this run: rocprofv3 --kernel-trace --pmc, {len(PMC_GROUPS)} separate passes over a 1-step child run of "
                      f"this workload ({time.perf_counter() - t0:.0f} s)")
    rec["roofline"] = roofline_record(kernel_of(st), st.kernel_variant, k_ms, st.n_launches, alg, pmc, source, wl,
                                      lds_bytes=st.lds_bytes if st.kernel_variant == 4 else None, scenarios=len(scen))
    if name == "config5" and c5_scen <= 256:          # what a host that starts from HOST buffers waits for at this size (staging included)
        rec["end_to_end"] = end_to_end(capi, torch, prob, scen, orders, device)
    return rec


# ---------------------------------------------------------------------------------------------------------------
# Output: full record -> sidecar + `#detail` lines; compact record (<= LINE_BUDGET bytes) -> the LAST stdout line
# ---------------------------------------------------------------------------------------------------------------
LINE_BUDGET = 3072                # bytes of the final JSON line (the driver keeps ~8 KB of the stdout tail)


def _short_roofline(r):
    if not r:
        return None
    src = r.get("counters_source") or ""
    keep = ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms", "launches_per_step", "measured_hbm_gbs",
            "measured_hbm_frac", "algorithmic_ratio", "lds_frac", "valu_pipe_busy_frac")
    out = {k: r.get(k) for k in keep if k in r}
    out["counters"] = "live" if src.startswith("measured live") else ("replayed" if src.startswith("replayed") else None)
    return out


DIGEST_COLS = ["name", "scenarios_per_s", "kernel_ms", "generation", "roofline_frac", "hbm_frac", "cpu_scenarios_per_s", "team_speedup", "checked", "mismatches"]


def _digest_row(w):
    """One row per other workload, in DIGEST_COLS order (a list, not a dict: the line has a byte budget)."""
    r = w.get("roofline") or {}
    if "error" in w:
        return [w.get("workload"), None, None, None, None, None, None, None, None, "error: " + str(w["error"])[:60]]
    return [w.get("workload"), w.get("value"), w.get("kernel_ms"), w.get("kernel_generation"), r.get("frac"), r.get("measured_hbm_frac"),
            (w.get("cpu_baseline") or {}).get("value"), (w.get("team") or w.get("lds_workspace") or {}).get("speedup"), (w.get("parity_sample") or {}).get("scenarios"),
            (w.get("parity_sample") or {}).get("mismatches")]


def compact_line(out, detail_name):
    """The record the driver parses: the contract's keys, the headline `roofline` / `cpu_baseline` / `parity_sample` without their
    prose, one digest row per other workload.  Trimmed further (never the contract's keys) if it would exceed LINE_BUDGET."""
    cfg = out.get("config", {})
    line = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                    "vs_baseline", "dtype", "data", "pods_placed_per_sec")}
    line["config"] = {k: cfg.get(k) for k in ("workload", "scenarios_per_gpu", "pods", "node_pool", "kernel", "kernel_generation", "plan") if k in cfg}
    if out.get("roofline"):
        line["roofline"] = _short_roofline(out["roofline"])
    cb = out.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {"value": cb["value"], "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"], "sample": cb["sample"][:120]}
    ps = out.get("parity_sample")
    if ps:
        line["parity_sample"] = {k: ps.get(k) for k in ("scenarios", "placement_rows", "mismatches")}
    rk = out.get("ranks") or {}
    if rk.get("world_size", 1) > 1 or "devices" in rk or rk.get("forced_one_rank_group"):
        sc = rk.get("selfcheck") or {}
        line["ranks"] = {k: v for k, v in {"world_size": rk.get("world_size"), "backend": rk.get("backend"), "mode": (rk.get("mode") or "")[:40] or None,
                                           "all_gather_ms_per_step": rk.get("all_gather_ms_per_step"), "collective": rk.get("plan_collective"),
                                           "distinct_devices": sc.get("distinct_devices_over_ranks"), "forced_one_rank_group": rk.get("forced_one_rank_group"),
                                           "rccl_version": sc.get("rccl_version"),
                                           "rccl_ranks": (rk.get("world_size") if rk.get("backend") == "nccl" else 0) if rk.get("world_size") else None,
                                           "devices": sc.get("device_list"),
                                           "one_device_test_hook": sc.get("one_device_test_hook", rk.get("one_device_test_hook"))}.items() if v is not None}
    e2e = out.get("end_to_end")
    if e2e:
        line["end_to_end"] = {k: e2e.get(k) for k in ("load_problem_ms", "batch_ms", "kernel_ms", "scenarios_per_s")}
    if out.get("other_workloads"):
        line["digest"] = {"cols": DIGEST_COLS, "rows": [_digest_row(w) for w in out["other_workloads"]]}
    line["detail"] = detail_name
    def drop_col(name):
        def f():
            dg = line.get("digest")
            if dg and name in dg["cols"]:
                i = dg["cols"].index(name)
                dg["cols"] = dg["cols"][:i] + dg["cols"][i + 1:]
                dg["rows"] = [r[:i] + r[i + 1:] for r in dg["rows"]]
        return f

    def sig4():                                                      # digest numbers to 4 significant digits
        dg = line.get("digest")
        if dg:
            dg["rows"] = [[float(f"{v:.4g}") if isinstance(v, float) else v for v in r] for r in dg["rows"]]

    trims = [lambda: line.pop("end_to_end", None), sig4, lambda: line["roofline"].pop("lds_frac", None) if line.get("roofline") else None,
             lambda: line["roofline"].pop("valu_pipe_busy_frac", None) if line.get("roofline") else None,
             drop_col("cpu_scenarios_per_s"), drop_col("checked"), drop_col("generation"), drop_col("team_speedup"), drop_col("hbm_frac"),
             lambda: line["cpu_baseline"].update(sample=line["cpu_baseline"]["sample"][:40]) if line.get("cpu_baseline") else None,
             drop_col("roofline_frac"), lambda: line.pop("digest", None)]
    for trim in trims:                                               # never the contract's keys, roofline, cpu_baseline, parity_sample
        if len(json.dumps(line)) <= LINE_BUDGET:
            break
        trim()
    return line


_RECORD_OUT = None


def claim_stdout():
    """The record must be the LAST line of stdout, and libraries print there too: RCCL writes its version banner ("RCCL version : ...",
    five lines) through C stdio when the first communicator comes up, which a pipe delivers at exit -- after the record (seen on the
    GPU box, r06b).  So the process keeps the real stdout for its own lines and points fd 1 at stderr for everything else."""
    global _RECORD_OUT
    if _RECORD_OUT is None:
        try:
            sys.stdout.flush()
            fd = os.dup(1)
            os.dup2(2, 1)
            _RECORD_OUT = os.fdopen(fd, "w", buffering=1)
        except OSError as e:                                   # a closed stdout / stderr: print as before
            print(f"[bench] could not re-point stdout ({e}); library output may follow the record", file=sys.stderr)


def emit(out):
    """Sidecar + `#detail` lines first, the compact record LAST (flushed): whatever tail of stdout survives holds it whole."""
    path = os.environ.get("SIMON_BENCH_DETAIL") or os.path.join(ROOT, "bench_detail.json")
    name = os.path.relpath(path, ROOT) if path.startswith(ROOT + os.sep) else path
    try:
        with open(path, "w") as f:
            json.dump(out, f, indent=1)
    except OSError as e:
        print(f"[bench] could not write {path}: {e}", file=sys.stderr)
        name = None
    rec_out = _RECORD_OUT or sys.stdout
    head = {k: v for k, v in out.items() if k != "other_workloads"}
    print("#detail " + json.dumps(head), file=rec_out, flush=True)
    for w in out.get("other_workloads", []):
        print("#detail " + json.dumps(w), file=rec_out, flush=True)
    line = compact_line(out, name)
    text = json.dumps(line)
    assert len(text) <= 4096, f"final bench line is {len(text)} bytes"
    print(text, file=rec_out, flush=True)


def multi_rank_selfcheck(torch, dist, world, rank, local_rank, backend):
    """What must hold when N ranks claim N GPUs (the nccl branch has never met hardware: the line states what it saw).  Unless the
    one-device test hook is set: enough visible devices, and every rank on its own device (all-gather of the device identity)."""
    shared = os.environ.get("SIMON_BENCH_SHARE_DEVICE") == "1"
    ndev = torch.cuda.device_count()
    props = torch.cuda.get_device_properties(local_rank)
    ident = str(getattr(props, "uuid", "")) or f"{getattr(props, 'pci_bus_id', -1)}:{local_rank}"
    idents = [None] * world
    dist.all_gather_object(idents, (rank, local_rank, ident))
    distinct = len({i[2] for i in idents})
    if not shared:
        if ndev < world:
            raise SystemExit(f"bench.py: {world} ranks but only {ndev} visible device(s)")
        if distinct != world:
            raise SystemExit(f"bench.py: {world} ranks share {distinct} device(s): {idents}")
    info = {"visible_devices": ndev, "distinct_devices_over_ranks": distinct, "one_device_test_hook": shared,
            "devices": [f"rank {r}: local_rank {lr}" for r, lr, _ in idents],
            # per rank, in rank order: local device index and the device's identity (uuid / bus id) -- what the driver's SCALE record
            # is judged by the day a multi-GPU node runs this
            "device_list": [f"{lr}:{str(idt)[-12:]}" for _, lr, idt in sorted(idents)]}
    if backend == "nccl":
        try:
            info["rccl_version"] = ".".join(str(x) for x in torch.cuda.nccl.version())
        except Exception as e:                                     # noqa: BLE001
            info["rccl_version"] = f"unavailable ({e!r})"
    return info


def run_group(args, capi, synth, torch):
    """`--group N`: ONE process, N devices behind simon_group_* -- what a Go host is (integration/go/hipengine).  The library deals
    scenario s to member s % N, replicates the inputs, runs the members concurrently and reduces their plans on the host; the same
    BASELINE config 3 / 4 grid as the multi-rank launch."""
    n = args.group
    ndev = capi.load_library().simon_hip_device_count()
    shared = os.environ.get("SIMON_BENCH_SHARE_DEVICE") == "1"
    if ndev < n and not shared:
        raise SystemExit(f"bench.py --group {n}: only {ndev} GPU(s) visible (SIMON_BENCH_SHARE_DEVICE=1 puts the members on device 0: test hook)")
    devices = [i % max(ndev, 1) for i in range(n)] if ndev < n else list(range(n))
    (prob, scen, orders), n_orders = build_workload(args, synth, n)
    with capi.Group(devices) as g:
        g.load_problem(prob)
        g.load_scenarios(scen, orders)
        for _ in range(args.warmup):
            g.run_loaded(True)
            g.min_plan()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            g.run_loaded(True)
            plan, _ = g.min_plan()
        collective = g.collective()
        for d in sorted(set(devices)):
            torch.cuda.synchronize(d)
        dt = time.perf_counter() - t0
        sts = [g.member_stats(i) for i in range(n)]
        par = None
        if not args.no_cpu_baseline:
            pick, refs, _ = oracle_sample(prob, scen, orders, budget_s=4.0, max_k=48)
            res = g.fetch(False)
            bad = [int(i) for i, ref in zip(pick.tolist(), refs)
                   if not (int(res.unscheduled[i]) == int(ref.unscheduled[0]) and int(res.used_cpu[i]) == int(ref.used_cpu[0])
                           and bool((g.fetch_placement(int(i)) == ref.placement[0]).all()))]
            par = {"scenarios": len(pick), "placement_rows": len(pick), "mismatches": len(bad), "first_mismatches": bad[:8],
                   "checker": "oracle/simon_oracle.c on scenarios of the timed batch (global indices through the group)"}
    value = len(scen) * args.steps / dt
    st0 = sts[0]
    out = {"metric": "capacity-plan scenarios/sec (10k pods x ~1k nodes)", "value": round(value, 3), "unit": "scenarios/s", "n_gpus": n,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong" if args.strong else "weak",
           "vs_baseline": None, "dtype": DTYPE.get(st0.kernel_variant, "int64 + f64"), "data": "synthetic",
           "pods_placed_per_sec": round(value * prob.n_pods, 1),
           "config": {"workload": workload_name(args, prob, scen, n_orders, len(scen) // n, n), "scenarios_per_gpu": len(scen) // n,
                      "pods": prob.n_pods, "node_pool": prob.n_nodes, "kernel": KERNEL_SHORT.get(st0.kernel_variant, "?"),
                      "kernel_generation": st0.kernel_generation, "plan": plan.as_dict()},
           "ranks": {"mode": "one process, simon_group over the device list", "plan_collective": collective,
                     "plan_collective_note": "rccl_all_gather = ONE ncclAllGather of the 8-byte plan keys over the members' own devices (distinct devices only); "
                                             "host = the members' plans reduced on the host (one member, shared devices under the test hook, no librccl)",
                     "devices": devices, "one_device_test_hook": shared and ndev < n,
                     "member_kernel_ms": [round(s.kernel_ms, 3) for s in sts]}}
    if par is not None:
        out["parity_sample"] = par
    emit(out)
    if par and par["mismatches"]:
        raise SystemExit(3)


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks through torch.distributed.run on 127.0.0.1."""
    from open_simulator_amd import capi
    ndev = capi.load_library().simon_hip_device_count()
    env = dict(os.environ)
    if ndev < n:
        if env.get("SIMON_BENCH_SHARE_DEVICE") != "1":
            raise SystemExit(f"bench.py --gpus {n}: only {ndev} GPU(s) visible (SIMON_BENCH_SHARE_DEVICE=1 runs the ranks on one device over gloo: test hook)")
        env.setdefault("SIMON_BENCH_BACKEND", "gloo")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["SIMON_BENCH_SELF_SPAWNED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *argv]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--counts", type=int, default=1024, help="node counts in the sweep (1024 = BASELINE config 3/4)")
    ap.add_argument("--pods", type=int, default=10000)
    ap.add_argument("--orders-per-gpu", type=int, default=4)
    ap.add_argument("--sigs", type=int, default=100, help="request signatures of --workload config3sig")
    ap.add_argument("--classes", type=int, default=CLASS_RECORD, help="distinct node shapes of --workload config3classes")
    ap.add_argument("--c5-scenarios", type=int, default=0, help="scenarios per GPU of --workload config5 (default 256; env SIMON_BENCH_C5_SCEN)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the oracle legs (cpu_baseline and parity_sample)")
    ap.add_argument("--no-sub", action="store_true", help="skip the config-2 / config-5 sub-records and the end-to-end leg")
    ap.add_argument("--pmc", choices=["auto", "live", "replay", "off"], default="auto",
                    help="PMC counters for the roofline records: live = rocprofv3 passes over child runs (auto: live at N = 1 when "
                         "rocprofv3 exists, else the committed profile)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--hard", type=int, default=0, help="--workload service: services (every third) with a hard zone constraint on their own pods (maxSkew 2)")
    ap.add_argument("--pref", type=int, default=0, help="--workload service: services whose pods prefer not to sit next to their own kind (hostname 100, zone 50)")
    ap.add_argument("--gpu-services", type=int, default=0, help="--workload service: services whose pods ask for GPU share (a gpushare cluster behind Services)")
    ap.add_argument("--shapes", type=int, default=0, help="--workload service: distinct node shapes of the existing nodes (x 3 zones = internal node classes of generation 7)")
    ap.add_argument("--anti", type=int, default=0, help="--workload service: services whose pods also require anti-affinity to their own kind on the hostname key")
    ap.add_argument("--workload", choices=["config3", "config5", "config5service", "config5asdrawn", "config2", "config3sig", "config3classes", "service", "typical", "widemix"], default="config3",
                    help="config3 = the BASELINE metric's workload (default); config5 = gpushare-style 50k pods x 5k nodes "
                         "(GPU share + anti-affinity + taints) on generation 6 of the score-table kernel, --c5-scenarios per GPU")
    ap.add_argument("--placement", type=int, default=1, help="store the [S][P] placement matrix in HBM (default on)")
    ap.add_argument("--strong", action="store_true", help="strong scaling: BASELINE config 3's fixed batch (--orders-per-gpu orders in total, 4 096 scenarios) "
                                                          "split over the N ranks instead of a batch that grows with N (the default, weak: config 4 at N = 8)")
    ap.add_argument("--group", type=int, default=0, help="N > 0: ONE process driving N devices through simon_group_* (the Go-host shape) "
                                                          "instead of one rank per GPU")
    args = ap.parse_args()

    spawning = args.group <= 0 and "WORLD_SIZE" not in os.environ and args.gpus > 1 and not args.pmc_child
    if not spawning:                                          # (the self-spawning parent leaves fd 1 to its ranks)
        claim_stdout()
    if args.group > 0:
        import torch
        from open_simulator_amd import capi, synth
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
        return run_group(args, capi, synth, torch)

    if "WORLD_SIZE" not in os.environ and args.gpus > 1 and not args.pmc_child:
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist
    from open_simulator_amd import capi, sweep, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print(f"[bench] --gpus {args.gpus} but the launcher started {world} rank(s); reporting n_gpus = {world}", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    # test hook (tests/test_gpu_parity.py::test_bench_*_share_one_gpu): several ranks on ONE device over gloo, to exercise the
    # world > 1 code path on a single-GPU box; the real launch is one rank per GPU over RCCL ("nccl")
    backend = os.environ.get("SIMON_BENCH_BACKEND", "nccl")
    if os.environ.get("SIMON_BENCH_SHARE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    selfcheck = None
    # SIMON_BENCH_FORCE_DIST=1 (test hook): a launch of ONE rank initialises the process group all the same and runs every collective
    # of the multi-GPU launch (self-check all-gather, plan all-gather, barriers, max over ranks) -- on a one-GPU box that is how the
    # RCCL calls execute at all; the line says `ranks.forced_one_rank_group`
    use_dist = world > 1 or os.environ.get("SIMON_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29513")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        selfcheck = multi_rank_selfcheck(torch, dist, world, rank, local_rank, backend)

    (prob, scen_all, orders), n_orders = build_workload(args, synth, world)
    scen = sweep.shard(scen_all, rank, world)                 # every rank gets every node count
    S_local, S_total = len(scen), len(scen_all)
    placement = bool(args.placement)

    ctx = capi.Context(local_rank)
    ctx.load_problem(prob)
    ctx.load_scenarios(scen, orders)                          # inputs resident in HBM before timing
    best = [None]
    gather_s = [0.0]

    def after_step(plan):                                     # device-side reduction of this rank's batch, then
        rec = sweep.plan_record(bool(plan.found), plan.n_nodes, plan.scenario, plan.order_id, rank, world)
        t0 = time.perf_counter()
        best[0] = sweep.all_gather_plan(rec, device="cuda" if backend == "nccl" else "cpu", force=use_dist).as_list()   # RCCL all-gather, 32 B per rank (no-op at N=1)
        gather_s[0] += time.perf_counter() - t0

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    dt, k_ms = time_steps(ctx, args.steps, args.warmup, placement, fence, after_step)
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if args.pmc_child:                                        # profiled child of pmc_live(): the kernels ran, nothing to print
        ctx.close()
        return

    rc = 0
    if rank == 0:
        st = ctx.stats()
        wl = "config5" if args.workload in ("config5", "config5service") else "config3"
        alg = algorithmic_bytes(scen, prob.n_pods, wl)
        kname = kernel_of(st)
        value = S_total * args.steps / dt
        out = {
            "metric": "capacity-plan scenarios/sec (10k pods x ~1k nodes)", "value": round(value, 3),
            "unit": "scenarios/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong" if args.strong else "weak",
            "vs_baseline": None, "dtype": DTYPE.get(st.kernel_variant, "int64 + f64"), "data": "synthetic",
            "pods_placed_per_sec": round(value * prob.n_pods, 1),
            "config": {"workload": workload_name(args, prob, scen_all, n_orders, S_local, world),
                       "scenarios_per_gpu": S_local, "pods": prob.n_pods, "node_pool": prob.n_nodes,
                       "placement_matrix": placement, "kernel": KERNEL_SHORT.get(st.kernel_variant, "?"), "kernel_generation": st.kernel_generation,
                       "workgroup": st.workgroup_size, "slots_per_lane": st.slots_per_lane, "plan": best[0]},
            "ranks": {"world_size": dist.get_world_size(),
                      "backend": dist.get_backend(),
                      "collective": "all_gather of one 32-byte plan record per rank and step",
                      "all_gather_ms_per_step": round(gather_s[0] / max(args.steps, 1) * 1e3, 4),
                      "selfcheck": selfcheck, **({"forced_one_rank_group": True} if world == 1 else {}),
                      "launched_by": "bench.py --gpus N (self-spawned torch.distributed.run)" if os.environ.get("SIMON_BENCH_SELF_SPAWNED") else "external launcher"} if use_dist else
                     {"world_size": 1, "backend": None},
        }
        # ---- roofline of the dominant kernel ---------------------------------------------------------------------
        pmc, source = None, None
        mode = args.pmc
        if mode in ("auto", "live") and world == 1:
            child = ["--workload", args.workload, "--steps", "1", "--warmup", "0", "--counts", str(args.counts), "--pods", str(args.pods),
                     "--orders-per-gpu", str(args.orders_per_gpu), "--sigs", str(args.sigs), "--classes", str(args.classes), "--placement", str(args.placement),
                     "--c5-scenarios", str(c5_scenarios(args))]
            ctx.close()                                        # free the device for the profiled children
            t0 = time.perf_counter()
            pmc = pmc_live(kname, child, budget_s=float(os.environ.get("SIMON_BENCH_PMC_BUDGET_S", "240")))
            if pmc:
                source = (f"measured live in this run: rocprofv3 --kernel-trace --pmc, {len(PMC_GROUPS)} separate passes over a 1-step child run "
                          f"of the same workload ({time.perf_counter() - t0:.0f} s); per-dispatch averages x launches per step")
            ctx = capi.Context(local_rank)                     # the parity legs below need the batch's results again
            ctx.load_problem(prob)
            ctx.load_scenarios(scen, orders)
            ctx.run_loaded(want_placement=placement)
        if pmc is None and mode != "off":
            pmc, src = pmc_replayed(kname, S_local, prob.n_pods)
            if pmc:
                source = f"replayed from the committed profile ({src}), not measured in this run"
        out["roofline"] = roofline_record(kname, st.kernel_variant, k_ms, st.n_launches, alg, pmc, source, wl,
                                          lds_bytes=st.lds_bytes if st.kernel_variant == 4 else None, scenarios=S_local)
        # ---- the oracle: CPU baseline (N = 1) and parity of the timed batch ---------------------------------------
        if not args.no_cpu_baseline:
            if world == 1:
                pick, refs, tm = oracle_sample(prob, scen, orders, budget_s=float(os.environ.get("SIMON_BENCH_CPU_BUDGET_S", "12")))
                out["cpu_baseline"] = cpu_baseline_record(tm, S_local, "this rank's batch")
            else:
                pick, refs, tm = oracle_sample(prob, scen, orders, budget_s=4.0, max_k=64)
            out["parity_sample"] = parity_check(ctx, pick, refs, placement)
            if out["parity_sample"]["mismatches"]:
                rc = 3
        ctx.close()
        # ---- driver-timed records of the other single-GPU configurations, each a measurement of its own --------------
        if world == 1 and args.workload == "config3" and not args.no_sub:
            out["end_to_end"] = end_to_end(capi, torch, prob, scen, orders, local_rank)
            subs = []
            nchk5 = int(os.environ.get("SIMON_BENCH_C5_CHECK", "32"))
            sub_steps = int(os.environ.get("SIMON_BENCH_SUB_STEPS", "5"))       # every sub-record times >= 5 steps after a warm-up
            for name, steps, warm, nchk, c5s in (("config2", 20, 2, 1, 0), ("config3sig", sub_steps, 1, 64, 0), ("service", sub_steps, 1, 64, 0),
                                                 ("service_anti", sub_steps, 1, 48, 0), ("service_pref", sub_steps, 1, 48, 0),
                                                 ("config5", sub_steps, 1, nchk5, c5_scenarios(args)), ("config5", sub_steps, 1, nchk5, C5_SATURATING),
                                                 ("service_small", sub_steps, 1, 16, 0), ("service_gpu", sub_steps, 1, 16, 0), ("service_shapes", sub_steps, 1, 16, 0), ("config5_service", sub_steps, 1, 16, c5_scenarios(args)), ("config5_asdrawn", sub_steps, 1, 16, c5_scenarios(args)), ("typical", sub_steps, 1, 2, 0), ("widemix", sub_steps, 1, 2, 0),
                                                 ("config3sig_cliff", sub_steps, 1, 16, 0), ("config3_classes", sub_steps, 1, 16, 0), ("config3_classes_cliff", sub_steps, 1, 16, 0),
                                                 ("config3_small", sub_steps, 1, 16, 0)):
                try:
                    cliff = name in ("config3sig_cliff", "config3_classes", "config3_classes_cliff", "config5_service", "config5_asdrawn")   # the cliff rows and config 5 behind Services: a shorter CPU sample; since round 5 they carry their live-PMC roofline like every other row
                    subs.append(sub_record(name, capi, synth, torch, steps, warm, 0 if args.no_cpu_baseline else nchk, mode, c5s,
                                           cpu_budget_s=3.0 if cliff else 6.0))
                    if subs[-1].get("parity_sample", {}).get("mismatches"):
                        rc = 3
                except Exception as e:                         # noqa: BLE001
                    subs.append({"workload": name, "error": repr(e)})
                    rc = rc or 4
            out["other_workloads"] = subs
        emit(out)
    else:
        ctx.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rc:
        raise SystemExit(rc)


if __name__ == "__main__":
    main()
