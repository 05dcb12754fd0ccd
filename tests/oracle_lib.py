"""Test-side loader of the CPU oracle (oracle/libsimon_oracle.so).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

from open_simulator_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_LIB = None


class OQuantity(C.Structure):
    _fields_ = [("value", C.c_int64), ("scale", C.c_int32)]


def load():
    global _LIB
    if _LIB is not None:
        return _LIB
    so = os.path.join(ORACLE_DIR, "libsimon_oracle.so")
    src = os.path.join(ORACLE_DIR, "simon_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    lib = C.CDLL(so)
    p32, p64, pu16 = C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_uint16)
    lib.simon_oracle_run.restype = C.c_int
    lib.simon_oracle_run.argtypes = [C.POINTER(capi.NodesSoA), C.POINTER(capi.PodsSoA), C.POINTER(capi.ClassTables),
                                     C.POINTER(capi.Scenario), C.c_int32, p32, C.c_int32, C.POINTER(capi.BatchOut),
                                     C.c_int32, p32, pu16, C.c_int32, p32]
    lib.simon_oracle_score_pod.restype = C.c_int
    lib.simon_oracle_score_pod.argtypes = [C.POINTER(capi.NodesSoA), C.POINTER(capi.PodsSoA),
                                           C.POINTER(capi.ClassTables), C.c_int32, C.c_int32, p64, p64, p64, p64, p64]
    lib.simon_oracle_score_pod_after.restype = C.c_int
    lib.simon_oracle_score_pod_after.argtypes = [C.POINTER(capi.NodesSoA), C.POINTER(capi.PodsSoA),
                                                 C.POINTER(capi.ClassTables), C.c_int32, p32, C.c_int32, C.c_int32,
                                                 p64, p64, p64, p64, p64, pu16]
    lib.simon_oracle_min_plan.restype = C.c_int
    lib.simon_oracle_min_plan.argtypes = [C.POINTER(capi.NodesSoA), C.POINTER(capi.Scenario), C.c_int32,
                                          C.POINTER(capi.BatchOut), C.c_int32, C.c_int32, C.POINTER(capi.Plan)]
    lib.simon_oracle_simon_raw.restype = C.c_int64
    lib.simon_oracle_simon_raw.argtypes = [C.POINTER(OQuantity), C.POINTER(OQuantity), C.c_int32, C.c_int32]
    lib.simon_oracle_splitmix64.restype = C.c_uint64
    lib.simon_oracle_splitmix64.argtypes = [C.POINTER(C.c_uint64)]
    lib.simon_oracle_gen_nodes.argtypes = [C.c_uint64, C.c_int32, C.c_int32, p64, p64, p32, p32]
    lib.simon_oracle_gen_pods.argtypes = [C.c_uint64, C.c_int32, p64, p64]
    _LIB = lib
    return lib


LAST_LOCAL_DETAIL = [None]        # [k][n][4] of the last explaining run() (None without Open-Local in the problem)


def run(prob: capi.Problem, scen, orders, want_placement=True, explain_scenario=-1, max_failed=0, node_ranks=None, want_gpu_slices=False):
    """Run scenarios on the oracle; returns BatchResult (+ (n_failed, failed_pods, codes) when explaining).
    Thread-safe without node_ranks (the ranked entry keeps the current scenario's ranks in a global)."""
    lib = load()
    prob.normalise()
    scen = capi.scenarios_array(scen)
    orders = np.ascontiguousarray(orders, np.int32).reshape(-1, prob.n_pods)
    res = capi.BatchResult.alloc(len(scen), prob.n_pods, want_placement, want_gpu_slices)
    out = res.c_out()
    n, p, t = prob.c_nodes(), prob.c_pods(), prob.c_tables()
    ranks = None
    if node_ranks is not None:
        ranks = np.ascontiguousarray(node_ranks, np.int32)
        assert ranks.shape == (len(scen), prob.n_nodes)
    lib.simon_oracle_run_ranked.restype = C.c_int
    # ABI v6 inputs that travel outside simon_pods_soa: per calling thread, for the run below (res.preempt_risk: the oracle's twin of
    # simon_fetch_preempt_risk)
    lib.simon_oracle_set_v6.argtypes = [C.POINTER(C.c_uint8), C.POINTER(C.c_int32), C.c_int32, C.POINTER(C.c_uint8)]
    lib.simon_oracle_set_v6.restype = None
    risk = np.zeros(len(scen), np.uint8)
    lib.simon_oracle_set_v6(capi._ptr(prob.scalar_entries, C.c_uint8), capi._ptr(prob.priority, C.c_int32), int(prob.init_min_priority),
                            capi._ptr(risk, C.c_uint8))
    res.preempt_risk = risk
    try:
        return _run_set(lib, prob, scen, orders, res, out, n, p, t, ranks, explain_scenario, max_failed)
    finally:
        lib.simon_oracle_set_v6(None, None, 0x7fffffff, None)


def _run_set(lib, prob, scen, orders, res, out, n, p, t, ranks, explain_scenario, max_failed):
    if explain_scenario >= 0:
        nmax = int(scen[explain_scenario, 0])
        failed = np.full(max_failed, -1, np.int32)
        codes = np.zeros((max_failed, nmax), np.uint16)
        nf = C.c_int32(0)
        detail = np.zeros((max_failed, nmax, 4), np.int64)          # what Open-Local's error texts carry (simon_explain_local_detail)
        lib.simon_oracle_set_local_detail.argtypes = [C.c_void_p]
        lib.simon_oracle_set_local_detail(detail.ctypes.data_as(C.c_void_p) if prob.local_flags is not None and max_failed > 0 else None)
        try:
            rc = lib.simon_oracle_run_ranked(C.byref(n), C.byref(p), C.byref(t), scen.ctypes.data_as(C.POINTER(capi.Scenario)),
                                             C.c_int32(len(scen)), capi._ptr(orders, C.c_int32), C.c_int32(orders.shape[0]),
                                             capi._ptr(ranks, C.c_int32), C.byref(out), C.c_int32(explain_scenario),
                                             capi._ptr(failed, C.c_int32), capi._ptr(codes, C.c_uint16), C.c_int32(max_failed), C.byref(nf))
        finally:
            lib.simon_oracle_set_local_detail(None)
        assert rc == 0, rc
        k = min(nf.value, max_failed)
        LAST_LOCAL_DETAIL[0] = detail[:k] if prob.local_flags is not None else None
        return res, (nf.value, failed[:k], codes[:k])
    if ranks is not None:
        rc = lib.simon_oracle_run_ranked(C.byref(n), C.byref(p), C.byref(t), scen.ctypes.data_as(C.POINTER(capi.Scenario)),
                                         C.c_int32(len(scen)), capi._ptr(orders, C.c_int32), C.c_int32(orders.shape[0]),
                                         capi._ptr(ranks, C.c_int32), C.byref(out), C.c_int32(-1), None, None, C.c_int32(0), None)
        assert rc == 0, rc
        return res
    rc = lib.simon_oracle_run(C.byref(n), C.byref(p), C.byref(t), scen.ctypes.data_as(C.POINTER(capi.Scenario)),
                              len(scen), capi._ptr(orders, C.c_int32), orders.shape[0], C.byref(out), -1, None, None,
                              0, None)
    assert rc == 0, rc
    return res


def score_pod(prob: capi.Problem, n_nodes: int, pod: int):
    lib = load()
    prob.normalise()
    n, p, t = prob.c_nodes(), prob.c_pods(), prob.c_tables()
    arrs = [np.zeros(n_nodes, np.int64) for _ in range(5)]
    best = lib.simon_oracle_score_pod(C.byref(n), C.byref(p), C.byref(t), n_nodes, pod,
                                      *[capi._ptr(a, C.c_int64) for a in arrs])
    return best, dict(zip(["feasible", "la", "ba", "sn", "total"], arrs))


def score_pod_after(prob: capi.Problem, n_nodes: int, n_before: int, pod: int, order=None):
    """Schedule the first n_before pods of `order` (identity by default) on the oracle, then score `pod`."""
    lib = load()
    prob.normalise()
    n, p, t = prob.c_nodes(), prob.c_pods(), prob.c_tables()
    arrs = [np.zeros(n_nodes, np.int64) for _ in range(5)]
    codes = np.zeros(n_nodes, np.uint16)
    order = None if order is None else np.ascontiguousarray(order, np.int32)
    best = lib.simon_oracle_score_pod_after(C.byref(n), C.byref(p), C.byref(t), n_nodes, capi._ptr(order, C.c_int32),
                                            n_before, pod, *[capi._ptr(a, C.c_int64) for a in arrs],
                                            capi._ptr(codes, C.c_uint16))
    out = dict(zip(["feasible", "la", "ba", "sn", "total"], arrs))
    out["codes"] = codes
    return best, out


def min_plan_vg(prob: capi.Problem, scen, res: capi.BatchResult, max_cpu=100, max_mem=100, max_vg=100):
    lib = load()
    scen = capi.scenarios_array(scen)
    n = prob.c_nodes()
    out = res.c_out()
    plan, vg = capi.Plan(), C.c_int32(0)
    lib.simon_oracle_min_plan_vg.restype = C.c_int
    rc = lib.simon_oracle_min_plan_vg(C.byref(n), scen.ctypes.data_as(C.POINTER(capi.Scenario)), C.c_int32(len(scen)),
                                      C.byref(out), C.c_int32(max_cpu), C.c_int32(max_mem), C.c_int32(max_vg), C.byref(plan), C.byref(vg))
    assert rc == 0
    return plan, int(vg.value)


def min_plan(prob: capi.Problem, scen, res: capi.BatchResult, max_cpu=100, max_mem=100) -> capi.Plan:
    lib = load()
    scen = capi.scenarios_array(scen)
    n = prob.c_nodes()
    out = res.c_out()
    plan = capi.Plan()
    rc = lib.simon_oracle_min_plan(C.byref(n), scen.ctypes.data_as(C.POINTER(capi.Scenario)), len(scen),
                                   C.byref(out), max_cpu, max_mem, C.byref(plan))
    assert rc == 0
    return plan


def run_threaded(prob: capi.Problem, scen, orders, want_placement=True, threads=None, want_gpu_slices=False) -> capi.BatchResult:
    """The oracle over many scenarios, one scenario per task on `threads` host threads (the C call releases the GIL;
    scenarios are independent).  Same result layout as run()."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    load()
    prob.normalise()
    scen = capi.scenarios_array(scen)
    threads = threads or host_threads()
    res = capi.BatchResult.alloc(len(scen), prob.n_pods, want_placement, want_gpu_slices)

    def one(i):
        r = run(prob, scen[[i]], orders, want_placement, want_gpu_slices=want_gpu_slices)
        if want_gpu_slices:
            res.gpu_slices[i] = r.gpu_slices[0]
        res.unscheduled[i] = r.unscheduled[0]
        res.used_cpu[i] = r.used_cpu[0]
        res.used_mem[i] = r.used_mem[0]
        if res.used_vg is not None and r.used_vg is not None:
            res.used_vg[i] = r.used_vg[0]
        if want_placement:
            res.placement[i] = r.placement[0]
    with ThreadPoolExecutor(max(1, threads)) as pool:
        list(pool.map(one, range(len(scen))))
    return res


def host_threads() -> int:
    """Host threads worth using: the affinity mask capped by the container's CPU quota."""
    import os
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for qf in ("/sys/fs/cgroup/cpu.max",):
        try:
            quota, period = open(qf).read().split()
            if quota != "max":
                cores = max(1, min(cores, -(-int(quota) // int(period))))
        except (OSError, ValueError):
            pass
    return max(1, cores)
