"""ctypes binding of include/simon_hip.h (the C-ABI of libsimon_hip.so).

This is the Python twin of the cgo stub shown in INTEGRATION.md: plain pointers and sizes, no torch
types.  The struct layouts below mirror include/simon_hip.h field for field; `Problem` is a
numpy-side container that keeps the arrays alive while the C structs point into them.

The product path fails loudly when the HIP library is missing: `load_library()` raises, there is no
CPU fallback (the oracle under oracle/ is test infrastructure and is never imported from here).
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

ABI_VERSION = 6
MAX_GPU_DEV = 8
MAX_SCALAR = 4

UNSCHEDULED = -1
GATED = -2

FAIL_STATIC = 0x8000
REASON_NODE_AFFINITY = 3
FAIL_FIT = 0x4000
FIT_PODS, FIT_CPU, FIT_MEM, FIT_EPH, FIT_SCALAR0 = 0x1, 0x2, 0x4, 0x8, 0x10
FAIL_ANTI_INCOMING = 0x2001
FAIL_ANTI_EXISTING = 0x2002
FAIL_AFFINITY = 0x2003
FAIL_SPREAD = 0x2010
FAIL_SPREAD_LABEL = 0x2011
FAIL_GPUSHARE = 0x1000
FAIL_PORTS = 0x0800
CLASS_AFF_SELF = 0x1
MAX_SPREAD = 4
MAX_VG, MAX_LDEV, MAX_LVOL = 4, 8, 4
FAIL_LOCAL, FAIL_LOCAL_LVM, FAIL_LOCAL_DEV = 0x0400, 0x0401, 0x0402
LOCAL_ERR_NONE, LOCAL_ERR_NO_SUCH_VG, LOCAL_ERR_NO_VG, LOCAL_ERR_LVM, LOCAL_ERR_DEVICE = 0, 1, 2, 3, 4   # simon_explain_local_detail
SPREAD_DUP_KEY = 0x40000000

KERNEL_NARROW = 1
KERNEL_WIDE = 2
KERNEL_NARROW_FAST = 3
KERNEL_NARROW_CACHE = 4

_p64 = C.POINTER(C.c_int64)
_p32 = C.POINTER(C.c_int32)
_pu64 = C.POINTER(C.c_uint64)
_pu8 = C.POINTER(C.c_uint8)
_pu16 = C.POINTER(C.c_uint16)
_pf64 = C.POINTER(C.c_double)


class NodesSoA(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("n_nodes", C.c_int32),
        ("alloc_cpu", _p64), ("alloc_mem", _p64), ("alloc_eph", _p64), ("alloc_pods", _p32),
        ("init_req_cpu", _p64), ("init_req_mem", _p64), ("init_req_eph", _p64),
        ("init_nz_cpu", _p64), ("init_nz_mem", _p64), ("init_npods", _p32),
        ("node_class", _p32),
        ("n_scalar", C.c_int32), ("scalar_alloc", _p64), ("init_scalar_req", _p64),
        ("gpu_cnt", _p32), ("gpu_mem_total", _p64), ("init_gpu_used", _p64),
        ("local_flags", _p32), ("local_vg_cnt", _p32), ("local_vg_cap", _p64), ("init_vg_req", _p64), ("local_vg_name", _p32),
        ("local_dev_cnt", _p32), ("local_dev_cap", _p64), ("local_dev_media", _p32), ("init_dev_alloc", _p32),
        ("n_topo_keys", C.c_int32), ("topo_dom", _p32), ("topo_n_dom", _p32),
    ]


class PodsSoA(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("n_pods", C.c_int32),
        ("req_cpu", _p64), ("req_mem", _p64), ("req_eph", _p64), ("nz_cpu", _p64), ("nz_mem", _p64),
        ("scalar_req", _p64), ("pod_class", _p32), ("preset_node", _p32), ("gate_node", _p32),
        ("gpu_mem", _p64), ("gpu_cnt", _p32), ("pin_node", _p32), ("gpu_index", C.POINTER(C.c_uint32)),
    ]


class LocalSpec(C.Structure):
    _fields_ = [("n_lvm", C.c_int32), ("n_ssd", C.c_int32), ("n_hdd", C.c_int32), ("pad", C.c_int32),
                ("lvm_size", C.c_int64 * MAX_LVOL), ("lvm_vg", C.c_int32 * MAX_LVOL),
                ("ssd_size", C.c_int64 * MAX_LVOL), ("hdd_size", C.c_int64 * MAX_LVOL)]


LOCAL_SPEC_DTYPE = np.dtype([("n_lvm", "<i4"), ("n_ssd", "<i4"), ("n_hdd", "<i4"), ("pad", "<i4"), ("lvm_size", "<i8", (MAX_LVOL,)),
                             ("lvm_vg", "<i4", (MAX_LVOL,)), ("ssd_size", "<i8", (MAX_LVOL,)), ("hdd_size", "<i8", (MAX_LVOL,))])
assert LOCAL_SPEC_DTYPE.itemsize == C.sizeof(LocalSpec)


class ClassTables(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("n_pod_classes", C.c_int32), ("n_node_classes", C.c_int32),
        ("static_mask", _pu64), ("static_reason", _pu8), ("simon_raw", _p64), ("const_score", _p64),
        ("node_affinity_raw", _p64), ("taint_prefer_raw", _p64), ("static_add", _p64),
        ("n_terms", C.c_int32), ("term_topo_key", _p32), ("term_node_set", _p32),
        ("n_node_sets", C.c_int32), ("node_sets", _pu64),
        ("match_off", _p32), ("match_idx", _p32), ("anti_off", _p32), ("anti_idx", _p32),
        ("port_off", _p32), ("port_idx", _p32),
        ("aff_off", _p32), ("aff_idx", _p32), ("class_flags", _pu8),
        ("pref_off", _p32), ("pref_idx", _p32), ("pref_w", _p32),
        ("own_off", _p32), ("own_idx", _p32), ("own_w", _p32),
        ("spread_hard_off", _p32), ("spread_hard_idx", _p32), ("spread_hard_skew", _p32),
        ("spread_hard_self", _p32), ("spread_hard_set", _p32),
        ("spread_soft_off", _p32), ("spread_soft_idx", _p32), ("spread_soft_skew", _p32),
        ("local_spec_of", _p32), ("n_local_specs", C.c_int32), ("local_specs", C.POINTER(LocalSpec)),
        ("topo_is_hostname", _pu8), ("spread_log", _pf64),
    ]


class Scenario(C.Structure):
    _fields_ = [("n_nodes", C.c_int32), ("order_id", C.c_int32)]


class BatchOut(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("flags", C.c_uint32),
        ("unscheduled", _p32), ("used_cpu", _p64), ("used_mem", _p64), ("placement", _p32), ("used_vg", _p64),
        ("gpu_slices", _pu64),
    ]


class Plan(C.Structure):
    _fields_ = [
        ("found", C.c_int32), ("scenario", C.c_int32), ("n_nodes", C.c_int32), ("order_id", C.c_int32),
        ("cpu_pct", C.c_int32), ("mem_pct", C.c_int32), ("used_cpu", C.c_int64), ("used_mem", C.c_int64),
    ]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class Stats(C.Structure):
    _fields_ = [
        ("kernel_ms", C.c_double), ("h2d_ms", C.c_double), ("d2h_ms", C.c_double),
        ("n_launches", C.c_int32), ("kernel_variant", C.c_int32), ("workgroup_size", C.c_int32),
        ("slots_per_lane", C.c_int32), ("lds_bytes", C.c_int64), ("kernel_generation", C.c_int32), ("reserved", C.c_int32),
    ]


def _arr(a, dtype, shape=None) -> Optional[np.ndarray]:
    if a is None:
        return None
    a = np.ascontiguousarray(a, dtype=dtype)
    if shape is not None and tuple(a.shape) != tuple(shape):
        raise ValueError(f"expected shape {shape}, got {a.shape}")
    return a


def _ptr(a: Optional[np.ndarray], ctype):
    if a is None:
        return C.cast(None, C.POINTER(ctype))
    return a.ctypes.data_as(C.POINTER(ctype))


@dataclass
class Problem:
    """Numpy-side mirror of (simon_nodes_soa, simon_pods_soa, simon_class_tables).

    Field meaning is documented in include/simon_hip.h; None = optional array omitted.
    """
    # nodes
    alloc_cpu: np.ndarray = None
    alloc_mem: np.ndarray = None
    alloc_pods: np.ndarray = None
    alloc_eph: Optional[np.ndarray] = None
    init_req_cpu: Optional[np.ndarray] = None
    init_req_mem: Optional[np.ndarray] = None
    init_req_eph: Optional[np.ndarray] = None
    init_nz_cpu: Optional[np.ndarray] = None
    init_nz_mem: Optional[np.ndarray] = None
    init_npods: Optional[np.ndarray] = None
    node_class: Optional[np.ndarray] = None
    scalar_alloc: Optional[np.ndarray] = None      # [K][N]
    init_scalar_req: Optional[np.ndarray] = None   # [K][N]
    gpu_cnt: Optional[np.ndarray] = None
    gpu_mem_total: Optional[np.ndarray] = None
    init_gpu_used: Optional[np.ndarray] = None     # [N][8]
    local_flags: Optional[np.ndarray] = None       # [N]
    local_vg_cnt: Optional[np.ndarray] = None
    local_vg_cap: Optional[np.ndarray] = None      # [N][4]
    init_vg_req: Optional[np.ndarray] = None
    local_vg_name: Optional[np.ndarray] = None     # [N][4]
    local_dev_cnt: Optional[np.ndarray] = None
    local_dev_cap: Optional[np.ndarray] = None     # [N][8]
    local_dev_media: Optional[np.ndarray] = None
    init_dev_alloc: Optional[np.ndarray] = None
    topo_dom: Optional[np.ndarray] = None          # [Kt][N]
    topo_n_dom: Optional[np.ndarray] = None        # [Kt]
    # pods
    req_cpu: np.ndarray = None
    req_mem: np.ndarray = None
    req_eph: Optional[np.ndarray] = None
    nz_cpu: Optional[np.ndarray] = None
    nz_mem: Optional[np.ndarray] = None
    scalar_req: Optional[np.ndarray] = None        # [K][P]
    pod_class: Optional[np.ndarray] = None
    preset_node: Optional[np.ndarray] = None
    gate_node: Optional[np.ndarray] = None
    pin_node: Optional[np.ndarray] = None          # [P] >= 0: the only node the pod's node affinity admits (DaemonSet pods)
    gpu_mem: Optional[np.ndarray] = None
    pod_gpu_cnt: Optional[np.ndarray] = None
    gpu_index: Optional[np.ndarray] = None         # [P] uint32: gpu-index annotation the pod arrives with, packed (pack_gpu_index)
    scalar_entries: Optional[np.ndarray] = None    # [P] uint8, ABI v6: bit k = the request holds an ENTRY for extended resource k (bit 7: one no node tracks), a zero quantity included
    priority: Optional[np.ndarray] = None          # [P] int32, ABI v6: spec.priority (None: all equal)
    init_min_priority: int = 0x7fffffff            # lowest priority among the pods bound before the stream (behind init_*)
    # class tables
    n_pod_classes: int = 1
    n_node_classes: int = 1
    static_mask: Optional[np.ndarray] = None       # [Cp][ceil(N/64)] uint64
    static_reason: Optional[np.ndarray] = None     # [Cp][N] uint8
    simon_raw: Optional[np.ndarray] = None         # [Cp][Cn]
    const_score: Optional[np.ndarray] = None       # [Cp]
    node_affinity_raw: Optional[np.ndarray] = None  # [Cp][Cn]
    taint_prefer_raw: Optional[np.ndarray] = None   # [Cp][Cn]
    static_add: Optional[np.ndarray] = None         # [Cp][Cn]
    term_topo_key: Optional[np.ndarray] = None     # [T]
    term_node_set: Optional[np.ndarray] = None     # [T]
    node_sets: Optional[np.ndarray] = None         # [R][ceil(N/64)] uint64
    anti_off: Optional[np.ndarray] = None
    anti_idx: Optional[np.ndarray] = None
    match_off: Optional[np.ndarray] = None
    match_idx: Optional[np.ndarray] = None
    port_off: Optional[np.ndarray] = None
    port_idx: Optional[np.ndarray] = None
    aff_off: Optional[np.ndarray] = None
    aff_idx: Optional[np.ndarray] = None
    class_flags: Optional[np.ndarray] = None       # [Cp] uint8
    pref_off: Optional[np.ndarray] = None
    pref_idx: Optional[np.ndarray] = None
    pref_w: Optional[np.ndarray] = None
    own_off: Optional[np.ndarray] = None
    own_idx: Optional[np.ndarray] = None
    own_w: Optional[np.ndarray] = None
    spread_hard_off: Optional[np.ndarray] = None
    spread_hard_idx: Optional[np.ndarray] = None
    spread_hard_skew: Optional[np.ndarray] = None
    spread_hard_self: Optional[np.ndarray] = None
    spread_hard_set: Optional[np.ndarray] = None
    spread_soft_off: Optional[np.ndarray] = None
    spread_soft_idx: Optional[np.ndarray] = None
    spread_soft_skew: Optional[np.ndarray] = None
    local_spec_of: Optional[np.ndarray] = None     # [Cp]
    local_specs: Optional[np.ndarray] = None       # structured array LOCAL_SPEC_DTYPE
    topo_is_hostname: Optional[np.ndarray] = None  # [Kt] uint8
    spread_log: Optional[np.ndarray] = None        # [N+1] float64
    _keep: list = field(default_factory=list, repr=False)

    @property
    def n_nodes(self) -> int:
        return int(len(self.alloc_cpu))

    @property
    def n_pods(self) -> int:
        return int(len(self.req_cpu))

    def normalise(self) -> "Problem":
        N, P = self.n_nodes, self.n_pods
        i64, i32 = np.int64, np.int32
        self.alloc_cpu = _arr(self.alloc_cpu, i64, (N,))
        self.alloc_mem = _arr(self.alloc_mem, i64, (N,))
        self.alloc_pods = _arr(self.alloc_pods, i32, (N,))
        for name in ("alloc_eph", "init_req_cpu", "init_req_mem", "init_req_eph", "init_nz_cpu", "init_nz_mem",
                     "gpu_mem_total"):
            setattr(self, name, _arr(getattr(self, name), i64, (N,)))
        for name in ("init_npods", "node_class", "gpu_cnt"):
            setattr(self, name, _arr(getattr(self, name), i32, (N,)))
        K = 0
        if self.scalar_alloc is not None:
            self.scalar_alloc = _arr(self.scalar_alloc, i64)
            K = self.scalar_alloc.shape[0]
            assert self.scalar_alloc.shape == (K, N) and K <= MAX_SCALAR
        self.init_scalar_req = _arr(self.init_scalar_req, i64, (K, N)) if self.init_scalar_req is not None else None
        self.init_gpu_used = _arr(self.init_gpu_used, i64, (N, MAX_GPU_DEV)) if self.init_gpu_used is not None else None
        if self.local_flags is not None:
            self.local_flags = _arr(self.local_flags, i32, (N,))
            self.local_vg_cnt = _arr(self.local_vg_cnt if self.local_vg_cnt is not None else np.zeros(N), i32, (N,))
            self.local_vg_cap = _arr(self.local_vg_cap if self.local_vg_cap is not None else np.zeros((N, MAX_VG)), i64, (N, MAX_VG))
            self.init_vg_req = _arr(self.init_vg_req, i64, (N, MAX_VG)) if self.init_vg_req is not None else None
            self.local_vg_name = _arr(self.local_vg_name if self.local_vg_name is not None else np.zeros((N, MAX_VG)), i32, (N, MAX_VG))
            self.local_dev_cnt = _arr(self.local_dev_cnt if self.local_dev_cnt is not None else np.zeros(N), i32, (N,))
            self.local_dev_cap = _arr(self.local_dev_cap if self.local_dev_cap is not None else np.zeros((N, MAX_LDEV)), i64, (N, MAX_LDEV))
            self.local_dev_media = _arr(self.local_dev_media if self.local_dev_media is not None else np.zeros(N), i32, (N,))
            self.init_dev_alloc = _arr(self.init_dev_alloc, i32, (N,)) if self.init_dev_alloc is not None else None
        Kt = 0
        if self.topo_dom is not None:
            self.topo_dom = _arr(self.topo_dom, i32)
            Kt = self.topo_dom.shape[0]
            assert self.topo_dom.shape == (Kt, N)
            self.topo_n_dom = _arr(self.topo_n_dom, i32, (Kt,))
        self.req_cpu = _arr(self.req_cpu, i64, (P,))
        self.req_mem = _arr(self.req_mem, i64, (P,))
        for name in ("req_eph", "nz_cpu", "nz_mem", "gpu_mem"):
            setattr(self, name, _arr(getattr(self, name), i64, (P,)))
        for name in ("pod_class", "preset_node", "gate_node", "pod_gpu_cnt", "pin_node"):
            setattr(self, name, _arr(getattr(self, name), i32, (P,)))
        self.gpu_index = _arr(self.gpu_index, np.uint32, (P,))
        self.scalar_entries = _arr(self.scalar_entries, np.uint8, (P,))
        self.priority = _arr(self.priority, i32, (P,))
        self.scalar_req = _arr(self.scalar_req, i64, (K, P)) if self.scalar_req is not None else None
        Cp, Cn = self.n_pod_classes, self.n_node_classes
        words = (N + 63) // 64
        self.static_mask = _arr(self.static_mask, np.uint64, (Cp, words)) if self.static_mask is not None else None
        self.static_reason = _arr(self.static_reason, np.uint8, (Cp, N)) if self.static_reason is not None else None
        if self.simon_raw is None:
            self.simon_raw = np.zeros((Cp, Cn), dtype=i64)
        self.simon_raw = _arr(self.simon_raw, i64, (Cp, Cn))
        self.const_score = _arr(self.const_score, i64, (Cp,)) if self.const_score is not None else None
        for name in ("node_affinity_raw", "taint_prefer_raw", "static_add"):
            v = getattr(self, name)
            setattr(self, name, _arr(v, i64, (Cp, Cn)) if v is not None else None)
        if self.local_spec_of is not None:
            self.local_spec_of = _arr(self.local_spec_of, i32, (Cp,))
            self.local_specs = np.ascontiguousarray(self.local_specs, dtype=LOCAL_SPEC_DTYPE)
        if self.term_topo_key is not None:
            self.term_topo_key = _arr(self.term_topo_key, i32)
            T = len(self.term_topo_key)
            if self.term_node_set is not None:
                self.term_node_set = _arr(self.term_node_set, i32, (T,))
            if self.node_sets is not None:
                self.node_sets = _arr(self.node_sets, np.uint64)
                assert self.node_sets.ndim == 2 and self.node_sets.shape[1] == words
            zero_off = np.zeros(Cp + 1, i32)
            for off, cols in (("match_off", ("match_idx",)), ("anti_off", ("anti_idx",)), ("port_off", ("port_idx",)), ("aff_off", ("aff_idx",)),
                              ("pref_off", ("pref_idx", "pref_w")), ("own_off", ("own_idx", "own_w")),
                              ("spread_hard_off", ("spread_hard_idx", "spread_hard_skew", "spread_hard_self",
                                                   "spread_hard_set")),
                              ("spread_soft_off", ("spread_soft_idx", "spread_soft_skew"))):
                o = getattr(self, off)
                if o is None:
                    if off in ("match_off", "anti_off"):       # always materialised (ABI v1 compatibility of callers)
                        setattr(self, off, zero_off.copy())
                        for cname in cols:
                            setattr(self, cname, np.zeros(1, i32))
                    continue
                o = _arr(o, i32, (Cp + 1,))
                setattr(self, off, o)
                for cname in cols:
                    v = getattr(self, cname)
                    if v is None and cname == "spread_hard_set":
                        continue
                    v = _arr(v if v is not None and len(v) else np.zeros(1, i32), i32)
                    assert len(v) >= max(1, int(o[-1])), (cname, len(v), int(o[-1]))
                    setattr(self, cname, v)
            if self.class_flags is not None:
                self.class_flags = _arr(self.class_flags, np.uint8, (Cp,))
            if self.topo_is_hostname is not None:
                self.topo_is_hostname = _arr(self.topo_is_hostname, np.uint8, (Kt,))
            if self.spread_log is not None:
                self.spread_log = _arr(self.spread_log, np.float64, (N + 1,))
        return self

    # --- C views ---------------------------------------------------------------------------
    def c_nodes(self) -> NodesSoA:
        self.normalise()
        s = NodesSoA()
        s.struct_size = C.sizeof(NodesSoA)
        s.n_nodes = self.n_nodes
        for name in ("alloc_cpu", "alloc_mem", "alloc_eph", "init_req_cpu", "init_req_mem", "init_req_eph",
                     "init_nz_cpu", "init_nz_mem", "scalar_alloc", "init_scalar_req", "gpu_mem_total",
                     "init_gpu_used"):
            setattr(s, name, _ptr(getattr(self, name), C.c_int64))
        for name in ("alloc_pods", "init_npods", "node_class", "gpu_cnt", "topo_dom", "topo_n_dom", "local_flags", "local_vg_cnt",
                     "local_vg_name", "local_dev_cnt", "local_dev_media", "init_dev_alloc"):
            setattr(s, name, _ptr(getattr(self, name), C.c_int32))
        for name in ("local_vg_cap", "init_vg_req", "local_dev_cap"):
            setattr(s, name, _ptr(getattr(self, name), C.c_int64))
        s.n_scalar = 0 if self.scalar_alloc is None else self.scalar_alloc.shape[0]
        s.n_topo_keys = 0 if self.topo_dom is None else self.topo_dom.shape[0]
        return s

    def c_pods(self) -> PodsSoA:
        self.normalise()
        s = PodsSoA()
        s.struct_size = C.sizeof(PodsSoA)
        s.n_pods = self.n_pods
        for name in ("req_cpu", "req_mem", "req_eph", "nz_cpu", "nz_mem", "scalar_req", "gpu_mem"):
            setattr(s, name, _ptr(getattr(self, name), C.c_int64))
        for name in ("pod_class", "preset_node", "gate_node", "pin_node"):
            setattr(s, name, _ptr(getattr(self, name), C.c_int32))
        s.gpu_cnt = _ptr(self.pod_gpu_cnt, C.c_int32)
        s.gpu_index = _ptr(self.gpu_index, C.c_uint32)
        return s

    def c_tables(self) -> ClassTables:
        self.normalise()
        s = ClassTables()
        s.struct_size = C.sizeof(ClassTables)
        s.n_pod_classes = self.n_pod_classes
        s.n_node_classes = self.n_node_classes
        s.static_mask = _ptr(self.static_mask, C.c_uint64)
        s.static_reason = _ptr(self.static_reason, C.c_uint8)
        s.simon_raw = _ptr(self.simon_raw, C.c_int64)
        s.const_score = _ptr(self.const_score, C.c_int64)
        for name in ("node_affinity_raw", "taint_prefer_raw", "static_add"):
            setattr(s, name, _ptr(getattr(self, name), C.c_int64))
        s.n_terms = 0 if self.term_topo_key is None else len(self.term_topo_key)
        s.n_node_sets = 0 if self.node_sets is None else int(self.node_sets.shape[0])
        s.node_sets = _ptr(self.node_sets, C.c_uint64)
        for name in ("term_topo_key", "term_node_set", "match_off", "match_idx", "anti_off", "anti_idx", "port_off", "port_idx", "aff_off",
                     "aff_idx", "pref_off", "pref_idx", "pref_w", "own_off", "own_idx", "own_w", "spread_hard_off",
                     "spread_hard_idx", "spread_hard_skew", "spread_hard_self", "spread_hard_set", "spread_soft_off",
                     "spread_soft_idx", "spread_soft_skew"):
            setattr(s, name, _ptr(getattr(self, name) if s.n_terms else None, C.c_int32))
        s.local_spec_of = _ptr(self.local_spec_of, C.c_int32)
        s.n_local_specs = 0 if self.local_specs is None else len(self.local_specs)
        s.local_specs = C.cast(None, C.POINTER(LocalSpec)) if self.local_specs is None else \
            self.local_specs.ctypes.data_as(C.POINTER(LocalSpec))
        s.class_flags = _ptr(self.class_flags, C.c_uint8)
        s.topo_is_hostname = _ptr(self.topo_is_hostname, C.c_uint8)
        s.spread_log = _ptr(self.spread_log, C.c_double)
        return s


@dataclass
class BatchResult:
    unscheduled: np.ndarray
    used_cpu: np.ndarray
    used_mem: np.ndarray
    placement: Optional[np.ndarray]
    used_vg: Optional[np.ndarray] = None
    gpu_slices: Optional[np.ndarray] = None        # [S][P] uint64: byte d = gpu-mem slices booked on device d (gpu_ids_of)
    preempt_risk: Optional[np.ndarray] = None      # [S] uint8 (oracle runs; the library: Context.fetch_preempt_risk)

    def c_out(self) -> BatchOut:
        o = BatchOut()
        o.struct_size = C.sizeof(BatchOut)
        o.flags = 0
        o.unscheduled = _ptr(self.unscheduled, C.c_int32)
        o.used_cpu = _ptr(self.used_cpu, C.c_int64)
        o.used_mem = _ptr(self.used_mem, C.c_int64)
        o.placement = _ptr(self.placement, C.c_int32)
        o.used_vg = _ptr(self.used_vg, C.c_int64)
        o.gpu_slices = _ptr(self.gpu_slices, C.c_uint64)
        return o

    @staticmethod
    def alloc(S: int, P: int, want_placement: bool = True, want_gpu_slices: bool = False) -> "BatchResult":
        return BatchResult(np.zeros(S, np.int32), np.zeros(S, np.int64), np.zeros(S, np.int64),
                           np.full((S, P), -9, np.int32) if want_placement else None, np.zeros(S, np.int64),
                           np.zeros((S, P), np.uint64) if want_gpu_slices else None)


WANT_PLACEMENT, WANT_GPU_SLICES = 1, 2


def pack_gpu_index(ids: Sequence[int]) -> int:
    """simon_pods_soa.gpu_index: device ids of an alibabacloud.com/gpu-index annotation ("0-0-1" -> [0, 0, 1]) as nibbles, low first."""
    ids = list(ids)
    if not 0 < len(ids) <= 8 or any(not 0 <= i < MAX_GPU_DEV for i in ids):
        raise ValueError(f"gpu-index {ids}: 1..8 ids in 0..{MAX_GPU_DEV - 1} can be packed")
    return sum((i + 1) << (4 * k) for k, i in enumerate(ids))


def gpu_ids_of(slices: int) -> List[int]:
    """Device ids of one simon_batch_out.gpu_slices entry, in the order GpuNodeInfo.AllocateGpuId lists them (ascending, an id once
    per slice): the content of the alibabacloud.com/gpu-index annotation ("-".join)."""
    out = []
    for d in range(MAX_GPU_DEV):
        out += [d] * ((int(slices) >> (8 * d)) & 0xFF)
    return out


def scenarios_array(scen: Sequence) -> np.ndarray:
    a = np.ascontiguousarray(np.asarray(scen, dtype=np.int32).reshape(-1, 2))
    return a


def _pkg_dir() -> str:
    return os.path.dirname(os.path.abspath(__file__))


def library_path() -> str:
    """In-tree libsimon_hip.so; SIMON_HIP_LIB names another build of the SAME library (kernel A/B runs, profile builds)."""
    return os.environ.get("SIMON_HIP_LIB") or os.path.join(_pkg_dir(), "csrc", "libsimon_hip.so")


_LIB = None

EXPORTS = [
    "simon_hip_version", "simon_hip_device_count", "simon_ctx_create", "simon_ctx_destroy", "simon_last_error",
    "simon_load_nodes", "simon_load_pods", "simon_load_class_tables", "simon_load_scenarios", "simon_run_loaded",
    "simon_fetch_results", "simon_fetch_placement", "simon_fetch_gpu_slices", "simon_run_batch", "simon_min_plan", "simon_min_plan_vg", "simon_min_plan_device", "simon_explain", "simon_set_node_ranks",
    "simon_get_stats", "simon_device_results", "simon_explain_loaded", "simon_explain_local_detail",
    "simon_set_scalar_entries", "simon_set_pod_priorities", "simon_fetch_preempt_risk",
    "simon_group_set_scalar_entries", "simon_group_set_pod_priorities", "simon_group_fetch_preempt_risk",
    "simon_group_create", "simon_group_destroy", "simon_group_last_error", "simon_group_size", "simon_group_member",
    "simon_group_load_nodes", "simon_group_load_pods", "simon_group_load_class_tables", "simon_group_load_scenarios",
    "simon_group_run_loaded", "simon_group_fetch_results", "simon_group_run_batch", "simon_group_fetch_placement", "simon_group_fetch_gpu_slices",
    "simon_group_min_plan", "simon_group_collective",
]


def load_library(path: Optional[str] = None):
    """dlopen libsimon_hip.so and declare prototypes.  Raises if the library is missing."""
    global _LIB
    if _LIB is not None and path is None:
        return _LIB
    path = path or library_path()
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the product path.")
    lib = C.CDLL(path)
    vp = C.c_void_p
    lib.simon_hip_version.restype = C.c_int
    lib.simon_hip_device_count.restype = C.c_int
    lib.simon_ctx_create.restype = vp
    lib.simon_ctx_create.argtypes = [C.c_int]
    lib.simon_ctx_destroy.argtypes = [vp]
    lib.simon_ctx_destroy.restype = None
    lib.simon_last_error.restype = C.c_char_p
    lib.simon_last_error.argtypes = [vp]
    lib.simon_load_nodes.argtypes = [vp, C.POINTER(NodesSoA)]
    lib.simon_load_pods.argtypes = [vp, C.POINTER(PodsSoA)]
    lib.simon_load_class_tables.argtypes = [vp, C.POINTER(ClassTables)]
    lib.simon_load_scenarios.argtypes = [vp, C.POINTER(Scenario), C.c_int32, _p32, C.c_int32]
    lib.simon_run_loaded.argtypes = [vp, C.c_int32]
    lib.simon_fetch_results.argtypes = [vp, C.POINTER(BatchOut)]
    lib.simon_fetch_placement.argtypes = [vp, C.c_int32, _p32]
    lib.simon_fetch_gpu_slices.argtypes = [vp, C.c_int32, _pu64]
    lib.simon_run_batch.argtypes = [vp, C.POINTER(Scenario), C.c_int32, _p32, C.c_int32, C.POINTER(BatchOut)]
    lib.simon_set_node_ranks.argtypes = [vp, _p32]
    lib.simon_min_plan.argtypes = [vp, C.c_int32, C.c_int32, C.POINTER(Plan)]
    lib.simon_min_plan_vg.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, C.POINTER(Plan), C.POINTER(C.c_int32)]
    lib.simon_explain.argtypes = [vp, Scenario, _p32, _p32, _pu16, C.c_int32]
    lib.simon_explain_loaded.argtypes = [vp, C.c_int32, _p32, _pu16, C.c_int32]
    lib.simon_explain_local_detail.argtypes = [vp, _p64, C.c_int32]
    _pu8 = C.POINTER(C.c_uint8)
    for pre in ("simon_", "simon_group_"):
        getattr(lib, pre + "set_scalar_entries").argtypes = [vp, _pu8]
        getattr(lib, pre + "set_scalar_entries").restype = C.c_int
        getattr(lib, pre + "set_pod_priorities").argtypes = [vp, _p32, C.c_int32]
        getattr(lib, pre + "set_pod_priorities").restype = C.c_int
        getattr(lib, pre + "fetch_preempt_risk").argtypes = [vp, _pu8]
        getattr(lib, pre + "fetch_preempt_risk").restype = C.c_int
    lib.simon_group_create.restype = vp
    lib.simon_group_create.argtypes = [_p32, C.c_int32]
    lib.simon_group_destroy.argtypes = [vp]
    lib.simon_group_destroy.restype = None
    lib.simon_group_last_error.restype = C.c_char_p
    lib.simon_group_last_error.argtypes = [vp]
    lib.simon_group_size.restype = C.c_int32
    lib.simon_group_size.argtypes = [vp]
    lib.simon_group_member.restype = vp
    lib.simon_group_member.argtypes = [vp, C.c_int32]
    lib.simon_group_load_nodes.argtypes = [vp, C.POINTER(NodesSoA)]
    lib.simon_group_load_pods.argtypes = [vp, C.POINTER(PodsSoA)]
    lib.simon_group_load_class_tables.argtypes = [vp, C.POINTER(ClassTables)]
    lib.simon_group_load_scenarios.argtypes = [vp, C.POINTER(Scenario), C.c_int32, _p32, C.c_int32]
    lib.simon_group_run_loaded.argtypes = [vp, C.c_int32]
    lib.simon_group_fetch_results.argtypes = [vp, C.POINTER(BatchOut)]
    lib.simon_group_run_batch.argtypes = [vp, C.POINTER(Scenario), C.c_int32, _p32, C.c_int32, C.POINTER(BatchOut)]
    lib.simon_group_fetch_placement.argtypes = [vp, C.c_int32, _p32]
    lib.simon_group_fetch_gpu_slices.argtypes = [vp, C.c_int32, _pu64]
    lib.simon_group_min_plan.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, C.POINTER(Plan), C.POINTER(C.c_int32)]
    lib.simon_group_collective.argtypes = [vp]
    lib.simon_group_collective.restype = C.c_int32
    lib.simon_min_plan_device.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, C.POINTER(vp), C.POINTER(vp)]
    lib.simon_get_stats.argtypes = [vp, C.POINTER(Stats)]
    lib.simon_device_results.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    for name in EXPORTS:
        fn = getattr(lib, name)
        if fn.restype is C.c_int or name.startswith(("simon_load", "simon_run", "simon_fetch", "simon_min",
                                                        "simon_explain", "simon_get", "simon_device", "simon_group_load",
                                                        "simon_group_run", "simon_group_fetch", "simon_group_min")):
            fn.restype = C.c_int
    if lib.simon_hip_version() != ABI_VERSION:
        raise RuntimeError(f"ABI mismatch: library {lib.simon_hip_version()} != binding {ABI_VERSION}")
    if path == library_path():
        _LIB = lib
    return lib


class SimonError(RuntimeError):
    pass


class Context:
    """RAII wrapper over simon_ctx (one HIP device, one stream)."""

    def __init__(self, device_id: int = 0, lib=None):
        self.lib = lib or load_library()
        self.h = self.lib.simon_ctx_create(int(device_id))
        if not self.h:
            raise SimonError("simon_ctx_create failed (no gfx950 device visible?)")
        self.problem: Optional[Problem] = None
        self.S = 0
        self.scen: Optional[np.ndarray] = None

    def close(self):
        if getattr(self, "h", None):
            self.lib.simon_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc: int, what: str):
        if rc < 0:
            msg = self.lib.simon_last_error(self.h)
            raise SimonError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")
        return rc

    def load_problem(self, prob: Problem):
        prob.normalise()
        n, p, t = prob.c_nodes(), prob.c_pods(), prob.c_tables()
        self._check(self.lib.simon_load_nodes(self.h, C.byref(n)), "simon_load_nodes")
        self._check(self.lib.simon_load_pods(self.h, C.byref(p)), "simon_load_pods")
        if prob.scalar_entries is not None:
            self._check(self.lib.simon_set_scalar_entries(self.h, _ptr(prob.scalar_entries, C.c_uint8)), "simon_set_scalar_entries")
        if prob.priority is not None:
            self._check(self.lib.simon_set_pod_priorities(self.h, _ptr(prob.priority, C.c_int32), int(prob.init_min_priority)), "simon_set_pod_priorities")
        self._check(self.lib.simon_load_class_tables(self.h, C.byref(t)), "simon_load_class_tables")
        self.problem = prob

    def load_scenarios(self, scen, orders: np.ndarray):
        scen = scenarios_array(scen)
        orders = np.ascontiguousarray(orders, dtype=np.int32).reshape(-1, self.problem.n_pods)
        self._check(self.lib.simon_load_scenarios(self.h, scen.ctypes.data_as(C.POINTER(Scenario)), len(scen),
                                                  _ptr(orders, C.c_int32), orders.shape[0]), "simon_load_scenarios")
        self.S = len(scen)
        self.scen = scen

    def run_loaded(self, want_placement: bool = True, want_gpu_slices: bool = False):
        flags = (WANT_PLACEMENT if want_placement else 0) | (WANT_GPU_SLICES if want_gpu_slices else 0)
        self._check(self.lib.simon_run_loaded(self.h, flags), "simon_run_loaded")

    def fetch(self, want_placement: bool = True, want_gpu_slices: bool = False) -> BatchResult:
        res = BatchResult.alloc(self.S, self.problem.n_pods, want_placement, want_gpu_slices)
        out = res.c_out()
        self._check(self.lib.simon_fetch_results(self.h, C.byref(out)), "simon_fetch_results")
        return res

    def fetch_gpu_slices(self, scenario: int) -> np.ndarray:
        row = np.zeros(self.problem.n_pods, np.uint64)
        self._check(self.lib.simon_fetch_gpu_slices(self.h, int(scenario), _ptr(row, C.c_uint64)), "simon_fetch_gpu_slices")
        return row

    def fetch_preempt_risk(self) -> np.ndarray:
        """[S] uint8: 1 = some pod failed while a pod of lower priority was placed -- DefaultPreemption could have evicted there, the
        scenario needs the reference's own path; 0 = exact (include/simon_hip.h: simon_fetch_preempt_risk)."""
        risk = np.zeros(self.S, np.uint8)
        self._check(self.lib.simon_fetch_preempt_risk(self.h, _ptr(risk, C.c_uint8)), "simon_fetch_preempt_risk")
        return risk

    def fetch_placement(self, scenario: int) -> np.ndarray:
        row = np.zeros(self.problem.n_pods, np.int32)
        self._check(self.lib.simon_fetch_placement(self.h, int(scenario), _ptr(row, C.c_int32)),
                    "simon_fetch_placement")
        return row

    def run_batch(self, scen, orders: np.ndarray, want_placement: bool = True, want_gpu_slices: bool = False) -> BatchResult:
        scen = scenarios_array(scen)
        orders = np.ascontiguousarray(orders, dtype=np.int32).reshape(-1, self.problem.n_pods)
        res = BatchResult.alloc(len(scen), self.problem.n_pods, want_placement, want_gpu_slices)
        out = res.c_out()
        self._check(self.lib.simon_run_batch(self.h, scen.ctypes.data_as(C.POINTER(Scenario)), len(scen),
                                             _ptr(orders, C.c_int32), orders.shape[0], C.byref(out)),
                    "simon_run_batch")
        self.S = len(scen)
        self.scen = scen
        return res

    def set_node_ranks(self, ranks) -> None:
        """Per-scenario nodeTree order: ranks[s][j] = position of pool node j in scenario s's canonical order (None: pool order)."""
        if ranks is None:
            self._check(self.lib.simon_set_node_ranks(self.h, None), "simon_set_node_ranks")
            return
        r = np.ascontiguousarray(ranks, dtype=np.int32)
        self._check(self.lib.simon_set_node_ranks(self.h, _ptr(r, C.c_int32)), "simon_set_node_ranks")

    def min_plan(self, max_cpu_pct: int = 100, max_mem_pct: int = 100) -> Plan:
        plan = Plan()
        self._check(self.lib.simon_min_plan(self.h, int(max_cpu_pct), int(max_mem_pct), C.byref(plan)),
                    "simon_min_plan")
        return plan

    def min_plan_vg(self, max_cpu_pct: int = 100, max_mem_pct: int = 100, max_vg_pct: int = 100):
        """simon_min_plan with the MaxVG cap; returns (Plan, vg_pct of the winning scenario)."""
        plan, vg = Plan(), C.c_int32(0)
        self._check(self.lib.simon_min_plan_vg(self.h, int(max_cpu_pct), int(max_mem_pct), int(max_vg_pct), C.byref(plan),
                                               C.byref(vg)), "simon_min_plan_vg")
        return plan, int(vg.value)

    def explain(self, n_nodes: int, order: np.ndarray, max_failed: int = 64):
        order = np.ascontiguousarray(order, dtype=np.int32)
        failed = np.full(max_failed, -1, np.int32)
        codes = np.zeros((max_failed, n_nodes), np.uint16)
        sc = Scenario(int(n_nodes), 0)
        n = self._check(self.lib.simon_explain(self.h, sc, _ptr(order, C.c_int32), _ptr(failed, C.c_int32),
                                               _ptr(codes, C.c_uint16), int(max_failed)), "simon_explain")
        k = min(n, max_failed)
        return n, failed[:k], codes[:k]

    def explain_loaded(self, scenario: int, max_failed: int = 64):
        """simon_explain_loaded: replay scenario `scenario` of the loaded batch (its order, its own node ranks)."""
        if not (self.scen is not None and 0 <= int(scenario) < len(self.scen)):
            raise SimonError(f"simon_explain_loaded: scenario {scenario} is not one of the {0 if self.scen is None else len(self.scen)} loaded")
        n_nodes = int(self.scen[scenario, 0])
        failed = np.full(max_failed, -1, np.int32)
        codes = np.zeros((max_failed, n_nodes), np.uint16)
        n = self._check(self.lib.simon_explain_loaded(self.h, int(scenario), _ptr(failed, C.c_int32), _ptr(codes, C.c_uint16),
                                                      int(max_failed)), "simon_explain_loaded")
        k = min(n, max_failed)
        return n, failed[:k], codes[:k]

    def explain_local_detail(self, n_failed: int, n_nodes: int) -> Optional[np.ndarray]:
        """simon_explain_local_detail after explain / explain_loaded: [n_failed][n_nodes][4] int64 {LOCAL_ERR_*, a, b, c} -- what
        Open-Local's error texts carry -- or None when the problem has no local storage."""
        if n_failed <= 0:
            return None
        detail = np.zeros((int(n_failed), int(n_nodes), 4), np.int64)
        k = self._check(self.lib.simon_explain_local_detail(self.h, _ptr(detail, C.c_int64), int(n_failed)), "simon_explain_local_detail")
        return detail[:k] if k > 0 else None

    def stats(self) -> Stats:
        st = Stats()
        self._check(self.lib.simon_get_stats(self.h, C.byref(st)), "simon_get_stats")
        return st

    def device_results(self):
        a, b, c = C.c_void_p(), C.c_void_p(), C.c_void_p()
        self._check(self.lib.simon_device_results(self.h, C.byref(a), C.byref(b), C.byref(c)),
                    "simon_device_results")
        return a.value, b.value, c.value


class Group:
    """RAII wrapper over simon_group: several devices of one node behind one handle (scenario s runs on member
    s % n_dev; all scenario indices are the caller's)."""

    def __init__(self, device_ids: Sequence[int], lib=None):
        self.lib = lib or load_library()
        ids = np.ascontiguousarray(device_ids, np.int32)
        self.h = self.lib.simon_group_create(_ptr(ids, C.c_int32), len(ids))
        if not self.h:
            raise SimonError(f"simon_group_create({list(device_ids)}) failed (no such gfx950 device?)")
        self.problem: Optional[Problem] = None
        self.S = 0

    def close(self):
        if getattr(self, "h", None):
            self.lib.simon_group_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc: int, what: str):
        if rc < 0:
            msg = self.lib.simon_group_last_error(self.h)
            raise SimonError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")
        return rc

    @property
    def size(self) -> int:
        return int(self.lib.simon_group_size(self.h))

    def member_stats(self, i: int) -> Stats:
        st = Stats()
        rc = self.lib.simon_get_stats(self.lib.simon_group_member(self.h, int(i)), C.byref(st))
        if rc < 0:
            raise SimonError(f"simon_get_stats(member {i}) failed ({rc})")
        return st

    def load_problem(self, prob: Problem):
        prob.normalise()
        n, p, t = prob.c_nodes(), prob.c_pods(), prob.c_tables()
        self._check(self.lib.simon_group_load_nodes(self.h, C.byref(n)), "simon_group_load_nodes")
        self._check(self.lib.simon_group_load_pods(self.h, C.byref(p)), "simon_group_load_pods")
        if prob.scalar_entries is not None:
            self._check(self.lib.simon_group_set_scalar_entries(self.h, _ptr(prob.scalar_entries, C.c_uint8)), "simon_group_set_scalar_entries")
        if prob.priority is not None:
            self._check(self.lib.simon_group_set_pod_priorities(self.h, _ptr(prob.priority, C.c_int32), int(prob.init_min_priority)), "simon_group_set_pod_priorities")
        self._check(self.lib.simon_group_load_class_tables(self.h, C.byref(t)), "simon_group_load_class_tables")
        self.problem = prob

    def load_scenarios(self, scen, orders: np.ndarray):
        scen = scenarios_array(scen)
        orders = np.ascontiguousarray(orders, dtype=np.int32).reshape(-1, self.problem.n_pods)
        self._check(self.lib.simon_group_load_scenarios(self.h, scen.ctypes.data_as(C.POINTER(Scenario)), len(scen),
                                                        _ptr(orders, C.c_int32), orders.shape[0]), "simon_group_load_scenarios")
        self.S = len(scen)

    def run_loaded(self, want_placement: bool = True, want_gpu_slices: bool = False):
        flags = (WANT_PLACEMENT if want_placement else 0) | (WANT_GPU_SLICES if want_gpu_slices else 0)
        self._check(self.lib.simon_group_run_loaded(self.h, flags), "simon_group_run_loaded")

    def fetch(self, want_placement: bool = True, want_gpu_slices: bool = False) -> BatchResult:
        res = BatchResult.alloc(self.S, self.problem.n_pods, want_placement, want_gpu_slices)
        out = res.c_out()
        self._check(self.lib.simon_group_fetch_results(self.h, C.byref(out)), "simon_group_fetch_results")
        return res

    def run_batch(self, scen, orders: np.ndarray, want_placement: bool = True, want_gpu_slices: bool = False) -> BatchResult:
        self.load_scenarios(scen, orders)
        self.run_loaded(want_placement, want_gpu_slices)
        return self.fetch(want_placement, want_gpu_slices)

    def fetch_placement(self, scenario: int) -> np.ndarray:
        row = np.zeros(self.problem.n_pods, np.int32)
        self._check(self.lib.simon_group_fetch_placement(self.h, int(scenario), _ptr(row, C.c_int32)), "simon_group_fetch_placement")
        return row

    def fetch_gpu_slices(self, scenario: int) -> np.ndarray:
        row = np.zeros(self.problem.n_pods, np.uint64)
        self._check(self.lib.simon_group_fetch_gpu_slices(self.h, int(scenario), _ptr(row, C.c_uint64)), "simon_group_fetch_gpu_slices")
        return row

    def fetch_preempt_risk(self) -> np.ndarray:
        risk = np.zeros(self.S, np.uint8)
        self._check(self.lib.simon_group_fetch_preempt_risk(self.h, _ptr(risk, C.c_uint8)), "simon_group_fetch_preempt_risk")
        return risk

    def min_plan(self, max_cpu_pct: int = 100, max_mem_pct: int = 100, max_vg_pct: int = 100):
        plan, vg = Plan(), C.c_int32(0)
        self._check(self.lib.simon_group_min_plan(self.h, int(max_cpu_pct), int(max_mem_pct), int(max_vg_pct), C.byref(plan),
                                                  C.byref(vg)), "simon_group_min_plan")
        return plan, int(vg.value)

    def collective(self) -> str:
        """How the last min_plan combined the members' plans: "rccl_all_gather" (members on distinct devices: one ncclAllGather of the
        8-byte plan keys, device to device) or "host" (one member / shared devices / librccl unavailable)."""
        return {0: "host", 1: "rccl_all_gather"}.get(int(self.lib.simon_group_collective(self.h)), "?")
