package hipengine

/*
#include "simon_hip.h"
*/
import "C"

import "unsafe"

// Compile-time checks: the Go constants of flatten_terms.go that restate header constants ARE the header's values (an array type of
// negative or non-zero length does not convert to [0]struct{}: a mismatch stops the build).  tests/test_go_static.py checks the same
// from the comments while no Go toolchain is at hand -- the round-4 advisor found spreadDupKey = 1 << 14 against 0x40000000 that way.
var (
	_ [0]struct{} = [spreadDupKey - C.SIMON_SPREAD_DUP_KEY]struct{}{}
	_ [0]struct{} = [maxSpread - C.SIMON_MAX_SPREAD]struct{}{}
	_ [0]struct{} = [classAffSelf - C.SIMON_CLASS_AFF_SELF]struct{}{}
	_ [0]struct{} = [maxVG - C.SIMON_MAX_VG]struct{}{}
	_ [0]struct{} = [maxLDev - C.SIMON_MAX_LDEV]struct{}{}
	_ [0]struct{} = [maxLVol - C.SIMON_MAX_LVOL]struct{}{}
)

// Flat is the SoA image of (cluster nodes + new-node clones, ordered pod list, per-class tables): the three input
// structs of include/simon_hip.h with Go slices in place of the C pointers.  A nil slice = the optional array is
// absent ("all zero" / feature off), exactly as a NULL pointer in the header.  The reference implementation that fills
// every field from Kubernetes objects is open-simulator_amd/flatten.py; flatten.go of this package fills the subset
// Supports() accepts.
type Flat struct {
	// ---- simon_nodes_soa ----
	AllocCPU, AllocMem, AllocEph                      []int64 // [N]
	AllocPods                                         []int32 // [N]
	InitReqCPU, InitReqMem, InitReqEph                []int64 // [N] pods bound before the stream
	InitNzCPU, InitNzMem                              []int64
	InitNPods                                         []int32
	NodeClass                                         []int32 // [N] column of the [Cp][Cn] tables
	NScalar                                           int
	ScalarAlloc, InitScalarReq                        []int64 // [K][N]
	GpuCnt                                            []int32 // [N]; nil = Open-Gpu-Share off
	GpuMemTotal                                       []int64 // [N]
	InitGpuUsed                                       []int64 // [N][SIMON_MAX_GPU_DEV]
	LocalFlags, LocalVGCnt, LocalVGName, LocalDevCnt  []int32 // Open-Local node rows; nil = feature off
	LocalDevMedia, InitDevAlloc                       []int32
	LocalVGCap, InitVGReq, LocalDevCap                []int64
	NTopoKeys                                         int
	TopoDom, TopoNDom                                 []int32 // [Kt][N], [Kt]
	// ---- simon_pods_soa ----
	ReqCPU, ReqMem, ReqEph, NzCPU, NzMem              []int64 // [P]
	ScalarReq                                         []int64 // [K][P]
	PodClass, Preset, Gate, Pin                       []int32 // [P]
	PodGpuMem                                         []int64
	PodGpuCnt                                         []int32
	PodGpuIndex                                       []uint32 // [P] gpu-index annotation the pod arrives with, packed (PackGpuIndex); nil = none
	ScalarEntries                                     []uint8  // [P] ABI v6 (simon_set_scalar_entries): bit k = the request holds an ENTRY for ScalarNames[k], bit 7 = for a resource nobody requests a quantity of; nil = none beyond the quantities
	Priority                                          []int32  // [P] ABI v6 (simon_set_pod_priorities): spec.priority; nil = all equal
	// ---- simon_class_tables ----
	Cp, Cn                                            int
	StaticMask                                        []uint64 // [Cp][ceil(N/64)]
	StaticReason                                      []uint8  // [Cp][N]
	SimonRaw, ConstScore                              []int64  // [Cp][Cn], [Cp]
	NodeAffinityRaw, TaintPreferRaw, StaticAdd        []int64  // [Cp][Cn]; nil = constant plugin
	NTerms, NNodeSets                                 int
	TermTopoKey, TermNodeSet                          []int32
	NodeSets                                          []uint64
	MatchOff, MatchIdx, AntiOff, AntiIdx              []int32
	PortOff, PortIdx, AffOff, AffIdx                  []int32
	ClassFlags                                        []uint8
	PrefOff, PrefIdx, PrefW, OwnOff, OwnIdx, OwnW     []int32
	SpreadHardOff, SpreadHardIdx, SpreadHardSkew      []int32
	SpreadHardSelf, SpreadHardSet                     []int32
	SpreadSoftOff, SpreadSoftIdx, SpreadSoftSkew      []int32
	LocalSpecOf                                       []int32
	LocalSpecs                                        []LocalSpec
	TopoIsHostname                                    []uint8
	SpreadLog                                         []float64 // [N+1]: math.Log(float64(i+2)), Go's own math.Log
	// ---- bookkeeping for the way back (not part of the ABI) ----
	NodeNames     []string
	StaticReasons []string // reason id -> FitError text
	ScalarNames   []string
	VGNames       []string // interned Open-Local volume-group names (LocalVGName / LocalSpec.LvmVG ids -> text)
}

// LocalSpec mirrors simon_local_spec (same field order and sizes: it is copied byte for byte).
type LocalSpec struct {
	NLvm, NSsd, NHdd, Pad int32
	LvmSize               [C.SIMON_MAX_LVOL]int64
	LvmVG                 [C.SIMON_MAX_LVOL]int32
	SsdSize               [C.SIMON_MAX_LVOL]int64
	HddSize               [C.SIMON_MAX_LVOL]int64
}

func (f *Flat) N() int { return len(f.AllocCPU) }
func (f *Flat) P() int { return len(f.ReqCPU) }

func (f *Flat) cNodes(a *cArena) C.simon_nodes_soa {
	n := C.simon_nodes_soa{struct_size: C.uint32_t(unsafe.Sizeof(C.simon_nodes_soa{})), n_nodes: C.int32_t(f.N())}
	n.alloc_cpu, n.alloc_mem, n.alloc_eph, n.alloc_pods = a.i64(f.AllocCPU), a.i64(f.AllocMem), a.i64(f.AllocEph), a.i32(f.AllocPods)
	n.init_req_cpu, n.init_req_mem, n.init_req_eph = a.i64(f.InitReqCPU), a.i64(f.InitReqMem), a.i64(f.InitReqEph)
	n.init_nz_cpu, n.init_nz_mem, n.init_npods = a.i64(f.InitNzCPU), a.i64(f.InitNzMem), a.i32(f.InitNPods)
	n.node_class = a.i32(f.NodeClass)
	n.n_scalar = C.int32_t(f.NScalar)
	n.scalar_alloc, n.init_scalar_req = a.i64(f.ScalarAlloc), a.i64(f.InitScalarReq)
	n.gpu_cnt, n.gpu_mem_total, n.init_gpu_used = a.i32(f.GpuCnt), a.i64(f.GpuMemTotal), a.i64(f.InitGpuUsed)
	n.local_flags, n.local_vg_cnt, n.local_vg_cap = a.i32(f.LocalFlags), a.i32(f.LocalVGCnt), a.i64(f.LocalVGCap)
	n.init_vg_req, n.local_vg_name, n.local_dev_cnt = a.i64(f.InitVGReq), a.i32(f.LocalVGName), a.i32(f.LocalDevCnt)
	n.local_dev_cap, n.local_dev_media, n.init_dev_alloc = a.i64(f.LocalDevCap), a.i32(f.LocalDevMedia), a.i32(f.InitDevAlloc)
	n.n_topo_keys = C.int32_t(f.NTopoKeys)
	n.topo_dom, n.topo_n_dom = a.i32(f.TopoDom), a.i32(f.TopoNDom)
	return n
}

func (f *Flat) cPods(a *cArena) C.simon_pods_soa {
	p := C.simon_pods_soa{struct_size: C.uint32_t(unsafe.Sizeof(C.simon_pods_soa{})), n_pods: C.int32_t(f.P())}
	p.req_cpu, p.req_mem, p.req_eph = a.i64(f.ReqCPU), a.i64(f.ReqMem), a.i64(f.ReqEph)
	p.nz_cpu, p.nz_mem, p.scalar_req = a.i64(f.NzCPU), a.i64(f.NzMem), a.i64(f.ScalarReq)
	p.pod_class, p.preset_node, p.gate_node, p.pin_node = a.i32(f.PodClass), a.i32(f.Preset), a.i32(f.Gate), a.i32(f.Pin)
	p.gpu_mem, p.gpu_cnt = a.i64(f.PodGpuMem), a.i32(f.PodGpuCnt)
	p.gpu_index = a.u32(f.PodGpuIndex)
	return p
}

func (f *Flat) cTables(a *cArena) C.simon_class_tables {
	t := C.simon_class_tables{struct_size: C.uint32_t(unsafe.Sizeof(C.simon_class_tables{})),
		n_pod_classes: C.int32_t(f.Cp), n_node_classes: C.int32_t(f.Cn)}
	t.static_mask, t.static_reason = a.u64(f.StaticMask), a.u8(f.StaticReason)
	t.simon_raw, t.const_score = a.i64(f.SimonRaw), a.i64(f.ConstScore)
	t.node_affinity_raw, t.taint_prefer_raw, t.static_add = a.i64(f.NodeAffinityRaw), a.i64(f.TaintPreferRaw), a.i64(f.StaticAdd)
	if f.NTerms > 0 {
		f.fillTermTables(&t, a)
	}
	if f.LocalSpecOf != nil {
		t.local_spec_of = a.i32(f.LocalSpecOf)
		t.n_local_specs = C.int32_t(len(f.LocalSpecs))
		if len(f.LocalSpecs) > 0 { // LocalSpec has the layout of simon_local_spec and holds no pointers
			bytes := len(f.LocalSpecs) * int(unsafe.Sizeof(C.simon_local_spec{}))
			p := a.alloc(bytes)
			copy((*[1 << 30]byte)(p)[:bytes:bytes], (*[1 << 30]byte)(unsafe.Pointer(&f.LocalSpecs[0]))[:bytes:bytes])
			t.local_specs = (*C.simon_local_spec)(p)
		}
	}
	return t
}

// fillTermTables attaches the topology-term tables (InterPodAffinity, PodTopologySpread, NodePorts role lists).
func (f *Flat) fillTermTables(t *C.simon_class_tables, a *cArena) {
	t.n_terms = C.int32_t(f.NTerms)
	t.term_topo_key, t.term_node_set = a.i32(f.TermTopoKey), a.i32(f.TermNodeSet)
	t.n_node_sets = C.int32_t(f.NNodeSets)
	t.node_sets = a.u64(f.NodeSets)
	t.match_off, t.match_idx = a.i32(f.MatchOff), a.i32(f.MatchIdx)
	t.anti_off, t.anti_idx = a.i32(f.AntiOff), a.i32(f.AntiIdx)
	t.port_off, t.port_idx = a.i32(f.PortOff), a.i32(f.PortIdx)
	t.aff_off, t.aff_idx = a.i32(f.AffOff), a.i32(f.AffIdx)
	t.class_flags = a.u8(f.ClassFlags)
	t.pref_off, t.pref_idx, t.pref_w = a.i32(f.PrefOff), a.i32(f.PrefIdx), a.i32(f.PrefW)
	t.own_off, t.own_idx, t.own_w = a.i32(f.OwnOff), a.i32(f.OwnIdx), a.i32(f.OwnW)
	t.spread_hard_off, t.spread_hard_idx = a.i32(f.SpreadHardOff), a.i32(f.SpreadHardIdx)
	t.spread_hard_skew, t.spread_hard_self, t.spread_hard_set = a.i32(f.SpreadHardSkew), a.i32(f.SpreadHardSelf), a.i32(f.SpreadHardSet)
	t.spread_soft_off, t.spread_soft_idx, t.spread_soft_skew = a.i32(f.SpreadSoftOff), a.i32(f.SpreadSoftIdx), a.i32(f.SpreadSoftSkew)
	t.topo_is_hostname = a.u8(f.TopoIsHostname)
	t.spread_log = a.f64(f.SpreadLog)
}

// PackGpuIndex packs the device ids of an alibabacloud.com/gpu-index annotation the way simon_pods_soa.gpu_index wants them: nibble i
// = 1 + i-th id, 0 ends the list.  ok = false when the list does not fit (more than 8 ids, an id beyond SIMON_MAX_GPU_DEV - 1): the
// caller keeps the Go path.
func PackGpuIndex(ids []int) (packed uint32, ok bool) {
	if len(ids) == 0 || len(ids) > 8 {
		return 0, false
	}
	for i, id := range ids {
		if id < 0 || id >= C.SIMON_MAX_GPU_DEV {
			return 0, false
		}
		packed |= uint32(id+1) << (4 * uint(i))
	}
	return packed, true
}
