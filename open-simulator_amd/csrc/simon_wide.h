// simon_wide.h -- host inputs shared by the staging code, and the WIDE (all-feature, int64,
// HBM/L2-streamed state) kernel interface.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/simon_hip.h"

namespace simon {

// Host-side copies of everything simon_load_* received (the caller's buffers are not retained).
struct HostInputs {
    int N = 0, P = 0, K = 0, Kt = 0, Cp = 1, Cn = 1, Tm = 0, R = 0;
    std::vector<int64_t> alloc_cpu, alloc_mem, alloc_eph, i_req_cpu, i_req_mem, i_req_eph, i_nz_cpu, i_nz_mem;
    std::vector<int32_t> alloc_pods, i_npods, node_class;
    std::vector<int64_t> scalar_alloc, i_scalar_req, gpu_mem_total, i_gpu_used;
    std::vector<int32_t> gpu_cnt, topo_dom, topo_n_dom;
    bool has_gpu = false;            // derived per staging: has_gpu_nodes, or some pod requests GPU memory
    bool has_gpu_index = false;      // some pod arrives with a gpu-index annotation (simon_pods_soa.gpu_index)
    bool has_gpu_nodes = false;      // the loaded pool carries GPU arrays (simon_load_nodes)
    std::vector<int64_t> p_req_cpu, p_req_mem, p_req_eph, p_nz_cpu, p_nz_mem, p_scalar, p_gpu_mem;
    std::vector<int32_t> p_cls, p_preset, p_gate, p_gpu_cnt, p_pin;
    std::vector<uint32_t> p_gpu_index;   // packed preset device ids (simon_pods_soa.gpu_index), 0 = none
    std::vector<uint8_t> p_entries;      // ABI v6 (simon_set_scalar_entries): bit k = the request holds an ENTRY for extended resource k (bit 7: one no node tracks); empty = where scalar_req != 0
    std::vector<int32_t> p_priority;     // ABI v6 (simon_set_pod_priorities): spec.priority; empty = all equal
    int32_t init_min_priority = 0x7fffffff;
    bool has_pin = false;         // some pod is pinned to one node (simon_pods_soa.pin_node)
    std::vector<uint64_t> static_mask;
    std::vector<uint8_t> static_reason;
    std::vector<int64_t> simon_raw, const_score;
    bool has_mask = false;
    // ABI v2: static (pod class, node class) score inputs
    std::vector<int64_t> na_raw, tt_raw, static_add;
    bool has_na = false, has_tt = false, has_add = false;
    // topology terms (InterPodAffinity + PodTopologySpread)
    std::vector<int32_t> term_key, term_set;
    std::vector<uint64_t> node_sets;
    std::vector<int32_t> match_off, match_idx, anti_off, anti_idx, aff_off, aff_idx, port_off, port_idx;
    std::vector<uint8_t> class_flags;
    std::vector<int32_t> pref_off, pref_idx, pref_w, own_off, own_idx, own_w;
    std::vector<int32_t> sh_off, sh_idx, sh_skew, sh_self, sh_set, ss_off, ss_idx, ss_skew;
    std::vector<uint8_t> topo_is_hostname;
    std::vector<double> spread_log;
    // Open-Local
    bool has_local = false;
    std::vector<int32_t> l_flags, l_vg_cnt, l_vg_name, l_dev_cnt, l_dev_media, l_dev_alloc, l_spec_of;
    std::vector<int64_t> l_vg_cap, l_vg_req, l_dev_cap;
    std::vector<simon_local_spec> l_specs;
    bool has_ipa_score = false;   // any pref_* or own_* entry exists
    // everything of v2_features() but NodePorts (port terms are node-level by construction: the score-table kernel's REST path takes them)
    bool v2_features_but_ports() const {
        return has_na || has_tt || has_add || !aff_idx.empty() || has_ipa_score || !sh_idx.empty() || !ss_idx.empty() || has_local;
    }
    // ... and but the static score tables (NodeAffinity preferred, TaintToleration PreferNoSchedule, weighted additions), which the
    // score-table kernel folds into its class term
    bool v2_features_but_ports_and_static() const {      // ... and but required affinity (REST path: rows that must be SET)
        return has_ipa_score || !sh_idx.empty() || !ss_idx.empty() || has_local;
    }
    bool v2_features() const {
        return has_na || has_tt || has_add || !aff_idx.empty() || has_ipa_score || !sh_idx.empty() || !ss_idx.empty() || !port_idx.empty() || has_local;
    }
};

// Static half of a node's resource row, one 32-byte load: allocatable cpu / memory and their correctly rounded
// reciprocals RN(1 / alloc) (host IEEE division), from which the kernel forms the exact quotients req / alloc of
// BalancedAllocation with two Markstein corrections (simon_device.h: div_by_rcp) instead of an IEEE division sequence.
struct WideNodeStatic {
    int64_t alloc_cpu, alloc_mem;
    double rcp_cpu, rcp_mem;
};
static_assert(sizeof(WideNodeStatic) == 32, "WideNodeStatic must be 32 bytes");

// One pod of the stream, WIDE layout (128 B).
struct WidePod {
    int64_t req_cpu, req_mem, req_eph, nz_cpu, nz_mem, gpu_mem;
    int64_t scalar[SIMON_MAX_SCALAR];
    int32_t cls, preset, gate, gpu_cnt;
    uint32_t flags;  // kPod* bits
    int32_t sig;     // request signature: row of the (signature, node) table
    int32_t pin;     // >= 0: the only node the pod's node affinity admits (DaemonSet pods); -1: none
    uint32_t gpu_index;   // packed device ids of the gpu-index annotation the pod arrives with (simon_pods_soa.gpu_index), 0 = none
    uint32_t pad[4];
};
static_assert(sizeof(WidePod) == 128, "WidePod must be 128 bytes");

// One distinct pod request signature (what NodeResourcesFit + LeastAllocated + BalancedAllocation depend on), 96 B.
struct WideSig {
    int64_t req_cpu, req_mem, req_eph, nz_cpu, nz_mem;
    int64_t scalar[SIMON_MAX_SCALAR];
    uint32_t flags;  // kPodZero
    uint32_t pad[5];
};
static_assert(sizeof(WideSig) == 96, "WideSig must be 96 bytes");
constexpr int kMaxWideSigs = 1024;   // more distinct signatures than this: the kernel evaluates every node every cycle
constexpr uint32_t kPodZero = 1u;      // all-zero request incl. scalars (fit.go:244-249): no cpu / memory / ephemeral storage and NO ScalarResources entry
constexpr uint32_t kPodEntry0 = 1u << 24;   // << k: the request holds an entry for extended resource k although its quantity is 0 (compared all the same, fit.go:275-299)
constexpr uint32_t kPodTerms = 2u;     // class touches topology counters at assume (match / anti / aff / own lists non-empty)
constexpr uint32_t kPodFilt = 128u;    // class has an InterPodAffinity FILTER to evaluate per node (anti, owned-anti match or required affinity)
constexpr uint32_t kPodHard = 4u;      // class has DoNotSchedule spread constraints
constexpr uint32_t kPodSoft = 8u;      // class has ScheduleAnyway spread constraints
constexpr uint32_t kPodIpa = 16u;      // InterPodAffinity.Score can be non-zero for this class
constexpr uint32_t kPodPorts = 32u;    // class binds host ports (NodePorts filter)
constexpr uint32_t kPodLocal = 64u;    // class requests Open-Local volumes

struct WideScenario {
    int32_t n_nodes, order_id;
};

// Kernel arguments are split in two: WideArgs (by value, lives in SGPRs: what every cycle touches) and WideCold (a
// struct in device memory, scalar-loaded on demand by the rarely taken feature paths).  Passing everything by value
// made the compiler spill ~100 argument SGPRs into VGPR lanes and pay a v_readlane per use.
struct WideCold {
    int32_t Kt, Cp, Tm, total_dom, seen_stride, max_failed;
    const int64_t* alloc_eph; const int64_t* scalar_alloc /*[K][N]*/;
    const int32_t* gpu_cnt; const int64_t* gpu_mem_total;
    const int32_t* topo_dom /*[Kt][N]*/;
    // initial state [N]
    const int64_t* i_req_cpu; const int64_t* i_req_mem; const int64_t* i_req_eph; const int64_t* i_nz_cpu;
    const int64_t* i_nz_mem; const int32_t* i_npods; const int64_t* i_scalar_req; const int64_t* i_gpu_used;
    // tables
    const uint8_t* static_reason;
    const int64_t* na_raw; const int64_t* tt_raw; const int64_t* static_add;   // [Cp][Cn] or null
    const int32_t* term_key; const int32_t* term_dom_off /*[Tm] offset of term t's counters*/;
    const int32_t* term_set /*[Tm] row of node_sets or -1*/; const uint64_t* node_sets;
    const int32_t* anti_off; const int32_t* anti_idx; const int32_t* match_off; const int32_t* match_idx;
    // match lists restricted to the terms some class OWNS as required anti-affinity (only those have cnt_owner != 0) resp.
    // as a scoring term (only those have w_owner != 0): what Filter's "existing pods" check and Score's symmetry walk read
    const int32_t* manti_off; const int32_t* manti_idx; const int32_t* mown_off; const int32_t* mown_idx;
    // The hot lists once more with the term's metadata INLINE (one scalar load per entry instead of idx -> term_dom_off / term_key, a
    // dependent chain per term and batch): low word = term_dom_off[t], bits 32..47 = topology key, bit 48 = the key's domain of node j
    // IS j (every node carries the label, domains numbered in node order: hostname-like keys, the node-identity key of NodePorts) --
    // no domain row to load; bits 49..50 = topo_is_hostname[key].  ss_ent is parallel to ss_idx; filt_ent[filt_off[c] ..) = the counters
    // that must be 0 for a pod of class c (its anti list against cnt_match, then the matched-and-required terms against cnt_owner:
    // the offset is absolute inside the scenario's counter block); ipa_ent / ipa_w likewise = the InterPodAffinity raw score's
    // summands (pref terms x cnt_match with their weights, matched-and-owned terms x w_owner with weight 1).
    const uint64_t* ss_ent; const int32_t* filt_off; const uint64_t* filt_ent; const int32_t* ipa_off; const uint64_t* ipa_ent; const int32_t* ipa_w;
    const int32_t* aff_off; const int32_t* aff_idx; const uint8_t* class_flags;
    const int32_t* port_off; const int32_t* port_idx;
    const int32_t* pref_off; const int32_t* pref_idx; const int32_t* pref_w;
    const int32_t* own_off; const int32_t* own_idx; const int32_t* own_w;
    const int32_t* sh_off; const int32_t* sh_idx; const int32_t* sh_skew; const int32_t* sh_self; const int32_t* sh_set;
    const int32_t* sh_first_reg /*[E][N]: lowest eligible node index sharing node j's domain, INT_MAX if none*/;
    const int32_t* ss_off; const int32_t* ss_idx; const int32_t* ss_skew;
    const uint8_t* topo_is_hostname; const double* spread_log; const int32_t* key_seen_off /*[Kt]*/;
    // Open-Local: node storage (static) and per-class volume specs
    const int32_t* l_flags; const int32_t* l_vg_cnt; const int64_t* l_vg_cap; const int32_t* l_vg_name; const int64_t* i_vg_req;
    const int32_t* l_dev_cnt; const int64_t* l_dev_cap; const int32_t* l_dev_media; const int32_t* i_dev_alloc;
    const int32_t* l_spec_of; const simon_local_spec* l_specs;
    int64_t* st_vg /*[S][N][SIMON_MAX_VG]*/; int32_t* st_dev /*[S][N]*/;
    // per-scenario mutable state of the optional features, [S_chunk][...]
    int64_t* st_req_eph; int64_t* st_nz_cpu; int64_t* st_nz_mem;
    int64_t* st_scalar /*[S][K][N]*/; int64_t* st_gpu /*[S][N][8]*/;
    int64_t* st_gmax /*[S][N]: largest idle memory over the node's GPU devices (0 without devices)*/;
    int32_t* st_cnt /*[S][3*total_dom + Tm]: cnt_match, cnt_owner, w_owner, term_total*/;
    int32_t* st_seen /*[S][seen_stride]: distinct-domain stamps of the soft spread constraints*/;
    // explain outputs (single scenario)
    int32_t* failed_pods; uint16_t* fail_codes; int32_t* n_failed;
    long long* fail_detail /*[max_failed][n][4]: what Open-Local's error text carries (local_eval DETAIL), or null*/;
    // diagnostics (env SIMON_WIDE_PROF): per (scenario, wave) cycle sums of the phases of a cycle, or null
    unsigned long long* prof;
    // per-scenario nodeTree order (simon_set_node_ranks), rows of the WHOLE batch [S_total][N]: rank of a pool node / node of a
    // rank; null = pool order.  Row of this launch's scenario s: (scen_base + s) * N
    const int32_t* node_rank; const int32_t* node_inv;
    // devices Reserve booked for every placed GPU pod (simon_batch_out.gpu_slices), rows of the WHOLE batch [S_total][P]; null = not recorded
    uint64_t* gpu_slices;
};

struct WideArgs {
    int32_t N, P, K, Cn, S;                 // S = scenarios in THIS launch (chunk)
    int32_t scen_base;                      // index of this launch's first scenario in the batch
    int32_t mask_words, bc_words /*LDS words of base|class*/, n_sigs /*0 = table off*/, tab_nstride /*bytes per table row*/;
    uint32_t flags;                         // kArg* bits
    size_t tab_stride /*bytes per scenario*/;
    // static node arrays [N] (shared)
    const WideNodeStatic* node_static; const int32_t* alloc_pods; const int32_t* node_class;
    const uint64_t* static_mask; const int64_t* simon_raw;
    const uint32_t* mask_lanes /*[Cp][T]: bit it = static mask of node tid + it*T (lane-major copy of static_mask), or null*/;
    // stream
    const WidePod* pods; const int32_t* orders; const WideScenario* scen /*[S] of this chunk*/; const WideSig* sigs;
    // per-scenario mutable state every cycle touches, [S_chunk][N]
    int64_t* st_req_cpu; int64_t* st_req_mem; int32_t* st_npods;
    unsigned char* st_tab /*[S][n_sigs][tab_nstride]*/;
    // outputs of this chunk
    int32_t* unscheduled; int64_t* used_cpu; int64_t* used_mem; int64_t* used_vg /*or null*/; int32_t* placement /*[S][P] or null*/;
    const WideCold* cold;
};
constexpr uint32_t kArgGpu = 1u, kArgMask = 2u, kArgEph = 4u, kArgNzeq = 8u, kArgClassMode = 16u /*Cn <= 64*/,
                   kArgKey32 = 32u /*32-bit arg-max key*/, kArgProf = 64u, kArgTerms = 128u /*Tm > 0*/, kArgLocal = 256u /*Open-Local*/,
                   kArgRanked = 1024u /*per-scenario canonical node ranks (simon_set_node_ranks)*/,
                   kArgLean = 512u /*no spread constraint, scoring term, host port, required affinity or local volume anywhere*/,
                   // stage A leaves a feasible node's InterPodAffinity raw score / the counts of its soft spread constraints in LDS and
                   // stage B reads them back instead of gathering the counters a second time (wide_run sets them when the arrays fit)
                   kArgIpaCache = 2048u, kArgPtsCache = 4096u, kArgPtsSlotsShift = 13u /*3 bits: soft constraints per class, at most*/,
                   // more than 64 node classes (no class mode): the pod class's rows of the four (pod class, node class) tables are
                   // staged in LDS all the same when they fit (round 6: per feasible node they were up to seven dependent FLAT loads)
                   kArgRowsLds = 65536u;

// tuning / experiment knobs: environment variables read ONCE, when the context is created (simon_ctx_create)
struct WideKnobs {
    bool no_lean = false;            // SIMON_WIDE_NO_LEAN
    bool no_table = false;           // SIMON_WIDE_NO_TABLE
    bool no_rows_lds = false;        // SIMON_WIDE_NO_ROWS_LDS: more than 64 node classes read their class rows from global memory as before round 6 (A/B runs)
    int no_cache_b = 0;              // SIMON_WIDE_NO_CACHE_B (bit 0: InterPodAffinity scores, bit 1: soft-spread counts; no value = both): stage B gathers its counters again (A/B runs)
    bool prof = false;               // SIMON_WIDE_PROF (only builds with -DSIMON_WIDE_PROFILE act on it)
    size_t state_budget = 16ull << 30;   // SIMON_STATE_BUDGET_MB: HBM for per-scenario state of the all-feature kernel
    bool no_zmask = false;   // env SIMON_WIDE_NO_ZMASK: distinct zones by atomic stamps instead of presence masks (A/B)
    bool no_ident = false;   // env SIMON_WIDE_NO_IDENT: domain rows loaded even where the domain of node j is j (A/B)
};

struct WideDevice {
    WideKnobs knobs;
    void* blobs[96] = {};
    int n_blobs = 0;
    int state_chunk = 0;   // scenarios whose state is allocated
    int total_dom = 0, seen_stride = 0;
    bool has_eph = false, nzeq = false;
    // typed views
    WideNodeStatic* node_static = nullptr;
    int64_t *alloc_eph = nullptr, *scalar_alloc = nullptr, *gpu_mem_total = nullptr;
    int32_t *alloc_pods = nullptr, *node_class = nullptr, *gpu_cnt = nullptr, *topo_dom = nullptr;
    int64_t *i_req_cpu = nullptr, *i_req_mem = nullptr, *i_req_eph = nullptr, *i_nz_cpu = nullptr, *i_nz_mem = nullptr,
            *i_scalar_req = nullptr, *i_gpu_used = nullptr;
    int32_t* i_npods = nullptr;
    uint64_t* static_mask = nullptr; uint8_t* static_reason = nullptr; int64_t* simon_raw = nullptr;
    int64_t *na_raw = nullptr, *tt_raw = nullptr, *static_add = nullptr;
    int32_t *term_key = nullptr, *term_dom_off = nullptr, *term_set = nullptr, *anti_off = nullptr, *anti_idx = nullptr,
            *match_off = nullptr, *match_idx = nullptr, *manti_off = nullptr, *manti_idx = nullptr, *mown_off = nullptr, *mown_idx = nullptr, *aff_off = nullptr, *aff_idx = nullptr, *port_off = nullptr, *port_idx = nullptr, *pref_off = nullptr,
            *pref_idx = nullptr, *pref_w = nullptr, *own_off = nullptr, *own_idx = nullptr, *own_w = nullptr,
            *sh_off = nullptr, *sh_idx = nullptr, *sh_skew = nullptr, *sh_self = nullptr, *sh_set = nullptr,
            *sh_first_reg = nullptr, *ss_off = nullptr, *ss_idx = nullptr, *ss_skew = nullptr, *key_seen_off = nullptr;
    uint64_t* node_sets = nullptr;
    uint64_t *ss_ent = nullptr, *filt_ent = nullptr, *ipa_ent = nullptr;
    int32_t *filt_off = nullptr, *ipa_off = nullptr, *ipa_w = nullptr;
    uint8_t *class_flags = nullptr, *topo_is_hostname = nullptr;
    double* spread_log = nullptr;
    WidePod* pods = nullptr;
    WideSig* sigs = nullptr;
    int n_sigs = 0;
    int32_t *l_flags = nullptr, *l_vg_cnt = nullptr, *l_vg_name = nullptr, *l_dev_cnt = nullptr, *l_dev_media = nullptr,
            *i_dev_alloc = nullptr, *l_spec_of = nullptr, *st_dev = nullptr;
    int64_t *l_vg_cap = nullptr, *i_vg_req = nullptr, *l_dev_cap = nullptr, *st_vg = nullptr;
    simon_local_spec* l_specs = nullptr;
    WideCold* d_cold = nullptr;   // two slots: [0] batch runs, [1] explain
    unsigned char* st_tab = nullptr;
    // state
    int64_t *st_req_cpu = nullptr, *st_req_mem = nullptr, *st_req_eph = nullptr, *st_nz_cpu = nullptr, *st_nz_mem = nullptr,
            *st_scalar = nullptr, *st_gpu = nullptr, *st_gmax = nullptr;
    uint32_t* mask_lanes = nullptr;      // lane-major static mask of the last launch's workgroup size
    int mask_lanes_T = 0;
    int32_t *st_npods = nullptr, *st_cnt = nullptr, *st_seen = nullptr;
    void release();
};

// staging / launching (simon_wide.hip)
// launches of the instantiations that live outside simon_wide.hip (simon_wide_local.hip: the Open-Local variant; simon_wide_explain.hip:
// the EXPLAIN form of the full variant); T in {64, 256, 512, 1024}
hipError_t wide_launch_local(const WideArgs& a, int T, bool explain, size_t lds, hipStream_t st);
hipError_t wide_launch_explain(const WideArgs& a, int T, size_t lds, hipStream_t st);
int wide_stage(WideDevice& w, const HostInputs& in, hipStream_t st, std::string& err);
int wide_run(WideDevice& w, const HostInputs& in, const WideScenario* d_scen, const int32_t* h_perm_unused, int S,
             const int32_t* d_orders, int max_n, int T, int32_t* d_unsched, int64_t* d_used_cpu, int64_t* d_used_mem,
             int64_t* d_used_vg, int32_t* d_place, const int32_t* d_node_rank, const int32_t* d_node_inv, uint64_t* d_gpu_slices,
             hipStream_t st, std::string& err);
int wide_explain(WideDevice& w, const HostInputs& in, int n_nodes, const int32_t* order, int32_t* failed_pods,
                 uint16_t* fail_codes, int32_t max_failed, int T, const int32_t* d_rank_row, const int32_t* d_inv_row, hipStream_t st,
                 std::string& err, std::vector<int64_t>* local_detail = nullptr);

}  // namespace simon
