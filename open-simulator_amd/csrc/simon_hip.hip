// simon_hip.hip -- C-ABI of libsimon_hip.so (include/simon_hip.h): context, input staging into
// HBM (SoA), kernel-variant selection, launches, result fetch, min-plan reduction.
//
// Data layout in HBM (one copy per context, shared by every scenario of a batch):
//   nodes   NARROW: a_cpu/a_mem u32 (gcd-normalised), a_pods/ncls i32, initial state u32/i32
//           WIDE:   the int64 arrays of simon_nodes_soa as they come
//   pods    PodRowN[P] (32 B rows, read through scalar loads) / WidePod[P]
//   orders  int32 [n_orders][P];  scenarios {n_nodes, order_id}[S];  perm[S] (LPT launch order)
//   tables  simon_raw as i32/i64 [Cp][Cn]; static_mask u64 [Cp][ceil(N/64)]
//   out     unscheduled i32[S], used_cpu/used_mem i64[S], placement i32[S][P] (optional)
#include "../../include/simon_hip.h"
#include "simon_device.h"
#include "simon_wide.h"
#include "simon_table.h"

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <numeric>
#include <set>
#include <tuple>
#include <string>
#include <vector>

namespace simon {
struct PodRowF { double req_c, req_m, nz_c, nz_m; int32_t cls, preset, gate; uint32_t flags; };
struct FastScalars { int32_t mask_words, Cn, Cp, P, S; uint64_t g_cpu, g_mem; };
struct FastLaunch {
    const uint32_t *a_cpu, *a_mem; const int32_t *a_pods, *ncls; const uint32_t *i_rq_cpu, *i_rq_mem, *i_nz_cpu, *i_nz_mem;
    const int32_t* i_npods; const PodRowF* pods; const int32_t* orders; const ScenarioDesc* scen; const int32_t* perm;
    const uint64_t* static_mask; const int32_t* simon_raw; int32_t* unscheduled; int64_t *used_cpu, *used_mem;
    int32_t* placement; FastScalars sc;
};
hipError_t launch_fast(const FastLaunch& a, int T, int slots, bool has_mask, bool nzeq, size_t lds_bytes, hipStream_t st);
hipError_t launch_narrow(const NarrowArgs& a, int T, int slots, bool has_mask, bool rcp_div, size_t lds_bytes,
                         hipStream_t st);
}  // namespace simon

using namespace simon;

namespace {

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    ~DevBuf() { release(); }
    void release() { if (p) { (void)hipFree(p); p = nullptr; n = 0; } }
    hipError_t ensure(size_t count) {
        if (count <= n && p) return hipSuccess;
        release();
        hipError_t e = hipMalloc((void**)&p, std::max<size_t>(count, 1) * sizeof(T));
        if (e == hipSuccess) n = std::max<size_t>(count, 1);
        return e;
    }
    hipError_t upload(const std::vector<T>& h, hipStream_t st) {
        hipError_t e = ensure(h.size());
        if (e != hipSuccess || h.empty()) return e;
        return hipMemcpyAsync(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, st);
    }
};

struct PlanKernelOut {
    unsigned long long key;  // (n_nodes << 32) | scenario, min-reduced; ~0 = none
};

}  // namespace

struct simon_ctx : simon::HostInputs {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::string err;
    // host copies of the inputs live in the HostInputs base (caller buffers are never retained)
    bool have_nodes = false, have_pods = false, have_tables = false, staged = false, wide_staged = false;
    // ---- variant decision ----
    int variant = 0;
    uint64_t g_cpu = 1, g_mem = 1;
    bool rcp_div = true;
    int force_T = 0;  // env SIMON_WG
    bool force_v1 = false;  // env SIMON_NARROW_V1: use simon_narrow.hip even when simon_fast.hip applies
    bool force_wide = false;  // env SIMON_FORCE_WIDE
    int n_cus = 256;          // compute units of the device (simon_ctx_create)
    size_t lds_pad = 0;       // env SIMON_CACHE_LDS_PAD (bytes): occupancy experiment knob -- extra LDS per workgroup
    // ---- device buffers ----
    DevBuf<uint32_t> d_a_cpu, d_a_mem, d_i_rq_cpu, d_i_rq_mem, d_i_nz_cpu, d_i_nz_mem;
    DevBuf<int32_t> d_a_pods, d_ncls, d_i_npods, d_raw32;
    DevBuf<PodRowN> d_podsN;
    DevBuf<PodRowF> d_podsF;
    bool fast_ok = false, nzeq = false;  // simon_fast.hip eligibility
    // ---- simon_table.hip (score table, one wave per scenario) ----
    bool table_ok = false, table_perm_ok = false, no_cache = false;  // no_cache: env SIMON_NO_CACHE (run generation 2 instead)
    DevBuf<unsigned char> d_ws, d_table_cold;
    DevBuf<unsigned long long> d_table_prof, d_ws_off;
    int n_sigs = 0, Cn_t = 0;                    // request signatures; internal node classes = distinct (node_class, allocatable) pairs
    DevBuf<SigRow> d_sigs;
    DevBuf<ShapeRow> d_shapes;                   // [Cn_t]: capacity + reciprocals of an internal class
    DevBuf<PodRowC> d_podsC;
    DevBuf<int32_t> d_t_ncls, d_rank, d_clsprefix, d_inv_orders, d_place_step, d_cls_list, d_cls_off, d_t_raw;
    std::vector<int32_t> h_clsprefix, scen_ni;   // [(N+1)][Cn_t]; padded (class-major) size of every loaded scenario
    size_t ws_total = 0;
    // REST path of the score-table kernel (Open-Gpu-Share + node-level required anti-affinity): decided by choose_variant
    std::vector<int32_t> h_ncls_t, h_cls_off;    // internal node class of every node; class offsets into the per-class node lists
    DevBuf<int32_t> d_rk_ids, d_rk_pos;          // per-scenario node order for the score-table kernel (simon_set_node_ranks)
    bool table_ranks_ok = false;
    bool has_static = false;                     // static score tables present (score-table kernel: class term; else all-feature kernel)
    DevBuf<int32_t> d_t_na, d_t_tt, d_t_add;
    bool raw_fits_lds = true;                    // generations 1 / 2 can run (else: score table or the all-feature kernel)
    bool no_gpu_split = false, force_table = false, no_class_content = false, no_gpu_fold = false;
    bool debug_route = false;                     // env SIMON_DEBUG_ROUTE: one line per run on stderr with what decided the kernel
    bool gfold = false;                           // Open-Gpu-Share folded into the score table: the GPU request is part of the signature (gfold_supported)
    bool table_prof = false;                     // env SIMON_TABLE_PROF in -DSIMON_TABLE_PROFILE builds
    bool rest = false, no_rest = false;          // no_rest: env SIMON_NO_REST (such problems take the all-feature kernel)
    // fold: required anti-affinity / host ports on node-level topology keys as MONOTONE INFEASIBILITY of the score table -- a pod that
    // lands on a node zeroes the node's byte of every signature it excludes there, for good (fold_supported); env SIMON_NO_FOLD
    bool fold = false, no_fold = false;
    int fold_sigs = 0;                            // upper bound of the signatures the fold needs
    int gfold_sigs = 0, gfold_base_sigs = 0;      // ... and the GPU fold (request shape x table class x filter class x GPU request); the same without the GPU request
    std::vector<int32_t> fold_fc;                 // [Cp] filter class of a pod class
    std::vector<uint8_t> fold_x;                  // [FC][FC]: a pod of filter class L on a node excludes filter class S from it
    int fold_FC = 0;
    DevBuf<uint32_t> d_foldx;                     // [K][ceil(K / 32)]: bit S of row L = signature L landing excludes signature S
    uint64_t g_gpu = 1, g_eph = 1;               // gcd of every GPU memory / ephemeral-storage quantity
    uint64_t gfold_g = 1;                        // the GPU fold's own gcd: rest_supported() is also called as a PROBE and resets g_gpu
    std::vector<int> zone_keys;                  // topology keys of terms that are not node-level (REST: a domain = many positions)
    std::vector<int> key_zslot;                  // [Kt] index into zone_keys or -1 (node-level / unused)
    DevBuf<int32_t> d_zdom;
    bool xres = false;                           // some pod requests ephemeral storage or an extended resource
    int rest_M = 0, rest_G = 0, rest_X = 0;      // mask rows; GPU requests; extra-resource requests
    DevBuf<uint32_t> d_xsig, d_xalloc, d_i_xused;
    DevBuf<int32_t> d_xrows, d_gpu_cnt;
    DevBuf<uint32_t> d_gpu_devtot, d_i_gused;
    DevBuf<uint2> d_gsig;
    // SPREAD path of the score-table kernel (soft PodTopologySpread constraints, generation 7): decided by choose_variant
    bool sig_twins = false, no_sig_twins = false; // upper-half signatures sit 64 slots above a twin (same request, other table class); env SIMON_TABLE_NO_TWINS
    // InterPodAffinity preferred terms in self-referential form, scored in spread_select's table (spread_supported); env SIMON_NO_IPA_FOLD
    bool hard_fold = false, no_hard_fold = false; // hard spread constraints on zone-like keys as per-class verdicts of spread_select; env SIMON_NO_HARD_FOLD
    bool ipa_fold = false, no_ipa_fold = false;
    bool no_cn2 = false;                          // env SIMON_NO_CN2: generation 7 stays at 64 node classes (A/B + tests)
    // REST && SPREAD (round 6, simon_table_rs.hip): soft spread constraints next to GPU requests / required anti-affinity / ports / extra
    // resources that do not fold into the table -- generation 7's walks over generation 6's position-mask rows.  rs_probe: spread_supported
    // called to answer "would the walks take it if the rows carried the filters"; env SIMON_NO_RS: never (A/B + tests)
    bool rs = false, rs_probe = false, no_rs = false;
    DevBuf<int32_t> d_sp_word;                   // [P] SPREAD descriptor by pod id (TableCold::sp_word)
    std::vector<int32_t> ipa_h_term, ipa_h_w;     // [Cp] the hostname-like term of a class's raw score (-1: none) and its coefficient
    std::vector<std::vector<std::pair<int32_t, int32_t>>> ipa_z;   // [Cp] (zone-like term, coefficient)
    bool spread = false, no_spread = false;      // no_spread: env SIMON_NO_SPREAD (such problems take the all-feature kernel)
    // team mode of generation 7 (simon_table.hip: NW = kTeamWaves waves per scenario): env SIMON_TEAM = 0 never / 1 (or 4) always /
    // unset: batches of at most team_max_s scenarios (default 2 per CU: beyond that one wave per scenario fills the SIMDs by
    // itself -- measured crossover between 512 and 1 024 scenarios, profiles/r04; env SIMON_TEAM_MAX_S)
    int team_mode = -1, team_max_s = -1;
    int ldsws_mode = -1;                         // env SIMON_LDS_WS = 0 never / 1 whenever it fits / unset: batches of at most one scenario per CU (simon_table.hip: LDSWS)
    std::vector<int> sp_kind, sp_row, sp_zslot;  // per term: 0 unused / 1 hostname-like / 2 zone-like; its counter row; zone key slot
    std::vector<int> sp_set_eff;                  // per id: the node set its row is counted on (-1: every node)
    std::vector<int> sp_rep;                      // per term: the first term with the same key, node set and matching classes (they share a counter row)
    std::vector<int> sp_zkeys;                   // topology keys of the zone-like soft terms (<= kSpreadMaxZoneKeys): the class split
    int sp_TH = 0, sp_TZ = 0;
    DevBuf<int32_t> d_sp_ent;                    // (entry, term row) pairs: TableCold::sp_ent
    DevBuf<double> d_spread_log;
    DevBuf<uint64_t> d_node_sets;
    DevBuf<signed char> d_cls_zdom;
    bool table_coarse = false;                   // two-level summary (simon_table.hip: COARSE), decided per loaded batch
    int force_coarse = -1, table_ni_top = 16;    // env SIMON_TABLE_COARSE = 0 / 1 (A/B)
    std::vector<int32_t> h_perm;
    std::vector<int64_t> explain_detail;   // [failed][n][4] of the last explain (Open-Local error sizes), simon_explain_local_detail
    int explain_nodes = 0;
    bool explain_ran = false;
    std::vector<int32_t> h_orders;   // host copy of the loaded orders (simon_explain_loaded replays one of them)
    DevBuf<uint64_t> d_mask, d_t_mask;           // static masks by pod class; by table class (simon_table.hip)
    DevBuf<int64_t> d_prefix_cpu, d_prefix_mem, d_prefix_vg;
    DevBuf<int32_t> d_node_rank, d_node_inv;      // [S][N] per-scenario nodeTree ranks (simon_set_node_ranks)
    bool has_ranks = false;
    std::vector<int64_t> prefix_vg;   // [N+1] Open-Local VG capacity of the first n nodes (0 without local storage)
    WideDevice wide;
    // scenarios
    int S = 0, n_orders = 0, max_n = 0;
    std::vector<ScenarioDesc> scen;
    DevBuf<ScenarioDesc> d_scen;
    DevBuf<int32_t> d_orders, d_perm;
    // outputs
    DevBuf<int32_t> d_unsched, d_place;
    DevBuf<int32_t> d_prio;                       // ABI v6: spec.priority per pod (simon_set_pod_priorities)
    DevBuf<unsigned char> d_risk;                 // [S] DefaultPreemption could have acted in the scenario (simon_fetch_preempt_risk)
    bool prio_staged = false, have_risk = false;
    DevBuf<int64_t> d_used_cpu, d_used_mem, d_used_vg;
    DevBuf<unsigned long long> d_plan;
    bool have_results = false, have_placement = false, have_slices = false;
    DevBuf<uint64_t> d_gpu_slices;      // [S][P] devices Reserve booked (simon_batch_out.gpu_slices), recorded on request
    simon_stats stats{};
};

namespace {

int fail(simon_ctx* c, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    return code;
}

#define HIP_TRY(c, call)                                                                         \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return fail((c), e_ == hipErrorOutOfMemory ? SIMON_ENOMEM : SIMON_ENODEV, "%s: %s", \
                        #call, hipGetErrorString(e_));                                           \
    } while (0)

template <class T>
void copy_opt(std::vector<T>& dst, const T* src, size_t n, T fill = T()) {
    if (src) dst.assign(src, src + n);
    else dst.assign(n, fill);
}

uint64_t gcd_u64(uint64_t a, uint64_t b) { while (b) { uint64_t t = a % b; a = b; b = t; } return a; }

uint64_t gcd_of(std::initializer_list<const std::vector<int64_t>*> vs) {
    uint64_t g = 0;
    for (auto* v : vs) for (int64_t x : *v) { if (x < 0) return 0; g = gcd_u64(g, (uint64_t)x); if (g == 1) return 1; }
    return g ? g : 1;
}

// Can the score-table kernel's REST path take this problem's GPU-share / topology-term features?  Terms: only required
// anti-affinity (match + anti lists; everything else is excluded by v2_features) on topology keys that give every node its own
// domain (kubernetes.io/hostname), counted on every node.  GPU: every quantity a multiple of a gcd with quotients < 2^31, at
// most kTableMaxGpuSigs distinct (gpu-mem, gpu-count) requests.  Fills g_gpu.
bool rest_supported(simon_ctx* c) {
    if (c->no_rest) return false;
    if (c->has_gpu_index) return false;   // pods that arrive with a gpu-index annotation (their own filter and Reserve rule): all-feature kernel
    if (c->Tm > kTableMaxTerms && !c->rs) return false;
    // a term's topology key is node-level (every node its own domain: kubernetes.io/hostname) or zone-like (a domain = several
    // nodes, some nodes without the label): at most 6 of the latter, domain ids below 65 535
    std::vector<int> key_kind(std::max(c->Kt, 1), -1);   // 1 node-level, 0 zone-like
    c->zone_keys.clear();
    c->key_zslot.assign(std::max(c->Kt, 1), -1);
    // (REST && SPREAD: the rows of a term nobody holds against others are never read -- stage_narrow drops them -- so the terms the spread
    // constraints alone count may carry node sets and any key)
    std::vector<char> row_read(std::max(c->Tm, 1), c->rs ? 0 : 1);
    if (c->rs)
        for (int cp = 0; cp < c->Cp; ++cp) {
            for (int e = c->anti_off.empty() ? 0 : c->anti_off[cp]; !c->anti_off.empty() && e < c->anti_off[cp + 1]; ++e) if (c->anti_idx[e] >= 0 && c->anti_idx[e] < c->Tm) row_read[c->anti_idx[e]] = 1;
            for (int e = c->port_off.empty() ? 0 : c->port_off[cp]; !c->port_off.empty() && e < c->port_off[cp + 1]; ++e) if (c->port_idx[e] >= 0 && c->port_idx[e] < c->Tm) row_read[c->port_idx[e]] = 1;
            for (int e = c->aff_off.empty() ? 0 : c->aff_off[cp]; !c->aff_off.empty() && e < c->aff_off[cp + 1]; ++e) if (c->aff_idx[e] >= 0 && c->aff_idx[e] < c->Tm) row_read[c->aff_idx[e]] = 1;
        }
    if (c->rs) {                                                          // (... and only those get rows: stage_narrow)
        int n_read = 0;
        for (int t = 0; t < c->Tm; ++t) n_read += row_read[t];
        if (n_read > kTableMaxTerms) return false;
    }
    for (int t = 0; t < c->Tm; ++t) {
        if (!row_read[t]) continue;
        if (!c->term_set.empty() && c->term_set[t] >= 0) return false;
        const int k = c->term_key[t];
        if (key_kind[k] < 0) {
            std::vector<char> seen(c->N, 0);
            key_kind[k] = 1;
            for (int j = 0; j < c->N; ++j) {
                const int d = c->topo_dom[(size_t)k * c->N + j];
                if (d < -1 || d >= 65535) return false;
                if (d < 0 || d >= c->N || seen[d]) key_kind[k] = 0; else seen[d] = 1;
            }
            if (!key_kind[k]) {
                if (c->zone_keys.size() == 6) return false;
                c->key_zslot[k] = (int)c->zone_keys.size();
                c->zone_keys.push_back(k);
            }
        }
    }
    c->g_eph = 1;
    static_assert(SIMON_MAX_SCALAR == 4, "simon_table.hip keeps ephemeral storage + 4 extended resources in 5 components");
    if (c->xres) {
        // Extra resources as mask rows: the rows stand for `Allocatable < request + Requested`, which is also what fitsRequest asks of
        // pods that request NONE of them (ephemeral storage is compared whatever the request) -- exact as long as Requested never
        // exceeds Allocatable: nothing over-committed at the start, and no pod bound by Spec.NodeName (no filter) requests any.
        uint64_t g = 0;
        const uint64_t lim31 = 1ull << 31;
        for (int j = 0; j < c->N; ++j) {
            if (c->i_req_eph[j] < 0 || c->i_req_eph[j] > c->alloc_eph[j]) return false;
            g = gcd_u64(gcd_u64(g, (uint64_t)c->alloc_eph[j]), (uint64_t)c->i_req_eph[j]);
            for (int k = 0; k < c->K; ++k) {
                const int64_t a = c->scalar_alloc[(size_t)k * c->N + j], u = c->i_scalar_req[(size_t)k * c->N + j];
                if (a < 0 || u < 0 || u > a || (uint64_t)a >= lim31) return false;
            }
        }
        std::set<std::vector<int64_t>> sigs;
        for (int p = 0; p < c->P; ++p) {
            std::vector<int64_t> key{c->p_req_eph[p]};
            bool any = c->p_req_eph[p] != 0;
            if (c->p_req_eph[p] < 0) return false;
            for (int k = 0; k < c->K; ++k) {
                const int64_t q = c->p_scalar[(size_t)k * c->P + p];
                if (q < 0 || (uint64_t)q >= lim31) return false;
                key.push_back(q); any = any || q != 0;
            }
            if (!any) continue;
            if (c->p_preset[p] >= 0) return false;
            g = gcd_u64(g, (uint64_t)c->p_req_eph[p]);
            sigs.insert(key);
            if ((int)sigs.size() > kTableMaxXres) return false;
        }
        if (!g) g = 1;
        for (int j = 0; j < c->N; ++j) if ((uint64_t)c->alloc_eph[j] / g >= lim31) return false;
        for (int p = 0; p < c->P; ++p) if ((uint64_t)c->p_req_eph[p] / g >= lim31) return false;
        c->g_eph = g;
    }
    c->g_gpu = 1;
    if (c->has_gpu) {
        uint64_t g = 0;
        auto take = [&](int64_t x) { if (x < 0) return false; g = gcd_u64(g, (uint64_t)x); return true; };
        for (int j = 0; j < c->N; ++j) {
            if (c->gpu_mem_total[j] < 0) return false;
            if (c->gpu_cnt[j] > 0 && !take(c->gpu_mem_total[j] / c->gpu_cnt[j])) return false;
            for (int d = 0; d < SIMON_MAX_GPU_DEV; ++d) if (!take(c->i_gpu_used[(size_t)j * SIMON_MAX_GPU_DEV + d])) return false;
        }
        for (int p = 0; p < c->P; ++p) if (c->p_gpu_mem[p] > 0) take(c->p_gpu_mem[p]);
        if (!g) g = 1;
        const uint64_t lim31 = 1ull << 31;
        for (int j = 0; j < c->N; ++j) {
            if (c->gpu_cnt[j] > 0 && (uint64_t)(c->gpu_mem_total[j] / c->gpu_cnt[j]) / g >= lim31) return false;
            for (int d = 0; d < SIMON_MAX_GPU_DEV; ++d) if ((uint64_t)c->i_gpu_used[(size_t)j * SIMON_MAX_GPU_DEV + d] / g >= lim31) return false;
        }
        std::set<std::pair<uint64_t, int>> sigs;
        for (int p = 0; p < c->P; ++p) {
            if (c->p_gpu_mem[p] <= 0) continue;
            if ((uint64_t)c->p_gpu_mem[p] / g >= lim31) return false;
            sigs.insert({(uint64_t)c->p_gpu_mem[p] / g, std::min(c->p_gpu_cnt[p], 64)});
            if ((int)sigs.size() > kTableMaxGpuSigs) return false;
        }
        c->g_gpu = g;
    }
    return true;
}

// Open-Gpu-Share FOLDED into the score table (round 4).  "GPU request g no longer fits node j" is monotone -- device memory is only ever
// booked -- which is what a 0 in the (signature, node) table already means.  With the GPU request made part of the request signature
// the assume of a GPU pod books the devices (Reserve) and zeroes the bytes of the signatures whose request stopped fitting; a GPU pod
// then needs NO select-time filter: it takes the summary scan, or generation 7's walk when a Service selects it.  Worth it when the
// signatures stay few (real gpushare workloads: a handful of Deployments) -- config 5's 42 shapes x 8 GPU requests keep the position
// masks of generation 6.  Needs: no arriving gpu-index lists, GPU quantities on a gcd with quotients < 2^31 (fills g_gpu), at most
// 1 023 signatures once the GPU request is part of them (beyond 128 only under generation 7's walks: choose_variant).
bool gfold_supported(simon_ctx* c) {
    if (c->no_gpu_fold || c->no_rest || !c->has_gpu || c->has_gpu_index || c->gpu_cnt.empty()) return false;   // (SIMON_NO_REST keeps meaning: GPU problems on the all-feature kernel)
    uint64_t g = 0;
    auto take = [&](int64_t x) { if (x < 0) return false; g = gcd_u64(g, (uint64_t)x); return true; };
    for (int j = 0; j < c->N; ++j) {
        if (c->gpu_mem_total[j] < 0) return false;
        if (c->gpu_cnt[j] > 0 && !take(c->gpu_mem_total[j] / c->gpu_cnt[j])) return false;
        for (int d = 0; d < SIMON_MAX_GPU_DEV; ++d) if (!take(c->i_gpu_used[(size_t)j * SIMON_MAX_GPU_DEV + d])) return false;
    }
    for (int p = 0; p < c->P; ++p) if (c->p_gpu_mem[p] > 0) take(c->p_gpu_mem[p]);
    if (!g) g = 1;
    const uint64_t lim31 = 1ull << 31;
    for (int j = 0; j < c->N; ++j) {
        if (c->gpu_cnt[j] > 0 && (uint64_t)(c->gpu_mem_total[j] / c->gpu_cnt[j]) / g >= lim31) return false;
        for (int d = 0; d < SIMON_MAX_GPU_DEV; ++d) if ((uint64_t)c->i_gpu_used[(size_t)j * SIMON_MAX_GPU_DEV + d] / g >= lim31) return false;
    }
    for (int p = 0; p < c->P; ++p) if (c->p_gpu_mem[p] > 0 && (uint64_t)c->p_gpu_mem[p] / g >= lim31) return false;
    // signatures the table would need: (request, content of the class's static rows as stage_narrow interns them, GPU request)
    const int Cp = c->Cp;
    std::vector<int32_t> content_of(Cp);
    {
        const size_t words = (size_t)(c->N + 63) / 64;
        std::map<std::pair<std::vector<uint64_t>, std::vector<int64_t>>, int> ids;
        for (int cp = 0; cp < Cp; ++cp) {
            std::vector<uint64_t> mrow;
            if (c->has_mask) mrow.assign(c->static_mask.begin() + (size_t)cp * words, c->static_mask.begin() + (size_t)(cp + 1) * words);
            std::vector<int64_t> rrow;
            for (const std::vector<int64_t>* tab : {&c->simon_raw, &c->na_raw, &c->tt_raw, &c->static_add})
                if (!tab->empty()) rrow.insert(rrow.end(), tab->begin() + (size_t)cp * c->Cn, tab->begin() + (size_t)(cp + 1) * c->Cn);
            content_of[cp] = ids.emplace(std::make_pair(std::move(mrow), std::move(rrow)), (int)ids.size()).first->second;
        }
    }
    std::set<std::tuple<int64_t, int64_t, int64_t, int64_t, int32_t, int32_t, int64_t, int32_t>> sig;
    std::set<std::tuple<int64_t, int64_t, int64_t, int64_t, int32_t>> base;   // ... and without the GPU request / filter class: generation 6's signatures
    for (int p = 0; p < c->P; ++p) {
        const bool gp = c->p_gpu_mem[p] > 0;
        if ((int)base.size() <= 128) base.insert(std::make_tuple(c->p_req_cpu[p], c->p_req_mem[p], c->p_nz_cpu[p], c->p_nz_mem[p], content_of[c->p_cls[p]]));
        sig.insert(std::make_tuple(c->p_req_cpu[p], c->p_req_mem[p], c->p_nz_cpu[p], c->p_nz_mem[p], content_of[c->p_cls[p]],
                                   c->fold ? c->fold_fc[c->p_cls[p]] : 0, gp ? c->p_gpu_mem[p] : 0, gp ? std::min(c->p_gpu_cnt[p], 64) : 0));
        if ((int)sig.size() > kTableMaxSigs) return false;
    }
    c->gfold_sigs = (int)sig.size();
    c->gfold_base_sigs = (int)base.size();
    c->gfold_g = g;
    c->g_gpu = g;
    return true;
}

// Can the score-table kernel's SPREAD path (generation 7) take this problem?  Of the ABI v2 features: SOFT PodTopologySpread constraints
// (+ the static score tables the class term folds in), InterPodAffinity PREFERRED terms whose owners hold one weight per pod with at
// most one hostname-like counter per class, HARD spread constraints on zone-like keys whose eligible nodes are all the
// labelled ones (5.3f), required anti-affinity / host ports where they fold into the table (fold_supported, 5.3d).  Not: required
// affinity, Open-Gpu-Share / extra-resource rows (the REST path), Open-Local.  Every term with a counter sits on a hostname-like key
// (flagged topo_is_hostname, every node its own domain: one counter byte per position) or on a zone-like key (<= 16 domains; <= 3 such
// keys: the node classes are split by their domains).  Fills sp_kind / sp_row / sp_zslot / sp_zkeys / sp_rep / sp_set_eff and the
// per-class entries of the preferred score.
bool spread_supported(simon_ctx* c) {
    c->ipa_fold = false;
    c->ipa_h_term.clear(); c->ipa_h_w.clear(); c->ipa_z.clear();
    c->hard_fold = false;
    if (c->no_spread || (c->ss_idx.empty() && !c->has_ipa_score && c->sh_idx.empty())) return false;
    if (!c->sh_idx.empty() && c->no_hard_fold) return false;
    if (c->has_ipa_score && c->no_ipa_fold) return false;
    if (c->has_local || (!c->aff_idx.empty() && !c->rs_probe)) return false;   // (required affinity: rows that must be SET -- the mask rows, rs_probe)
    if ((!c->anti_idx.empty() || !c->port_idx.empty()) && !c->fold && !c->rs_probe) return false;   // required anti-affinity / ports: folded into the table, or (rs_probe) on the mask rows
    if (c->has_gpu_index || (c->has_gpu && !c->gfold && !c->rs_probe)) return false;      // (GPU share: folded into the table, gfold_supported -- or on the mask rows)
    if (c->topo_is_hostname.empty()) return false;
    if (!c->ss_idx.empty() && c->spread_log.size() < (size_t)c->N + 1) return false;
    int64_t max_pods = 0;
    // the per-position counters are bytes.  Pods that pass NodeResourcesFit number at most alloc_pods on a node; pods bound by
    // Spec.NodeName (presets) skip the filters (V/eventhandlers.go:223-236) and come on top
    {
        std::vector<int32_t> preset_on(c->N, 0);
        for (int p = 0; p < c->P && !c->p_preset.empty(); ++p)
            if (c->p_preset[p] >= 0 && c->p_preset[p] < c->N) ++preset_on[c->p_preset[p]];
        for (int j = 0; j < c->N; ++j) {
            const int64_t x = (int64_t)c->alloc_pods[j] + preset_on[j];
            if (x > 255) return false;
            max_pods = std::max<int64_t>(max_pods, x);
        }
    }
    // ids: t < Tm = the pods MATCHING term t; Tm + t = the pods OWNING a scoring term t (one weight per pod: InterPodAffinity below)
    c->sp_kind.assign(2 * (size_t)c->Tm, 0); c->sp_row.assign(2 * (size_t)c->Tm, 0); c->sp_zslot.assign(2 * (size_t)c->Tm, 0);
    c->sp_zkeys.clear();
    c->sp_TH = c->sp_TZ = 0;
    std::vector<int> key_kind(std::max(c->Kt, 1), -1);                    // 1 hostname-like, 2 zone-like, 0 unusable
    // Terms on one key, counted on the same nodes, that the same pod classes match with the same multiplicities have the same
    // counters at all times (a Service's default constraint and the chart's anti-affinity term with the same selector): they share
    // ONE row, kept under the first of them (sp_rep).
    c->sp_rep.assign(2 * (size_t)c->Tm, 0);
    std::vector<long long> own_w_of(c->Tm, 0);                            // per-pod weight of the owners of term t (0: nobody owns it)
    bool own_uniform = true;
    {
        std::vector<std::vector<std::pair<int, int>>> who(2 * (size_t)c->Tm);
        for (int cp = 0; cp < c->Cp && !c->own_off.empty(); ++cp) {     // the owner classes of a scoring term, when they all hold one weight per pod
            std::map<int, long long> os;
            for (int e = c->own_off[cp]; e < c->own_off[cp + 1]; ++e) { const int t = c->own_idx[e]; if (t < 0 || t >= c->Tm) return false; os[t] += c->own_w[e]; }
            for (auto& kv : os) {
                if (kv.second == 0) continue;
                if (own_w_of[kv.first] == 0) own_w_of[kv.first] = kv.second;
                else if (own_w_of[kv.first] != kv.second) own_uniform = false;
                who[(size_t)c->Tm + kv.first].push_back(std::make_pair(cp, 1));
            }
        }
        for (int cp = 0; cp < c->Cp && !c->match_off.empty(); ++cp) {
            std::map<int, int> mult;
            for (int e = c->match_off[cp]; e < c->match_off[cp + 1]; ++e) { const int t = c->match_idx[e]; if (t < 0 || t >= c->Tm) return false; ++mult[t]; }
            for (auto& kv : mult) who[kv.first].push_back(std::make_pair(cp, kv.second));
        }
        const int words = (c->N + 63) / 64;                               // a node set that holds every node counts like no node set
        std::vector<char> full(std::max(c->R, 1), 0);
        for (int r = 0; r < c->R && (size_t)(r + 1) * words <= c->node_sets.size(); ++r) {
            long long n_in = 0;
            for (int wd = 0; wd < words; ++wd) n_in += __builtin_popcountll(c->node_sets[(size_t)r * words + wd] & (wd == words - 1 && (c->N & 63) ? (1ull << (c->N & 63)) - 1 : ~0ull));
            full[r] = n_in == c->N;
        }
        // A node set that leaves out only nodes WITHOUT the term's own label changes no counter (a zone-like key: such a node has no
        // domain); on a hostname-like key the set never matters to a reader at all -- a node outside it either fails the class's node
        // affinity (never feasible) or lacks a constraint key (IgnoredNodes: never scored).  Such terms join the unrestricted ones; a
        // group is kept under an unrestricted member when it has one (a reader of the preferred score wants the count everywhere).
        std::map<std::pair<int, int>, char> harmless;
        auto set_of = [&](int id) -> int {
            if (id >= c->Tm || c->term_set.empty()) return -1;            // (owners are counted wherever they land)
            const int set = c->term_set[id], k = c->term_key[id];
            if (set < 0 || set >= c->R || full[set]) return -1;
            if (c->topo_is_hostname[k]) return -1;
            auto it = harmless.find(std::make_pair(k, set));
            if (it == harmless.end()) {
                bool ok = true;
                for (int j = 0; j < c->N && ok; ++j)
                    if (!((c->node_sets[(size_t)set * words + (j >> 6)] >> (j & 63)) & 1ull)) ok = c->topo_dom[(size_t)k * c->N + j] < 0;
                it = harmless.emplace(std::make_pair(k, set), (char)ok).first;
            }
            return it->second ? -1 : set;
        };
        std::map<std::tuple<int, int, std::vector<std::pair<int, int>>>, int> first;
        for (int pass = 0; pass < 2; ++pass)
            for (int id = 0; id < 2 * c->Tm; ++id) {
                const int t = id < c->Tm ? id : id - c->Tm;
                const bool unrestricted = id >= c->Tm || c->term_set.empty() || c->term_set[id] < 0;
                if (unrestricted != (pass == 0)) continue;
                c->sp_rep[id] = first.emplace(std::make_tuple(c->term_key[t], set_of(id), who[id]), id).first->second;
            }
        // the set a row is COUNTED on: none where it does not matter -- members of one group may carry different (harmless) sets, and each
        // reads inside its own only, where counting everywhere and counting on the set agree
        c->sp_set_eff.assign(2 * (size_t)c->Tm, -1);
        for (int id = 0; id < c->Tm; ++id) c->sp_set_eff[id] = set_of(id);
    }
    // a term gets a counter row: a byte per position (hostname-like key) or a word per domain (zone-like key)
    auto classify = [&](int t) -> int {
        if (t < 0 || t >= 2 * c->Tm) return 0;
        const int r = c->sp_rep[t];
        const int k = c->term_key[r < c->Tm ? r : r - c->Tm];
        if (key_kind[k] < 0) {
            key_kind[k] = 0;
            if (c->topo_is_hostname[k]) {                                 // size = scored nodes (scoring.go:100-104): needs one domain per node
                std::vector<char> seen(c->N, 0);
                bool ok = true;
                for (int j = 0; j < c->N && ok; ++j) {
                    const int d = c->topo_dom[(size_t)k * c->N + j];
                    ok = d >= 0 && d < c->N && !seen[d];
                    if (ok) seen[d] = 1;
                }
                key_kind[k] = ok ? 1 : 0;
            } else {
                bool ok = c->topo_n_dom[k] <= kSpreadMaxZoneDom;
                for (int j = 0; j < c->N && ok; ++j) { const int d = c->topo_dom[(size_t)k * c->N + j]; ok = d >= -1 && d < kSpreadMaxZoneDom; }
                if (ok && (int)c->sp_zkeys.size() < kSpreadMaxZoneKeys) { key_kind[k] = 2; c->sp_zkeys.push_back(k); }
            }
        }
        if (key_kind[k] == 0) return 0;
        if (c->sp_kind[r] == 0) {
            c->sp_kind[r] = key_kind[k];
            if (key_kind[k] == 1) c->sp_row[r] = c->sp_TH++;
            else {
                c->sp_row[r] = c->sp_TZ++;
                c->sp_zslot[r] = (int)(std::find(c->sp_zkeys.begin(), c->sp_zkeys.end(), k) - c->sp_zkeys.begin());
            }
        }
        if (t != r) { c->sp_kind[t] = c->sp_kind[r]; c->sp_row[t] = c->sp_row[r]; c->sp_zslot[t] = c->sp_zslot[r]; }
        return key_kind[k];
    };
    for (int cp = 0; cp < c->Cp && !c->ss_idx.empty(); ++cp) {
        for (int e = c->ss_off[cp]; e < c->ss_off[cp + 1]; ++e) {
            const int maxskew = c->ss_skew[e] & ~SIMON_SPREAD_DUP_KEY;
            if (maxskew < 1 || maxskew >= (1 << 14)) return false;
            if (!classify(c->ss_idx[e])) return false;
        }
    }
    // Hard (DoNotSchedule) constraints on ZONE-like keys: a per-class verdict in spread_select (the classes are split by zone).  The
    // eligibility set of an entry (filtering.go:236-251: which nodes register a zone) may leave out unlabelled nodes only.
    if (!c->sh_idx.empty()) {
        const int words = (c->N + 63) / 64;
        for (int cp = 0; cp < c->Cp; ++cp) {
            if (c->sh_off[cp + 1] - c->sh_off[cp] > 3) return false;
            for (int e = c->sh_off[cp]; e < c->sh_off[cp + 1]; ++e) {
                const int t = c->sh_idx[e];
                if (t < 0 || t >= c->Tm || c->sh_skew[e] < 1 || c->sh_skew[e] >= (1 << 14)) return false;
                if (classify(t) != 2) return false;                       // hostname-like hard constraints: the all-feature kernel
                const int set = c->sh_set.empty() ? -1 : c->sh_set[e], k = c->term_key[t];
                if (set >= 0) {
                    if (set >= c->R) return false;
                    for (int j = 0; j < c->N; ++j)
                        if (!((c->node_sets[(size_t)set * words + (j >> 6)] >> (j & 63)) & 1ull) && c->topo_dom[(size_t)k * c->N + j] >= 0) return false;
                }
            }
        }
        // a pinned pod (one admissible node) is decided from the table byte of that node alone: it must not carry a hard constraint
        for (int p = 0; p < c->P && !c->p_pin.empty(); ++p)
            if (c->p_pin[p] >= 0 && c->sh_off[c->p_cls[p] + 1] > c->sh_off[c->p_cls[p]]) return false;
        c->hard_fold = true;
    }
    // InterPodAffinity preferred terms (scoring.go:87-271) in their usual, SELF-REFERENTIAL form: the raw score of pod class cp on a node is
    //   sum_e pref_w[e] * cnt_match[t_e][dom] + sum_{t in match(cp)} w_owner[t][dom],
    // and w_owner[t] = w_t * cnt_match[t] whenever every class owns t with w_t times the multiplicity it matches t with (the pods a
    // workload's (anti-)affinity term selects are the workload's own).  Then raw = sum_t coef(cp, t) * cnt_match[t][dom]: with at most ONE
    // hostname-like term per class -- the one its soft spread constraint counts, if it has one -- the raw score is a function of (class,
    // count) like the spread score and joins it in spread_select's table; zone-like terms add a per-class constant.
    if (c->has_ipa_score) {
        if (c->match_off.empty()) return false;
        const int Cp = c->Cp, T = c->Tm;
        std::vector<std::map<int, long long>> prefsum(Cp), mult(Cp);
        for (int cp = 0; cp < Cp; ++cp) {
            if (!c->pref_off.empty()) for (int e = c->pref_off[cp]; e < c->pref_off[cp + 1]; ++e) prefsum[cp][c->pref_idx[e]] += c->pref_w[e];
            for (int e = c->match_off[cp]; e < c->match_off[cp + 1]; ++e) mult[cp][c->match_idx[e]] += 1;
        }
        // w_owner[t] = (the owners' weight per pod) x (owner pods in the domain): a counter row of its own (id Tm + t) -- the SAME row as
        // cnt_match[t]'s when the owners are the matchers (sp_rep), the self-referential case
        if (!own_uniform) return false;
        c->ipa_h_term.assign(Cp, -1); c->ipa_h_w.assign(Cp, 0); c->ipa_z.assign(Cp, {});
        for (int cp = 0; cp < Cp; ++cp) {
            std::map<int, long long> coef;
            for (auto& kv : prefsum[cp]) { if (kv.first < 0 || kv.first >= T) return false; coef[c->sp_rep[kv.first]] += kv.second; }   // (per counter row)
            for (auto& kv : mult[cp]) if (own_w_of[kv.first] != 0) coef[c->sp_rep[(size_t)T + kv.first]] += kv.second * own_w_of[kv.first];
            int spread_host = -1;                                         // the class's own hostname-like soft constraint, if any
            for (int e = c->ss_idx.empty() ? 0 : c->ss_off[cp]; !c->ss_idx.empty() && e < c->ss_off[cp + 1]; ++e)
                if (c->sp_kind[c->ss_idx[e]] == 1) { const int r = c->sp_rep[c->ss_idx[e]]; if (spread_host >= 0 && spread_host != r) return false; spread_host = r; }
            for (auto& kv : coef) {
                if (kv.second == 0) continue;
                if (std::llabs(kv.second) >= (1ll << 20)) return false;
                if (kv.first < T && !c->term_set.empty() && c->term_set[kv.first] >= 0) return false;
                const int kind = classify(kv.first);
                if (kind == 0) return false;
                if (kind == 1) {
                    if (c->ipa_h_term[cp] >= 0) return false;             // two per-node counters: not a function of one count
                    if (spread_host >= 0 && spread_host != kv.first) return false;
                    c->ipa_h_term[cp] = kv.first; c->ipa_h_w[cp] = (int32_t)kv.second;
                } else {
                    if (c->ipa_z[cp].size() == 3) return false;
                    c->ipa_z[cp].push_back(std::make_pair(kv.first, (int32_t)kv.second));
                }
            }
            // a class whose spread constraints take the general walk of spread_select (several hostname-like constraints, or one behind a
            // zone-like one) keeps the all-feature kernel when it also has a preferred term
            if ((c->ipa_h_term[cp] >= 0 || !c->ipa_z[cp].empty()) && !c->ss_idx.empty()) {
                int nh = 0, first_kind = 0, ns = c->ss_off[cp + 1] - c->ss_off[cp];
                for (int e = c->ss_off[cp]; e < c->ss_off[cp + 1]; ++e) { nh += c->sp_kind[c->ss_idx[e]] == 1; if (e == c->ss_off[cp]) first_kind = c->sp_kind[c->ss_idx[e]]; }
                if (!(nh == 0 || (nh == 1 && first_kind == 1 && ns <= 2))) return false;
            }
        }
        c->ipa_fold = true;
    }
    if (c->sp_TH > kSpreadMaxHostTerms || c->sp_TZ > kSpreadMaxZoneTerms || c->R > 4094) return false;
    // counted terms of a class: the terms with a counter row among its match list, with multiplicity; one lane per distinct term
    for (int cp = 0; cp < c->Cp && !c->match_off.empty(); ++cp) {
        std::map<int, int> mult;                                           // per counter ROW (terms that share one count once)
        {
            std::map<int, int> per_term;
            for (int e = c->match_off[cp]; e < c->match_off[cp + 1]; ++e) if (c->sp_kind[c->match_idx[e]]) ++per_term[c->match_idx[e]];
            for (auto& kv : per_term) mult[c->sp_rep[kv.first]] = kv.second;
            for (int e = c->own_off.empty() ? 0 : c->own_off[cp]; !c->own_off.empty() && e < c->own_off[cp + 1]; ++e)      // rows that count this class as an owner
                if (c->sp_kind[(size_t)c->Tm + c->own_idx[e]] && own_w_of[c->own_idx[e]] != 0) mult[c->sp_rep[(size_t)c->Tm + c->own_idx[e]]] = 1;
        }
        const int ns = c->ss_idx.empty() ? 0 : c->ss_off[cp + 1] - c->ss_off[cp];
        const int ni = (c->ipa_fold ? (c->ipa_h_term[cp] >= 0 ? 1 : 0) + (int)c->ipa_z[cp].size() : 0) + (c->hard_fold ? c->sh_off[cp + 1] - c->sh_off[cp] : 0);
        if ((int)mult.size() + ns + ni > 64) return false;
        for (auto& kv : mult) if ((int64_t)kv.second * max_pods > 255) return false;   // a byte counter: pods on a node x multiplicity
    }
    return true;
}

// Required anti-affinity (both directions, filtering.go:319-346) and NodePorts conflicts (node_ports.go:104-127) on node-level topology
// keys never leave the node the pod lands on, and pods never leave: "signature S cannot use node j any more" is the monotone
// infeasibility the score table already knows (byte 0 stays 0).  The fold needs: only anti / match / port lists among the
// placement-dependent filters, every such term on a key that gives every node its own domain, counted on every node; the filter
// class (anti list, relevant match list, port list) becomes part of the request signature.
bool fold_supported(simon_ctx* c) {
    c->fold_fc.clear(); c->fold_x.clear(); c->fold_FC = 0; c->fold_sigs = 0;
    if (c->no_fold || c->Tm <= 0 || (c->anti_idx.empty() && c->port_idx.empty())) return false;
    // (GPU share next to the fold: only when it is folded too -- choose_variant drops the fold again when gfold_supported says no)
    if (!c->aff_idx.empty() || c->has_gpu_index || (c->has_gpu && (c->no_gpu_fold || c->no_rest)) || c->match_off.empty()) return false;
    const int Cp = c->Cp, N = c->N;
    std::vector<char> relevant(c->Tm, 0);
    auto list_of = [&](const std::vector<int32_t>& off, const std::vector<int32_t>& idx, int cp) {
        std::vector<int32_t> v;
        if (!off.empty()) v.assign(idx.begin() + off[cp], idx.begin() + off[cp + 1]);
        std::sort(v.begin(), v.end()); v.erase(std::unique(v.begin(), v.end()), v.end());
        return v;
    };
    for (int cp = 0; cp < Cp; ++cp) {
        for (int t : list_of(c->anti_off, c->anti_idx, cp)) { if (t < 0 || t >= c->Tm) return false; relevant[t] = 1; }
        for (int t : list_of(c->port_off, c->port_idx, cp)) { if (t < 0 || t >= c->Tm) return false; relevant[t] = 1; }
    }
    std::vector<int> key_ok(std::max(c->Kt, 1), -1);
    for (int t = 0; t < c->Tm; ++t) {
        if (!relevant[t]) continue;
        if (!c->term_set.empty() && c->term_set[t] >= 0) return false;
        const int k = c->term_key[t];
        if (key_ok[k] < 0) {
            std::vector<char> seen(N, 0);
            key_ok[k] = 1;
            for (int j = 0; j < N && key_ok[k]; ++j) {
                const int d = c->topo_dom[(size_t)k * N + j];
                if (d < 0 || d >= N || seen[d]) key_ok[k] = 0; else seen[d] = 1;
            }
        }
        if (!key_ok[k]) return false;
    }
    // filter classes by content
    std::map<std::tuple<std::vector<int32_t>, std::vector<int32_t>, std::vector<int32_t>>, int> fc_id;
    std::vector<std::tuple<std::vector<int32_t>, std::vector<int32_t>, std::vector<int32_t>>> fcs;
    c->fold_fc.assign(Cp, 0);
    for (int cp = 0; cp < Cp; ++cp) {
        std::vector<int32_t> anti = list_of(c->anti_off, c->anti_idx, cp), port = list_of(c->port_off, c->port_idx, cp), match;
        for (int t : list_of(c->match_off, c->match_idx, cp)) { if (t < 0 || t >= c->Tm) return false; if (relevant[t]) match.push_back(t); }
        auto key = std::make_tuple(anti, match, port);
        auto it = fc_id.emplace(key, (int)fcs.size());
        if (it.second) fcs.push_back(key);
        c->fold_fc[cp] = it.first->second;
    }
    const int FC = (int)fcs.size();
    if (FC > 4096) return false;
    auto meets = [](const std::vector<int32_t>& a, const std::vector<int32_t>& b) {       // sorted lists share an element
        size_t i = 0, j = 0;
        while (i < a.size() && j < b.size()) { if (a[i] == b[j]) return true; if (a[i] < b[j]) ++i; else ++j; }
        return false;
    };
    c->fold_x.assign((size_t)FC * FC, 0);
    for (int L = 0; L < FC; ++L)
        for (int S = 0; S < FC; ++S)       // L has landed: S meets a pod matching one of its anti terms / ports, or a pod requiring a term S matches
            c->fold_x[(size_t)L * FC + S] = meets(std::get<0>(fcs[S]), std::get<1>(fcs[L])) || meets(std::get<1>(fcs[S]), std::get<0>(fcs[L])) ||
                                            meets(std::get<2>(fcs[S]), std::get<1>(fcs[L]));
    c->fold_FC = FC;
    // signatures the table would need: (request, content of the class's static rows as stage_narrow interns them, filter class)
    std::vector<int32_t> content_of(Cp);
    {
        const size_t words = (size_t)(N + 63) / 64;
        std::map<std::pair<std::vector<uint64_t>, std::vector<int64_t>>, int> ids;
        for (int cp = 0; cp < Cp; ++cp) {
            std::vector<uint64_t> mrow;
            if (c->has_mask) mrow.assign(c->static_mask.begin() + (size_t)cp * words, c->static_mask.begin() + (size_t)(cp + 1) * words);
            std::vector<int64_t> rrow;
            for (const std::vector<int64_t>* tab : {&c->simon_raw, &c->na_raw, &c->tt_raw, &c->static_add})
                if (!tab->empty()) rrow.insert(rrow.end(), tab->begin() + (size_t)cp * c->Cn, tab->begin() + (size_t)(cp + 1) * c->Cn);
            content_of[cp] = ids.emplace(std::make_pair(std::move(mrow), std::move(rrow)), (int)ids.size()).first->second;
        }
    }
    std::set<std::tuple<int64_t, int64_t, int64_t, int64_t, int32_t, int32_t>> sig;
    for (int p = 0; p < c->P; ++p) {
        sig.insert(std::make_tuple(c->p_req_cpu[p], c->p_req_mem[p], c->p_nz_cpu[p], c->p_nz_mem[p], content_of[c->p_cls[p]], c->fold_fc[c->p_cls[p]]));
        if ((int)sig.size() > kTableMaxSigs) return false;
    }
    c->fold_sigs = (int)sig.size();
    return true;
}

// LDS of ONE workgroup of the two-level score-table instantiations: a gfx950 workgroup may hold all 160 KB of its CU (the launch sets
// hipFuncAttributeMaxDynamicSharedMemorySize).  Past 64 KB a CU holds two, then one scenario -- still far ahead of the all-feature kernel
// for the problems that need it (hundreds of signatures x thousands of nodes under generation 7's walks); the one-level layout, which
// exists for occupancy, keeps its 64 KB bound.
static constexpr size_t kTableLdsMaxWG = 159 * 1024;

// Decide NARROW vs WIDE and compute the gcd normalisation.
// NARROW needs: cpu+mem+pods only; every quantity non-negative; after dividing by the gcd all
// node totals and the worst-case accumulated NonZeroRequested stay < 2^31; simon raw scores fit
// 30 bits; original quantities < 2^53 (exact in fp64, so BalancedAllocation is scale-invariant).
void choose_variant(simon_ctx* c) {
    c->variant = SIMON_KERNEL_WIDE;
    c->g_cpu = c->g_mem = 1;
    if (c->force_wide) return;
    c->rest = false;
    c->spread = false;
    c->has_static = c->has_na || c->has_tt || c->has_add;
    c->fold = fold_supported(c);
    c->gfold = gfold_supported(c);
    if (c->fold && c->has_gpu && !c->gfold) c->fold = false;        // GPU share on position masks (generation 6) takes the terms along
    c->rs = false;
    if (c->v2_features_but_ports_and_static()) {
        if (!spread_supported(c)) {                               // only soft spread constraints: generation 7 of the score-table kernel
            // ... or its walks over the position-mask rows, when GPU share / required anti-affinity / ports are what keeps it out and the
            // walk scores soft constraints only
            if (c->no_rs || c->no_rest) return;
            c->rs_probe = true;
            const bool ok = spread_supported(c);
            c->rs_probe = false;
            if (c->debug_route) fprintf(stderr, "[route] walks over the mask rows: probe %d ipa %d hard %d soft %d\n", (int)ok, (int)c->ipa_fold, (int)c->hard_fold, (int)!c->ss_idx.empty());
            if (!ok) return;
            c->rs = true;
            c->fold = false; c->gfold = false;
        }
        c->spread = true;
    }
    if (c->has_static) {
        // The class term of the score-table kernel takes them if byte (<= 201) + 2 x Simon (<= 200) + NodeAffinity (<= 100) +
        // TaintToleration (<= 100) + the largest addition stays within the 10 bits the two-level summary gives a total.
        int64_t add_max = 0;
        for (int64_t x : c->na_raw) if (x < 0 || x >= (1ll << 30)) return;
        for (int64_t x : c->tt_raw) if (x < 0 || x >= (1ll << 30)) return;
        for (int64_t x : c->static_add) { if (x < 0) return; add_max = std::max(add_max, x); }
        if (201 + 200 + (c->has_na ? 100 : 0) + (c->has_tt ? 100 : 0) + add_max > 1023) return;
    }
    // Ephemeral storage and extended resources take part only when somebody requests them: with no request and nothing requested at
    // the start, fitsRequest's `Allocatable < request + Requested` (fit.go:264-299) is 0-false on every node whatever the allocatable.
    for (int64_t x : c->alloc_eph) if (x < 0) return;
    c->xres = false;
    for (int64_t x : c->i_req_eph) if (x) c->xres = true;
    for (int64_t x : c->p_req_eph) if (x) c->xres = true;
    for (int64_t x : c->i_scalar_req) if (x) c->xres = true;
    for (int64_t x : c->p_scalar) if (x) c->xres = true;
    if (c->fold && c->xres) c->fold = false;                        // (extra-resource rows live on the REST path)
    // anti-affinity / ports without soft spread constraints: the fold while two signatures per lane hold them, else the position masks
    if (c->fold && !c->spread && c->fold_sigs > 128 && rest_supported(c)) c->fold = false;
    // ... and GPU share alike: with more than 128 signatures the device rows of generation 6 beat the signature groups read from memory
    // (generation 6 itself holds <= 128 signatures: where the requests alone are more, the fold with its signature groups is what is left)
    if (c->gfold && !c->spread && c->gfold_sigs > 128 && c->gfold_base_sigs <= 128 && rest_supported(c)) { c->gfold = false; c->fold = false; }
    // the GPU fold serves problems that need no other per-node filter row: plain cpu+memory+GPU, and generation 7's (Services next to GPU pods)
    if (c->gfold && ((!c->spread && !c->fold && c->Tm > 0) || c->xres)) { c->gfold = false; if (c->has_gpu) c->fold = false; }
    if (c->spread && !c->rs && !c->no_rs && !c->no_rest && c->xres && c->aff_idx.empty()) {
        c->rs = true; c->fold = false; c->gfold = false;            // extra-resource rows under the walks: the same instantiation
    }
    const bool wants_rest = (!c->spread || c->rs) && !c->fold && ((c->has_gpu && !c->gfold) || c->Tm > 0 || c->xres);
    // rest_supported() above ran as a probe and leaves g_gpu = 1 when it says no (e.g. more than kTableMaxGpuSigs GPU requests): a fold
    // that survives the probes stages its quantities on ITS gcd -- the 31-bit bounds were checked against that one, not against 1
    if (c->gfold) c->g_gpu = c->gfold_g;
    if (c->spread && !c->rs && c->xres) { c->spread = false; return; }       // extra-resource rows live on the REST path: all-feature kernel
    if (c->spread && !c->rs && !c->fold && (!c->anti_idx.empty() || !c->port_idx.empty())) { c->spread = false; return; }
    if (c->rs && !wants_rest) c->rs = false;
    if (wants_rest && !rest_supported(c)) { if (c->debug_route) fprintf(stderr, "[route] mask rows refused (rs %d)\n", (int)c->rs); return; }
    if (c->N >= (1 << 20) - 1) return;
    const uint64_t gc = gcd_of({&c->alloc_cpu, &c->i_req_cpu, &c->i_nz_cpu, &c->p_req_cpu, &c->p_nz_cpu});
    const uint64_t gm = gcd_of({&c->alloc_mem, &c->i_req_mem, &c->i_nz_mem, &c->p_req_mem, &c->p_nz_mem});
    if (!gc || !gm) return;  // negative quantity somewhere
    const uint64_t lim53 = 1ull << 53, lim31 = 1ull << 31;
    auto bounded = [&](const std::vector<int64_t>& alloc, const std::vector<int64_t>& irq, const std::vector<int64_t>& inz,
                       const std::vector<int64_t>& prq, const std::vector<int64_t>& pnz, uint64_t g) {
        uint64_t max_node = 0, sum_p = 0;
        for (size_t j = 0; j < alloc.size(); ++j)
            max_node = std::max({max_node, (uint64_t)alloc[j], (uint64_t)irq[j], (uint64_t)inz[j]});
        for (size_t p = 0; p < prq.size(); ++p) {
            sum_p += std::max((uint64_t)prq[p], (uint64_t)pnz[p]);
            if (sum_p >= lim53) return false;
        }
        if (max_node + sum_p >= lim53) return false;
        return (max_node + sum_p) / g < lim31;
    };
    if (!bounded(c->alloc_cpu, c->i_req_cpu, c->i_nz_cpu, c->p_req_cpu, c->p_nz_cpu, gc)) return;
    if (!bounded(c->alloc_mem, c->i_req_mem, c->i_nz_mem, c->p_req_mem, c->p_nz_mem, gm)) return;
    for (int64_t r : c->simon_raw) if (r < 0 || r >= (1ll << 30)) return;
    // generations 1 and 2 keep simon_raw [Cp][Cn] in LDS; the score-table kernel reads it by table class from global memory
    c->raw_fits_lds = (size_t)c->Cp * c->Cn * sizeof(int32_t) <= 60 * 1024;
    c->variant = SIMON_KERNEL_NARROW;
    c->g_cpu = gc;
    c->g_mem = gm;
    c->rest = wants_rest;
}

int stage_narrow(simon_ctx* c) {
    const int N = c->N, P = c->P;
    std::vector<uint32_t> a_cpu(N), a_mem(N), rqc(N), rqm(N), nzc(N), nzm(N);
    for (int j = 0; j < N; ++j) {
        a_cpu[j] = (uint32_t)(c->alloc_cpu[j] / c->g_cpu);
        a_mem[j] = (uint32_t)(c->alloc_mem[j] / c->g_mem);
        rqc[j] = (uint32_t)(c->i_req_cpu[j] / c->g_cpu);
        rqm[j] = (uint32_t)(c->i_req_mem[j] / c->g_mem);
        nzc[j] = (uint32_t)(c->i_nz_cpu[j] / c->g_cpu);
        nzm[j] = (uint32_t)(c->i_nz_mem[j] / c->g_mem);
    }
    std::vector<PodRowN> rows(P);
    for (int p = 0; p < P; ++p) {
        PodRowN& r = rows[p];
        r.req_cpu = (uint32_t)(c->p_req_cpu[p] / c->g_cpu);
        r.req_mem = (uint32_t)(c->p_req_mem[p] / c->g_mem);
        r.nz_cpu = (uint32_t)(c->p_nz_cpu[p] / c->g_cpu);
        r.nz_mem = (uint32_t)(c->p_nz_mem[p] / c->g_mem);
        r.cls = c->p_cls[p];
        r.preset = c->p_preset[p];
        r.gate = c->p_gate[p];
        bool no_xres = c->p_req_eph[p] == 0;                     // fit.go:244-249: the early return needs EVERY request to be zero
        for (int k = 0; k < c->K && no_xres; ++k) no_xres = c->p_scalar[(size_t)k * c->P + p] == 0;
        if (!c->p_entries.empty() && c->p_entries[p]) no_xres = false;   // ... and to hold no ScalarResources ENTRY, not even one of quantity 0 (simon_set_scalar_entries)
        r.flags = (c->p_req_cpu[p] == 0 && c->p_req_mem[p] == 0 && no_xres) ? 1u : 0u;
    }
    std::vector<int32_t> raw32(c->simon_raw.begin(), c->simon_raw.end());
    hipStream_t st = c->stream;
    // simon_fast.hip: <= 32 node classes, no zero-capacity node; NZEQ when NonZeroRequested == Requested everywhere
    c->fast_ok = c->Cn <= 32 && (size_t)c->Cp * c->Cn * 8 <= 48 * 1024;
    c->nzeq = true;
    for (int j = 0; j < N; ++j) {
        if (a_cpu[j] == 0 || a_mem[j] == 0) c->fast_ok = false;
        if (rqc[j] != nzc[j] || rqm[j] != nzm[j]) c->nzeq = false;
    }
    std::vector<PodRowF> rowsF(P);
    for (int p = 0; p < P; ++p) {
        const PodRowN& r = rows[p];
        rowsF[p] = PodRowF{(double)r.req_cpu, (double)r.req_mem, (double)r.nz_cpu, (double)r.nz_mem, r.cls, r.preset, r.gate, r.flags};
        if (r.req_cpu != r.nz_cpu || r.req_mem != r.nz_mem) c->nzeq = false;
    }
    // simon_table.hip: intern pod request signatures and internal node classes
    bool nozero = true;
    for (int j = 0; j < N; ++j) if (a_cpu[j] == 0 || a_mem[j] == 0) nozero = false;
    c->table_ok = nozero && N <= kTableMaxNodesCoarse;
    if (c->table_ok) {
        // Table class of a pod class = its CONTENT: (static-mask row, Simon raw row).  Workload expansion hands over one pod class
        // per template (thousands with anti-affinity groups or DaemonSets) of which few differ in what this kernel reads, and a
        // request signature is (requests, table class).
        const size_t words = (size_t)(N + 63) / 64;
        std::map<std::pair<std::vector<uint64_t>, std::vector<int32_t>>, int> tc_id;
        std::vector<int32_t> tc_of(c->Cp), tc_rep;
        for (int cp = 0; cp < c->Cp; ++cp) {
            std::vector<uint64_t> mrow;
            if (c->has_mask) mrow.assign(c->static_mask.begin() + (size_t)cp * words, c->static_mask.begin() + (size_t)(cp + 1) * words);
            std::vector<int32_t> rrow(raw32.begin() + (size_t)cp * c->Cn, raw32.begin() + (size_t)(cp + 1) * c->Cn);
            for (const std::vector<int64_t>* tab : {&c->na_raw, &c->tt_raw, &c->static_add})       // static score rows are content too
                if (!tab->empty()) rrow.insert(rrow.end(), tab->begin() + (size_t)cp * c->Cn, tab->begin() + (size_t)(cp + 1) * c->Cn);
            auto it = tc_id.emplace(std::make_pair(std::move(mrow), std::move(rrow)), (int)tc_rep.size());
            if (it.second) tc_rep.push_back(cp);
            tc_of[cp] = it.first->second;
        }
        const int Ctc = (int)tc_rep.size();
        std::map<std::tuple<uint32_t, uint32_t, uint32_t, uint32_t, int32_t, uint32_t, int32_t, uint32_t, int32_t>, int> sig_id;
        std::vector<SigRow> sigs;
        std::vector<int32_t> sig_fc;          // fold: filter class of a signature
        std::vector<PodRowC> rowsC(P);
        for (int p = 0; p < P && c->table_ok; ++p) {
            const PodRowN& r = rows[p];
            const int32_t tc = tc_of[r.cls];
            const int32_t fc = c->fold ? c->fold_fc[r.cls] : 0;
            // GPU fold: the (gpu-mem per device in gcd units, device count) request is part of the signature
            const bool gp = c->gfold && c->p_gpu_mem[p] > 0;
            const uint32_t gq = gp ? (uint32_t)((uint64_t)c->p_gpu_mem[p] / c->g_gpu) : 0u;
            const int32_t gn = gp ? std::min(c->p_gpu_cnt[p], 64) : 0;
            auto key = std::make_tuple(r.req_cpu, r.req_mem, r.nz_cpu, r.nz_mem, tc, r.flags, fc, gq, gn);
            auto it = sig_id.find(key);
            if (it == sig_id.end()) {
                if ((int)sigs.size() == kTableMaxSigs) { c->table_ok = false; break; }
                it = sig_id.emplace(key, (int)sigs.size()).first;
                SigRow sr{};
                sr.req_c = r.req_cpu; sr.req_m = r.req_mem; sr.nz_c = r.nz_cpu; sr.nz_m = r.nz_mem; sr.cls = tc; sr.flags = r.flags;
                sr.pad[0] = (int32_t)gq; sr.pad[1] = gp ? (gn > 0 ? gn : -1) : 0;      // (pad[1] != 0 marks a GPU signature; a request without devices fits nowhere)
                sigs.push_back(sr);
                sig_fc.push_back(fc);
            }
            rowsC[p] = PodRowC{it->second | (tc << 10), (!c->p_pin.empty() && c->p_pin[p] >= 0) ? -2 - c->p_pin[p] : r.preset, r.gate, 0};
        }
        // 65 .. 128 signatures: the kernel keeps two per lane (k and k + 64).  When every signature of the upper half can sit 64 slots
        // above a TWIN -- the same request under another table class -- the kernel evaluates a lane's node byte once for both
        // (TableScalars::static_tables & 16; config 5: 42 request shapes x 2 table classes).  A permutation of the ids, nothing else.
        c->sig_twins = false;
        if (c->gfold && (int)sigs.size() > kTableMaxSigs) c->table_ok = false;      // (gfold_supported counted them: cannot happen)
        if (c->table_ok && !c->no_sig_twins && !c->gfold && sigs.size() > 64 && sigs.size() <= 128) {
            const int K = (int)sigs.size(), n_hi = K - 64;
            std::map<std::tuple<double, double, double, double, uint32_t>, std::vector<int>> by_shape;
            for (int k = 0; k < K; ++k) by_shape[std::make_tuple(sigs[k].req_c, sigs[k].req_m, sigs[k].nz_c, sigs[k].nz_m, sigs[k].flags)].push_back(k);
            std::vector<int> slot_of(K, -1), lo_rest;
            int paired = 0;
            for (auto& kv : by_shape) {
                std::vector<int>& v = kv.second;
                size_t i = 0;
                for (; i + 1 < v.size() && paired < n_hi; i += 2, ++paired) { slot_of[v[i]] = paired; slot_of[v[i + 1]] = 64 + paired; }
                for (; i < v.size(); ++i) lo_rest.push_back(v[i]);
            }
            if (paired == n_hi) {
                int next = n_hi;
                for (int k : lo_rest) slot_of[k] = next++;
                std::vector<SigRow> moved(K);
                std::vector<int32_t> moved_fc(K);
                for (int k = 0; k < K; ++k) { moved[slot_of[k]] = sigs[k]; moved_fc[slot_of[k]] = sig_fc[k]; }
                sigs.swap(moved); sig_fc.swap(moved_fc);
                for (int p = 0; p < P; ++p) rowsC[p].sigcls = (rowsC[p].sigcls & ~0x3FF) | slot_of[rowsC[p].sigcls & 0x3FF];
                c->sig_twins = true;
            }
        }
        // REST descriptors: term class (which mask rows a pod must find clear / sets) and GPU signature of every pod
        std::vector<int32_t> xrows;           // entries of every term class: filter row | set row << 16
        std::vector<uint2> gsigs;
        std::vector<uint32_t> xsigs;          // [X][8]
        c->rest_M = c->rest_G = c->rest_X = 0;
        if (c->rest && c->table_ok) {
            std::map<std::pair<uint32_t, int32_t>, int> gs_id;
            std::vector<int> gs_of(P, -1);
            for (int p = 0; p < P; ++p) {
                if (!c->has_gpu || c->p_gpu_mem[p] <= 0) continue;
                const auto key = std::make_pair((uint32_t)((uint64_t)c->p_gpu_mem[p] / c->g_gpu), (int32_t)std::min(c->p_gpu_cnt[p], 64));
                auto it = gs_id.emplace(key, (int)gsigs.size());
                if (it.second) gsigs.push_back(make_uint2(key.first, (unsigned)key.second));
                gs_of[p] = it.first->second;
            }
            std::map<std::vector<uint32_t>, int> xs_id;
            std::vector<int> xs_of(P, -1);
            for (int p = 0; p < P && c->xres; ++p) {
                std::vector<uint32_t> key(8, 0);
                key[0] = (uint32_t)((uint64_t)c->p_req_eph[p] / c->g_eph);
                bool any = key[0] != 0;
                for (int k = 0; k < c->K; ++k) { key[1 + k] = (uint32_t)c->p_scalar[(size_t)k * P + p]; any = any || key[1 + k] != 0; }
                if (!any) continue;
                auto it = xs_id.emplace(key, (int)xs_id.size());
                if (it.second) xsigs.insert(xsigs.end(), key.begin(), key.end());
                xs_of[p] = it.first->second;
            }
            const int Xn = (int)xs_id.size();
            // REST && SPREAD: the match lists also name the terms the spread constraints count -- rows nobody ever tests (and on a zone-like
            // key a set-row costs a pass over the scenario's blocks per landing pod): only the terms some class holds AGAINST others get
            // rows there (tm: term -> its row pair), so a cluster of hundreds of Services next to a few anti-affinity terms fits
            std::vector<char> row_read(std::max(c->Tm, 1), c->rs ? 0 : 1);
            if (c->rs)
                for (int cp = 0; cp < c->Cp; ++cp) {
                    for (int e = c->anti_off.empty() ? 0 : c->anti_off[cp]; !c->anti_off.empty() && e < c->anti_off[cp + 1]; ++e) row_read[c->anti_idx[e]] = 1;
                    for (int e = c->port_off.empty() ? 0 : c->port_off[cp]; !c->port_off.empty() && e < c->port_off[cp + 1]; ++e) row_read[c->port_idx[e]] = 1;
                    for (int e = c->aff_off.empty() ? 0 : c->aff_off[cp]; !c->aff_off.empty() && e < c->aff_off[cp + 1]; ++e) row_read[c->aff_idx[e]] = 1;
                }
            std::vector<int> tm(std::max(c->Tm, 1), 0);
            int n_rows_t = 0;
            for (int t = 0; t < c->Tm; ++t) if (row_read[t]) tm[t] = n_rows_t++;
            const int G = (int)gsigs.size(), T = n_rows_t, B = G + Xn;       // term rows start behind the request rows
            const int NZk = (int)c->zone_keys.size(), SCR = B + 2 * T, LBL = SCR + 1;
            // rows: requests | "a pod MATCHING t is here" | "a pod REQUIRING t is here" | scratch (set by entries that set nothing,
            // never read) | per zone-like key "the node lacks the label"
            c->rest_G = G; c->rest_X = Xn; c->rest_M = LBL + NZk;
            std::map<std::pair<std::vector<int32_t>, std::vector<int32_t>>, int> xc_id;   // -> n | offset << 6
            std::vector<int> xc_of(c->Cp, 0);

            for (int cp = 0; cp < c->Cp && T > 0; ++cp) {
                std::vector<int32_t> anti(c->anti_idx.begin() + c->anti_off[cp], c->anti_idx.begin() + c->anti_off[cp + 1]);
                std::vector<int32_t> match(c->match_idx.begin() + c->match_off[cp], c->match_idx.begin() + c->match_off[cp + 1]);
                std::vector<int32_t> aff;      // required affinity (filtering.go:348-377): DERIVED terms, in the class's order
                if (!c->aff_off.empty()) aff.assign(c->aff_idx.begin() + c->aff_off[cp], c->aff_idx.begin() + c->aff_off[cp + 1]);
                std::sort(aff.begin(), aff.end()); aff.erase(std::unique(aff.begin(), aff.end()), aff.end());
                const bool aff_self = !aff.empty() && !c->class_flags.empty() && (c->class_flags[cp] & SIMON_CLASS_AFF_SELF);
                std::vector<int32_t> port;     // NodePorts (node_ports.go:104-127): port terms this class CONFLICTS with
                if (!c->port_off.empty()) port.assign(c->port_idx.begin() + c->port_off[cp], c->port_idx.begin() + c->port_off[cp + 1]);
                std::sort(anti.begin(), anti.end()); anti.erase(std::unique(anti.begin(), anti.end()), anti.end());
                std::sort(match.begin(), match.end()); match.erase(std::unique(match.begin(), match.end()), match.end());
                match.erase(std::remove_if(match.begin(), match.end(), [&](int32_t t) { return !row_read[t]; }), match.end());
                std::sort(port.begin(), port.end()); port.erase(std::unique(port.begin(), port.end()), port.end());
                if (anti.empty() && match.empty() && port.empty() && aff.empty()) continue;
                size_t n_lbl = 0;              // one label entry per zone-like key among the affinity terms
                { std::set<int> zk; for (int t : aff) if (c->key_zslot[c->term_key[t]] >= 0) zk.insert(c->key_zslot[c->term_key[t]]); n_lbl = zk.size(); }
                const size_t n = anti.size() + match.size() + port.size() + aff.size() + n_lbl, off = xrows.size();
                if (n > 63 || off + n >= (1u << 14)) { c->table_ok = false; break; }   // one lane per entry; 14-bit offsets
                std::vector<int32_t> anti_key = anti;       // key: (anti list, -1, port list), match list
                anti_key.push_back(-1); anti_key.insert(anti_key.end(), port.begin(), port.end());
                anti_key.push_back(aff_self ? -3 : -2); anti_key.insert(anti_key.end(), aff.begin(), aff.end());
                auto it = xc_id.emplace(std::make_pair(anti_key, match), (int)(n | (off << 6)));
                if (it.second) {
                    // filter (filtering.go:319-346): a placed pod MATCHES one of my anti terms (row B + t), or a placed pod
                    // REQUIRES a term that matches me (row B + T + t); AddPod sets the mirror rows (oracle/simon_oracle.c: add_pod)
                    // bits 28..30: zone key + 1 of the term (the set-row then covers every position of the pod's domain)
                    auto zs = [&](int t) { return (unsigned)(c->key_zslot[c->term_key[t]] + 1) << 28; };
                    for (int t : anti) xrows.push_back((int)((unsigned)(B + tm[t]) | ((unsigned)(B + T + tm[t]) << 16) | zs(t)));
                    for (int t : match) xrows.push_back((int)((unsigned)(B + T + tm[t]) | ((unsigned)(B + tm[t]) << 16) | zs(t)));
                    // a conflicting port is in use on the node: a placed pod MATCHES (binds) port term t; nothing to set -- the
                    // entry's set-row repeats a row the class sets anyway, or a scratch row behind the last term row
                    for (int t : port) xrows.push_back((B + tm[t]) | (SCR << 16));
                    // required affinity: row B + t must be SET (bit 31; bit 27: the class matches its own terms -- the first-pod
                    // escape); on a zone-like key the node must carry the label whatever the escape says (label row, plain entry)
                    std::set<int> zk;
                    for (int t : aff) {
                        xrows.push_back((int)((unsigned)(B + tm[t]) | ((unsigned)SCR << 16) | (aff_self ? 1u << 27 : 0u) | (1u << 31)));
                        const int z = c->key_zslot[c->term_key[t]];
                        if (z >= 0 && zk.insert(z).second) xrows.push_back((LBL + z) | (SCR << 16));
                    }
                }
                xc_of[cp] = it.first->second;
            }
            bool any_rest = false;
            for (int p = 0; p < P; ++p) {
                rowsC[p].rest = (gs_of[p] + 1) | ((xs_of[p] + 1) << 6) | (int)((unsigned)xc_of[c->p_cls[p]] << 12);
                any_rest = any_rest || rowsC[p].rest != 0;
            }
            // GPU nodes without a GPU pod, terms no pod of the stream carries: nothing for the REST path to do -- generations 4 / 5
            // run the batch (their cycle is 17 % shorter than the REST instantiation's, profiles/README.md)
            if (!any_rest && c->table_ok) { c->rest = false; c->rs = false; c->rest_M = c->rest_G = c->rest_X = 0; }
        }
        // SPREAD descriptors: per pod class its soft constraints (term | maxSkew << 16 | dup << 30), the terms with a counter row its pods
        // are COUNTED on (term | multiplicity << 16) and the entries of its InterPodAffinity raw score (coefficient; the hostname-like term
        // first), interned by content; every entry travels with its term; PodRowC::rest = soft | counted << 3 | preferred << 10 | offset << 13
        std::vector<int32_t> sp_ent, sp_ent_term;
        if (c->spread && c->table_ok) {
            std::map<std::vector<int32_t>, int> sc_id;
            std::vector<int> desc_of(c->Cp, 0);
            for (int cp = 0; cp < c->Cp; ++cp) {
                std::vector<int32_t> ent, ent_term;
                const int ns = c->ss_idx.empty() ? 0 : c->ss_off[cp + 1] - c->ss_off[cp];
                for (int e = c->ss_idx.empty() ? 0 : c->ss_off[cp]; !c->ss_idx.empty() && e < c->ss_off[cp + 1]; ++e) {
                    ent.push_back(c->ss_idx[e] | ((c->ss_skew[e] & 0x3FFF) << 16) | ((c->ss_skew[e] & SIMON_SPREAD_DUP_KEY) ? 1 << 30 : 0));
                    ent_term.push_back(c->ss_idx[e]);
                }
                std::map<int, int> mult;                                   // per counter ROW (terms that share one count once)
                if (!c->match_off.empty()) {
                    std::map<int, int> per_term;
                    for (int e = c->match_off[cp]; e < c->match_off[cp + 1]; ++e) if (c->sp_kind[c->match_idx[e]]) ++per_term[c->match_idx[e]];
                    for (auto& kv : per_term) mult[c->sp_rep[kv.first]] = kv.second;
                }
                if (c->ipa_fold && !c->own_off.empty()) {                 // rows that count the class's pods as OWNERS of a scoring term (one per pod)
                    std::map<int, long long> os;
                    for (int e = c->own_off[cp]; e < c->own_off[cp + 1]; ++e) os[c->own_idx[e]] += c->own_w[e];
                    for (auto& kv : os) if (kv.second != 0 && c->sp_kind[(size_t)c->Tm + kv.first]) mult[c->sp_rep[(size_t)c->Tm + kv.first]] = 1;
                }
                for (auto& kv : mult) { ent.push_back((kv.first & 0xFFFF) | (kv.second << 16)); ent_term.push_back(kv.first); }
                int n_ipa = 0;
                if (c->ipa_fold) {
                    if (c->ipa_h_term[cp] >= 0) { ent.push_back(c->ipa_h_w[cp]); ent_term.push_back(c->ipa_h_term[cp]); ++n_ipa; }
                    for (auto& zt : c->ipa_z[cp]) { ent.push_back(zt.second); ent_term.push_back(zt.first); ++n_ipa; }
                }
                int n_hard = 0;
                if (c->hard_fold)
                    for (int e = c->sh_off[cp]; e < c->sh_off[cp + 1]; ++e, ++n_hard) {
                        ent.push_back((c->sh_skew[e] & 0x3FFF) | ((!c->sh_self.empty() && c->sh_self[e]) ? 1 << 14 : 0));
                        ent_term.push_back(c->sh_idx[e]);
                    }
                if (ent.empty()) continue;
                std::vector<int32_t> key = ent;                            // (entries of different kinds can spell the same word)
                key.insert(key.end(), ent_term.begin(), ent_term.end());
                key.push_back(ns); key.push_back(n_ipa); key.push_back(n_hard);
                auto it = sc_id.find(key);
                if (it == sc_id.end()) {
                    if (sp_ent.size() + ent.size() >= (1u << 16)) { c->table_ok = false; break; }
                    it = sc_id.emplace(key, ns | ((int)mult.size() << 3) | (n_ipa << 10) | (n_hard << 13) | ((int)sp_ent.size() << 15)).first;
                    sp_ent.insert(sp_ent.end(), ent.begin(), ent.end());
                    sp_ent_term.insert(sp_ent_term.end(), ent_term.begin(), ent_term.end());
                }
                desc_of[cp] = it->second;
            }
            if (c->rs) {                                                   // PodRowC::rest keeps the REST descriptor: the SPREAD word travels by pod id
                std::vector<int32_t> spw(std::max(P, 1), 0);
                for (int p = 0; p < P; ++p) spw[p] = desc_of[c->p_cls[p]];
                HIP_TRY(c, c->d_sp_word.upload(spw, st));
            } else {
                for (int p = 0; p < P && c->table_ok; ++p) rowsC[p].rest = desc_of[c->p_cls[p]];
            }
            if (c->Tm >= (1 << 16)) c->table_ok = false;
        }
        // Internal node class = (caller's node class, allocatable cpu, allocatable memory).  The caller's classes share their
        // allocatable by contract (include/simon_hip.h), so normally this IS the caller's partition; splitting a class that
        // does not keeps the kernel's "shape follows from class" exact for any input.
        // With GPU requests in the stream (REST) a class is also split into its nodes with and without devices when that fits: a GPU
        // request then excludes the units of a device-less class as a whole (one compare per 64 positions in rest_select) and a
        // class with devices holds candidates only.
        // SPREAD: a class is also split by the domains of the zone-like keys of the soft spread terms (sub = the domains, 5 bits each, 0 =
        // the label is missing): a summary unit then has ONE zone term and "which zones hold a scored node" follows from the feasible-node
        // counters per (signature, class).
        // The caller's node classes are interned by CONTENT first -- the class's column of every (pod class, node class) table the kernel
        // reads (Simon raw score, NodeAffinity preferred, TaintToleration PreferNoSchedule, weighted additions), over the table classes in
        // use.  A host that derives node classes from label / taint sets hands over hundreds (464 in profiles/e2e_sweep.py's mix of 3
        // capacities) of which few differ in those columns; the static filters are per node (static_mask), not per class.
        // (SIMON_TABLE_NO_CLASS_CONTENT=1: the caller's ids as they come -- A/B and tests.)
        std::vector<int32_t> content_of(std::max(c->Cn, 1));
        {
            std::map<std::vector<int32_t>, int> col_id;
            for (int nc = 0; nc < c->Cn; ++nc) {
                if (c->no_class_content) { content_of[nc] = nc; continue; }
                std::vector<int32_t> col;
                col.reserve((size_t)Ctc * 4);
                for (int tc = 0; tc < Ctc; ++tc) {
                    col.push_back(raw32[(size_t)tc_rep[tc] * c->Cn + nc]);
                    for (const std::vector<int64_t>* tab : {&c->na_raw, &c->tt_raw, &c->static_add})
                        if (!tab->empty()) col.push_back((int32_t)(*tab)[(size_t)tc_rep[tc] * c->Cn + nc]);
                }
                content_of[nc] = col_id.emplace(std::move(col), (int)col_id.size()).first->second;
            }
        }
        std::map<std::tuple<int32_t, uint32_t, uint32_t, int>, int> cls_id;
        std::vector<ShapeRow> shapes;
        std::vector<int32_t> orig_of, ncls_t(N), sub_of_class;
        bool split_gpu = c->rest && !c->spread && c->has_gpu && c->rest_G > 0 && !c->gpu_cnt.empty() && !c->no_gpu_split;
        if (split_gpu) {
            std::set<std::tuple<int32_t, uint32_t, uint32_t, int>> keys;
            for (int j = 0; j < N; ++j) keys.insert(std::make_tuple(content_of[c->node_class[j]], a_cpu[j], a_mem[j], c->gpu_cnt[j] > 0 ? 1 : 0));
            split_gpu = (int)keys.size() <= kTableMaxClasses;        // (the split is an optimisation: never at the price of the second class per lane)
        }
        auto zone_sub = [&](int j) -> int {
            int sub = 0;
            for (size_t z = 0; z < c->sp_zkeys.size(); ++z) sub |= (c->topo_dom[(size_t)c->sp_zkeys[z] * N + j] + 1) << (5 * z);
            return sub;
        };
        for (int j = 0; j < N && c->table_ok; ++j) {
            auto key = std::make_tuple(content_of[c->node_class[j]], a_cpu[j], a_mem[j], c->spread ? zone_sub(j) : (split_gpu && c->gpu_cnt[j] > 0) ? 1 : 0);
            auto it = cls_id.find(key);
            if (it == cls_id.end()) {
                // two classes per lane in the REST select and in the SPREAD walks beyond 64 (round 6, simon_table.hip: CN2), as in the
                // instantiations without rows and walks
                const int cls_max = (c->rest || c->spread) ? ((c->no_cn2 || c->rs) ? kTableMaxClasses : kTableMaxClassesSpread) : kTableMaxClassesPlain;
                if ((int)shapes.size() == cls_max) { c->table_ok = false; break; }
                it = cls_id.emplace(key, (int)shapes.size()).first;
                ShapeRow sh{};
                sh.cap_c = (double)a_cpu[j]; sh.cap_m = (double)a_mem[j];
                sh.rc_c = 1.0 / sh.cap_c; sh.rc_m = 1.0 / sh.cap_m;           // IEEE, as the device's correctly rounded '/'
                sh.rc100_c = 100.0 * sh.rc_c; sh.rc100_m = 100.0 * sh.rc_m;
                shapes.push_back(sh);
                orig_of.push_back(c->node_class[j]);
                sub_of_class.push_back(std::get<3>(key));
            }
            ncls_t[j] = it->second;
        }
        if (c->table_ok && c->spread && (int)shapes.size() > kTableMaxClasses && sigs.size() > 128) c->table_ok = false;   // (CN2 serves <= 128 signatures)
        if (c->table_ok && c->rest && (int)shapes.size() > kTableMaxClasses && sigs.size() > 128) c->table_ok = false;        // (... and so does the second class per lane of the REST select)
        if (c->table_ok) {
            const int Ct = (int)shapes.size();
            c->n_sigs = (int)sigs.size(); c->Cn_t = Ct;
            if (sigs.empty()) sigs.push_back(SigRow{});
            std::vector<int32_t> raw_t((size_t)Ctc * Ct);
            for (int tc = 0; tc < Ctc; ++tc)
                for (int d = 0; d < Ct; ++d) raw_t[(size_t)tc * Ct + d] = raw32[(size_t)tc_rep[tc] * c->Cn + orig_of[d]];
            for (auto [tab, buf] : {std::make_pair(&c->na_raw, &c->d_t_na), std::make_pair(&c->tt_raw, &c->d_t_tt), std::make_pair(&c->static_add, &c->d_t_add)}) {
                if (tab->empty()) continue;
                std::vector<int32_t> t((size_t)Ctc * Ct);
                for (int tc = 0; tc < Ctc; ++tc)
                    for (int d = 0; d < Ct; ++d) t[(size_t)tc * Ct + d] = (int32_t)(*tab)[(size_t)tc_rep[tc] * c->Cn + orig_of[d]];
                HIP_TRY(c, buf->upload(t, st));
            }
            if (c->has_mask) {
                std::vector<uint64_t> mask_t((size_t)Ctc * words);
                for (int tc = 0; tc < Ctc; ++tc)
                    std::copy(c->static_mask.begin() + (size_t)tc_rep[tc] * words, c->static_mask.begin() + (size_t)(tc_rep[tc] + 1) * words, mask_t.begin() + (size_t)tc * words);
                HIP_TRY(c, c->d_t_mask.upload(mask_t, st));
            }
            // class-major layout: rank of a node among its class, per-class node lists in canonical order, class counts of
            // every prefix of the pool (a scenario = a prefix)
            std::vector<int32_t> rank(N), prefix((size_t)(N + 1) * Ct, 0), cls_off(Ct + 1, 0), cls_list((size_t)N + 64, 0);   // (+ 64: generation 7 reads a whole unit of the last, padded class)
            for (int j = 0; j < N; ++j) {
                const int d = ncls_t[j];
                rank[j] = prefix[(size_t)j * Ct + d];
                for (int e = 0; e < Ct; ++e) prefix[(size_t)(j + 1) * Ct + e] = prefix[(size_t)j * Ct + e] + (e == d);
            }
            for (int d = 0; d < Ct; ++d) cls_off[d + 1] = cls_off[d] + prefix[(size_t)N * Ct + d];
            for (int j = 0; j < N; ++j) cls_list[cls_off[ncls_t[j]] + rank[j]] = j;
            c->h_clsprefix = prefix;
            c->h_ncls_t = ncls_t; c->h_cls_off = cls_off;
            HIP_TRY(c, c->d_sigs.upload(sigs, st));
            if (c->fold) {                                            // which signatures a landing signature excludes from its node
                const int K = (int)sigs.size(), KW = (K + 31) / 32;
                std::vector<uint32_t> fx((size_t)K * KW, 0u);
                for (int L = 0; L < K; ++L)
                    for (int S = 0; S < K; ++S)
                        if (c->fold_x[(size_t)sig_fc[L] * c->fold_FC + sig_fc[S]]) fx[(size_t)L * KW + (S >> 5)] |= 1u << (S & 31);
                HIP_TRY(c, c->d_foldx.upload(fx, st));
            }
            HIP_TRY(c, c->d_shapes.upload(shapes, st));
            HIP_TRY(c, c->d_podsC.upload(rowsC, st));
            HIP_TRY(c, c->d_t_ncls.upload(ncls_t, st));
            HIP_TRY(c, c->d_t_raw.upload(raw_t, st));
            HIP_TRY(c, c->d_rank.upload(rank, st));
            HIP_TRY(c, c->d_clsprefix.upload(prefix, st));
            HIP_TRY(c, c->d_cls_list.upload(cls_list, st));
            HIP_TRY(c, c->d_cls_off.upload(cls_off, st));
            if (c->rest) {
                std::vector<uint32_t> devtot(N, 0), gused((size_t)N * 8, 0);
                for (int j = 0; j < N && c->has_gpu; ++j) {
                    if (c->gpu_cnt[j] > 0) devtot[j] = (uint32_t)((uint64_t)(c->gpu_mem_total[j] / c->gpu_cnt[j]) / c->g_gpu);
                    for (int d = 0; d < 8; ++d) gused[(size_t)j * 8 + d] = (uint32_t)((uint64_t)c->i_gpu_used[(size_t)j * SIMON_MAX_GPU_DEV + d] / c->g_gpu);
                }
                if (gsigs.empty()) gsigs.push_back(make_uint2(0, 0));
                if (xrows.empty()) xrows.push_back(0);
                std::vector<int32_t> gcnt(N, 0);
                if (c->has_gpu) gcnt = c->gpu_cnt;
                HIP_TRY(c, c->d_xrows.upload(xrows, st));
                std::vector<int32_t> zdom(std::max<size_t>(c->zone_keys.size(), 1) * N, -1);
                for (size_t z = 0; z < c->zone_keys.size(); ++z)
                    std::copy(c->topo_dom.begin() + (size_t)c->zone_keys[z] * N, c->topo_dom.begin() + (size_t)(c->zone_keys[z] + 1) * N, zdom.begin() + z * N);
                HIP_TRY(c, c->d_zdom.upload(zdom, st));
                std::vector<uint32_t> xalloc((size_t)N * 8, 0), xused((size_t)N * 8, 0);
                for (int j = 0; j < N && c->xres; ++j) {
                    xalloc[(size_t)j * 8] = (uint32_t)((uint64_t)c->alloc_eph[j] / c->g_eph);
                    xused[(size_t)j * 8] = (uint32_t)((uint64_t)c->i_req_eph[j] / c->g_eph);
                    for (int k = 0; k < c->K; ++k) {
                        xalloc[(size_t)j * 8 + 1 + k] = (uint32_t)c->scalar_alloc[(size_t)k * N + j];
                        xused[(size_t)j * 8 + 1 + k] = (uint32_t)c->i_scalar_req[(size_t)k * N + j];
                    }
                }
                if (xsigs.empty()) xsigs.assign(8, 0);
                HIP_TRY(c, c->d_xsig.upload(xsigs, st)); HIP_TRY(c, c->d_xalloc.upload(xalloc, st)); HIP_TRY(c, c->d_i_xused.upload(xused, st));
                HIP_TRY(c, c->d_gsig.upload(gsigs, st)); HIP_TRY(c, c->d_gpu_cnt.upload(gcnt, st));
                HIP_TRY(c, c->d_gpu_devtot.upload(devtot, st)); HIP_TRY(c, c->d_i_gused.upload(gused, st));
            }
            if (c->gfold && !c->rest) {                              // GPU fold: the pool's devices (the kernel keeps them per position)
                std::vector<uint32_t> devtot(N, 0), gused((size_t)N * 8, 0);
                for (int j = 0; j < N; ++j) {
                    if (c->gpu_cnt[j] > 0) devtot[j] = (uint32_t)((uint64_t)(c->gpu_mem_total[j] / c->gpu_cnt[j]) / c->g_gpu);
                    for (int d = 0; d < 8; ++d) gused[(size_t)j * 8 + d] = (uint32_t)((uint64_t)c->i_gpu_used[(size_t)j * SIMON_MAX_GPU_DEV + d] / c->g_gpu);
                }
                HIP_TRY(c, c->d_gpu_cnt.upload(c->gpu_cnt, st));
                HIP_TRY(c, c->d_gpu_devtot.upload(devtot, st)); HIP_TRY(c, c->d_i_gused.upload(gused, st));
            }
            if (c->spread) {
                std::vector<int32_t> sp_term(std::max(2 * c->Tm, 1), 0);      // ids Tm + t: rows that count the OWNERS of scoring term t (no node set)
                for (int id = 0; id < 2 * c->Tm; ++id)
                    sp_term[id] = c->sp_kind[id] | (c->sp_row[id] << 2) | (c->sp_zslot[id] << 16) | ((c->sp_set_eff[id] + 1) << 19);
                const int nzk = (int)c->sp_zkeys.size();
                std::vector<signed char> zdom((size_t)std::max(nzk, 1) * Ct, 0);
                for (int z = 0; z < nzk; ++z)
                    for (int d = 0; d < Ct; ++d) zdom[(size_t)z * Ct + d] = (signed char)(((sub_of_class[d] >> (5 * z)) & 31) - 1);
                if (sp_ent.empty()) { sp_ent.push_back(0); sp_ent_term.push_back(0); }
                std::vector<int32_t> sp_pairs(sp_ent.size() * 2);            // the term's row travels with the entry (one load in the kernel)
                for (size_t i = 0; i < sp_ent.size(); ++i) { sp_pairs[2 * i] = sp_ent[i]; sp_pairs[2 * i + 1] = sp_term[sp_ent_term[i]]; }
                std::vector<uint64_t> sets = c->node_sets;
                if (sets.empty()) sets.push_back(0);
                HIP_TRY(c, c->d_sp_ent.upload(sp_pairs, st));
                HIP_TRY(c, c->d_cls_zdom.upload(zdom, st));
                { std::vector<double> lg = c->spread_log; if (lg.empty()) lg.push_back(0.0); HIP_TRY(c, c->d_spread_log.upload(lg, st)); }
                HIP_TRY(c, c->d_node_sets.upload(sets, st));
            }
            HIP_TRY(c, hipStreamSynchronize(st));
        }
    }
    HIP_TRY(c, c->d_podsF.upload(rowsF, st));
    HIP_TRY(c, c->d_a_cpu.upload(a_cpu, st));
    HIP_TRY(c, c->d_a_mem.upload(a_mem, st));
    HIP_TRY(c, c->d_i_rq_cpu.upload(rqc, st));
    HIP_TRY(c, c->d_i_rq_mem.upload(rqm, st));
    HIP_TRY(c, c->d_i_nz_cpu.upload(nzc, st));
    HIP_TRY(c, c->d_i_nz_mem.upload(nzm, st));
    HIP_TRY(c, c->d_a_pods.upload(c->alloc_pods, st));
    HIP_TRY(c, c->d_ncls.upload(c->node_class, st));
    HIP_TRY(c, c->d_i_npods.upload(c->i_npods, st));
    HIP_TRY(c, c->d_podsN.upload(rows, st));
    HIP_TRY(c, c->d_raw32.upload(raw32, st));
    if (c->has_mask) HIP_TRY(c, c->d_mask.upload(c->static_mask, st));
    HIP_TRY(c, hipStreamSynchronize(st));  // staging vectors die at scope exit
    return SIMON_OK;
}

int stage(simon_ctx* c) {
    if (!(c->have_nodes && c->have_pods && c->have_tables))
        return fail(c, SIMON_ESTATE, "load nodes, pods and class tables before scenarios");
    if (c->staged) return SIMON_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    // cross-validation of the three inputs
    for (int p = 0; p < c->P; ++p) {
        if (c->p_cls[p] < 0 || c->p_cls[p] >= c->Cp) return fail(c, SIMON_EINVAL, "pod %d: class %d out of range", p, c->p_cls[p]);
        if (c->p_preset[p] >= c->N) return fail(c, SIMON_EINVAL, "pod %d: preset node %d out of range", p, c->p_preset[p]);
    }
    for (int j = 0; j < c->N; ++j)
        if (c->node_class[j] < 0 || c->node_class[j] >= c->Cn) return fail(c, SIMON_EINVAL, "node %d: class out of range", j);
    if (c->has_mask && c->static_mask.size() != (size_t)c->Cp * ((c->N + 63) / 64))
        return fail(c, SIMON_EINVAL, "static_mask size mismatch");
    // derived per staging, never carried over from an earlier pod set: the pool's own GPU arrays, or a pod that asks for GPU memory
    // in a pool without them (every node then fails Open-Gpu-Share, :64-67)
    c->has_gpu = c->has_gpu_nodes;
    if (!c->has_gpu)
        for (int p = 0; p < c->P; ++p) if (c->p_gpu_mem[p] > 0) { c->has_gpu = true; break; }
    choose_variant(c);
    if (c->variant != SIMON_KERNEL_NARROW) c->spread = false;         // (a later precondition sent the problem to the all-feature kernel)
    // prefix sums of allocatable for the occupancy caps (satisfyResourceSetting, apply.go:737-760)
    std::vector<int64_t> pc(c->N + 1, 0), pm(c->N + 1, 0);
    for (int j = 0; j < c->N; ++j) { pc[j + 1] = pc[j] + c->alloc_cpu[j]; pm[j + 1] = pm[j] + c->alloc_mem[j]; }
    c->prefix_vg.assign(c->N + 1, 0);
    for (int j = 0; j < c->N; ++j) {
        int64_t cap = 0;
        if (c->has_local && (c->l_flags[j] & 1))
            for (int q = 0; q < c->l_vg_cnt[j]; ++q) cap += c->l_vg_cap[(size_t)j * SIMON_MAX_VG + q];
        c->prefix_vg[j + 1] = c->prefix_vg[j] + cap;
    }
    HIP_TRY(c, c->d_prefix_cpu.upload(pc, c->stream));
    HIP_TRY(c, c->d_prefix_mem.upload(pm, c->stream));
    HIP_TRY(c, c->d_prefix_vg.upload(c->prefix_vg, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->wide_staged = false;
    int rc = c->variant == SIMON_KERNEL_NARROW ? stage_narrow(c) : wide_stage(c->wide, *c, c->stream, c->err);
    if (rc) return rc;
    c->wide_staged = c->variant != SIMON_KERNEL_NARROW;
    c->staged = true;
    return SIMON_OK;
}

// ---- DefaultPreemption could have acted (ABI v6: simon_set_pod_priorities / simon_fetch_preempt_risk) ----------------------------------
// One wave per scenario walks the pods in scheduling order, 64 steps at a time: risk = some pod failed (SIMON_UNSCHEDULED) while a pod of
// LOWER priority was placed before it -- by the scheduler, by Spec.NodeName, or before the stream (init_min).  PostFilter runs only for a
// failed pod (V/scheduler.go:479) and selectVictimsOnNode only considers pods of lower priority (default_preemption.go:578-592): without
// such a pair no eviction can have happened in the reference and the scenario's result is exact.  HBM-bound: 12 B per pod-step.
__global__ __launch_bounds__(64) void preempt_risk_kernel_impl(const int32_t* __restrict__ place, const int32_t* __restrict__ orders,
                                                               const ScenarioDesc* __restrict__ scen, const int32_t* __restrict__ prio, int P,
                                                               int init_min, unsigned char* __restrict__ risk) {
    const int s = blockIdx.x, lane = threadIdx.x;
    const int32_t* __restrict__ order = orders + (size_t)scen[s].order_id * P;
    const int32_t* __restrict__ pl = place + (size_t)s * P;
    int runmin = init_min;
    bool bad = false;
    for (int i0 = 0; i0 < P; i0 += 64) {
        const int idx = i0 + lane;
        int placed_prio = 0x7fffffff, mine = 0;
        bool failed = false;
        if (idx < P) {
            const int pod = order[idx];
            const int v = pl[pod];
            mine = prio[pod];
            if (v >= 0) placed_prio = mine;
            failed = v == SIMON_UNSCHEDULED;
        }
        int inc = placed_prio;                                            // inclusive prefix minimum over the 64 steps
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(inc, off, 64);
            if (lane >= off) inc = t < inc ? t : inc;
        }
        int exc = __shfl_up(inc, 1, 64);
        if (lane == 0) exc = 0x7fffffff;
        exc = exc < runmin ? exc : runmin;                                // lowest priority among everything placed before this step
        bad = bad || (failed && mine > exc);
        const int last = __shfl(inc, 63, 64);
        runmin = last < runmin ? last : runmin;
    }
    const bool any = __ballot(bad) != 0ull;
    if (lane == 0) risk[s] = any ? 1 : 0;
}

// ---- min-plan reduction (pkg/apply/apply.go:203-259 + satisfyResourceSetting :689-775) --------
__global__ void plan_kernel(const ScenarioDesc* __restrict__ scen, int S, const int32_t* __restrict__ unsched,
                            const int64_t* __restrict__ used_cpu, const int64_t* __restrict__ used_mem,
                            const int64_t* __restrict__ prefix_cpu, const int64_t* __restrict__ prefix_mem, int max_cpu,
                            int max_mem, const int64_t* __restrict__ used_vg, const int64_t* __restrict__ prefix_vg, int max_vg,
                            unsigned long long* __restrict__ out) {
    unsigned long long best = ~0ull;
    for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < S; s += gridDim.x * blockDim.x) {
        if (unsched[s] != 0) continue;
        const int n = scen[s].n_nodes;
        // int(float64(used.MilliValue()) / float64(alloc.MilliValue()) * 100), apply.go:759-760
        const int cpu = (int)((double)used_cpu[s] / (double)prefix_cpu[n] * 100.0);
        const int mem = (int)((double)(used_mem[s] * 1000) / (double)(prefix_mem[n] * 1000) * 100.0);
        if (cpu > max_cpu || mem > max_mem) continue;
        if (used_vg && prefix_vg[n] != 0 &&                      // MaxVG, apply.go:767-771
            (int)((double)used_vg[s] / (double)prefix_vg[n] * 100.0) > max_vg) continue;
        const unsigned long long key = ((unsigned long long)(unsigned)n << 32) | (unsigned)s;
        best = key < best ? key : best;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long o = __shfl_xor(best, off, 64);
        best = o < best ? o : best;
    }
    if ((threadIdx.x & 63) == 0 && best != ~0ull) atomicMin(out, best);
}

}  // namespace

// ============================================================================================
extern "C" {

int simon_hip_version(void) { return SIMON_HIP_ABI_VERSION; }

int simon_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

simon_ctx* simon_ctx_create(int device_id) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device_id < 0 || device_id >= n) return nullptr;
    if (hipSetDevice(device_id) != hipSuccess) return nullptr;
    simon_ctx* c = new (std::nothrow) simon_ctx();
    if (!c) return nullptr;
    c->device = device_id;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) {
        delete c;
        return nullptr;
    }
    { int n_cu = 0; if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, device_id) == hipSuccess && n_cu > 0) c->n_cus = n_cu; }
    // Tuning / experiment knobs: every environment variable is read HERE, once per context; none changes a result.
    if (const char* e = getenv("SIMON_WG")) c->force_T = atoi(e);
    if (const char* e = getenv("SIMON_RCP_DIV")) c->rcp_div = atoi(e) != 0;
    if (const char* e = getenv("SIMON_NARROW_V1")) c->force_v1 = atoi(e) != 0;
    if (const char* e = getenv("SIMON_NO_CACHE")) c->no_cache = atoi(e) != 0;
    if (const char* e = getenv("SIMON_FORCE_WIDE")) c->force_wide = e[0] == '1';
    if (const char* e = getenv("SIMON_TABLE_COARSE")) c->force_coarse = atoi(e) != 0;
    c->no_rest = getenv("SIMON_NO_REST") != nullptr;
    c->no_fold = getenv("SIMON_NO_FOLD") != nullptr;                  // A/B + tests: anti-affinity / ports through the position masks (or the all-feature kernel)
    c->no_sig_twins = getenv("SIMON_TABLE_NO_TWINS") != nullptr;      // A/B: signature ids in order of appearance
    c->no_hard_fold = getenv("SIMON_NO_HARD_FOLD") != nullptr;        // A/B + tests: hard spread constraints always on the all-feature kernel
    c->no_cn2 = getenv("SIMON_NO_CN2") != nullptr;
    c->no_rs = getenv("SIMON_NO_RS") != nullptr;
    c->no_ipa_fold = getenv("SIMON_NO_IPA_FOLD") != nullptr;          // A/B + tests: preferred pod (anti-)affinity always on the all-feature kernel
    c->no_spread = getenv("SIMON_NO_SPREAD") != nullptr;              // A/B + tests: soft spread constraints on the all-feature kernel
    if (const char* e = getenv("SIMON_TEAM")) { const int v = atoi(e); c->team_mode = v == 0 ? 0 : (v == 1 || v == kTeamWaves) ? 1 : -1; }
    if (const char* e = getenv("SIMON_TEAM_MAX_S")) c->team_max_s = atoi(e);
    if (const char* e = getenv("SIMON_LDS_WS")) c->ldsws_mode = atoi(e) ? 1 : 0;
    c->no_gpu_split = getenv("SIMON_TABLE_NO_GPU_SPLIT") != nullptr;
    c->no_class_content = getenv("SIMON_TABLE_NO_CLASS_CONTENT") != nullptr;
    c->no_gpu_fold = getenv("SIMON_NO_GPU_FOLD") != nullptr;
    c->debug_route = getenv("SIMON_DEBUG_ROUTE") != nullptr;
    c->force_table = getenv("SIMON_FORCE_TABLE") != nullptr;         // A/B + tests: keep 257 .. 384 signatures on the score-table kernel   // A/B: node classes not split into with / without devices
#ifdef SIMON_TABLE_PROFILE
    c->table_prof = getenv("SIMON_TABLE_PROF") != nullptr;   // phase profile of simon_table.hip: profiling builds only
#endif
    if (const char* e = getenv("SIMON_CACHE_LDS_PAD")) c->lds_pad = (size_t)atol(e);
    c->wide.knobs.no_lean = getenv("SIMON_WIDE_NO_LEAN") != nullptr;
    c->wide.knobs.no_table = getenv("SIMON_WIDE_NO_TABLE") != nullptr;
    c->wide.knobs.prof = getenv("SIMON_WIDE_PROF") != nullptr;
    c->wide.knobs.no_zmask = getenv("SIMON_WIDE_NO_ZMASK") != nullptr;
    c->wide.knobs.no_ident = getenv("SIMON_WIDE_NO_IDENT") != nullptr;
    c->wide.knobs.no_rows_lds = getenv("SIMON_WIDE_NO_ROWS_LDS") != nullptr;
    if (const char* e = getenv("SIMON_WIDE_NO_CACHE_B")) c->wide.knobs.no_cache_b = *e ? atoi(e) : 3;
    if (const char* e = getenv("SIMON_STATE_BUDGET_MB")) c->wide.knobs.state_budget = (size_t)atoll(e) << 20;
    return c;
}

void simon_ctx_destroy(simon_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    c->wide.release();
    // DevBuf destructors free device memory
    hipStream_t st = c->stream;
    delete c;
    if (st) (void)hipStreamDestroy(st);
}

const char* simon_last_error(simon_ctx* c) { return c ? c->err.c_str() : "null context"; }

int simon_load_nodes(simon_ctx* c, const simon_nodes_soa* nd) {
    if (!c || !nd) return SIMON_EINVAL;
    if (nd->struct_size != sizeof(simon_nodes_soa)) return fail(c, SIMON_EINVAL, "simon_nodes_soa size mismatch");
    const int N = nd->n_nodes;
    if (N <= 0 || !nd->alloc_cpu || !nd->alloc_mem || !nd->alloc_pods) return fail(c, SIMON_EINVAL, "nodes: missing required arrays");
    if (nd->n_scalar < 0 || nd->n_scalar > SIMON_MAX_SCALAR) return fail(c, SIMON_EINVAL, "n_scalar out of range");
    if (nd->n_scalar > 0 && !nd->scalar_alloc) return fail(c, SIMON_EINVAL, "scalar_alloc missing");
    if (nd->n_topo_keys < 0 || (nd->n_topo_keys > 0 && (!nd->topo_dom || !nd->topo_n_dom))) return fail(c, SIMON_EINVAL, "topology arrays missing");
    c->N = N; c->K = nd->n_scalar; c->Kt = nd->n_topo_keys;
    copy_opt(c->alloc_cpu, nd->alloc_cpu, N); copy_opt(c->alloc_mem, nd->alloc_mem, N);
    copy_opt(c->alloc_eph, nd->alloc_eph, N); copy_opt(c->alloc_pods, nd->alloc_pods, N);
    copy_opt(c->i_req_cpu, nd->init_req_cpu, N); copy_opt(c->i_req_mem, nd->init_req_mem, N);
    copy_opt(c->i_req_eph, nd->init_req_eph, N); copy_opt(c->i_nz_cpu, nd->init_nz_cpu, N);
    copy_opt(c->i_nz_mem, nd->init_nz_mem, N); copy_opt(c->i_npods, nd->init_npods, N);
    copy_opt(c->node_class, nd->node_class, N);
    copy_opt(c->scalar_alloc, nd->scalar_alloc, (size_t)c->K * N);
    copy_opt(c->i_scalar_req, nd->init_scalar_req, (size_t)c->K * N);
    c->has_gpu_nodes = c->has_gpu = nd->gpu_cnt != nullptr;
    if (c->has_gpu && !nd->gpu_mem_total) return fail(c, SIMON_EINVAL, "gpu_mem_total missing");
    copy_opt(c->gpu_cnt, nd->gpu_cnt, N); copy_opt(c->gpu_mem_total, nd->gpu_mem_total, N);
    copy_opt(c->i_gpu_used, nd->init_gpu_used, (size_t)N * SIMON_MAX_GPU_DEV);
    for (int j = 0; j < N; ++j)
        if (c->gpu_cnt[j] < 0 || c->gpu_cnt[j] > SIMON_MAX_GPU_DEV) return fail(c, SIMON_ERANGE, "node %d: gpu_cnt %d > %d", j, c->gpu_cnt[j], SIMON_MAX_GPU_DEV);
    c->has_local = nd->local_flags != nullptr;
    if (c->has_local) {
        if (!nd->local_vg_cnt || !nd->local_vg_cap || !nd->local_vg_name || !nd->local_dev_cnt || !nd->local_dev_cap || !nd->local_dev_media)
            return fail(c, SIMON_EINVAL, "Open-Local node arrays missing");
        copy_opt(c->l_flags, nd->local_flags, N); copy_opt(c->l_vg_cnt, nd->local_vg_cnt, N);
        copy_opt(c->l_vg_cap, nd->local_vg_cap, (size_t)N * SIMON_MAX_VG); copy_opt(c->l_vg_req, nd->init_vg_req, (size_t)N * SIMON_MAX_VG);
        copy_opt(c->l_vg_name, nd->local_vg_name, (size_t)N * SIMON_MAX_VG); copy_opt(c->l_dev_cnt, nd->local_dev_cnt, N);
        copy_opt(c->l_dev_cap, nd->local_dev_cap, (size_t)N * SIMON_MAX_LDEV); copy_opt(c->l_dev_media, nd->local_dev_media, N);
        copy_opt(c->l_dev_alloc, nd->init_dev_alloc, N);
        for (int j = 0; j < N; ++j) {
            if (c->l_vg_cnt[j] < 0 || c->l_vg_cnt[j] > SIMON_MAX_VG || c->l_dev_cnt[j] < 0 || c->l_dev_cnt[j] > SIMON_MAX_LDEV)
                return fail(c, SIMON_ERANGE, "node %d: more than %d volume groups or %d devices", j, SIMON_MAX_VG, SIMON_MAX_LDEV);
            for (int v = 0; v < c->l_vg_cnt[j]; ++v)
                if (c->l_vg_cap[(size_t)j * SIMON_MAX_VG + v] <= 0) return fail(c, SIMON_EINVAL, "node %d: volume group %d without capacity", j, v);
            for (int d = 0; d < c->l_dev_cnt[j]; ++d)
                if (c->l_dev_cap[(size_t)j * SIMON_MAX_LDEV + d] <= 0) return fail(c, SIMON_EINVAL, "node %d: device %d without capacity", j, d);
        }
    }
    copy_opt(c->topo_dom, nd->topo_dom, (size_t)c->Kt * N); copy_opt(c->topo_n_dom, nd->topo_n_dom, (size_t)c->Kt);
    for (int k = 0; k < c->Kt; ++k)
        for (int j = 0; j < N; ++j) {
            const int d = c->topo_dom[(size_t)k * N + j];
            if (d < -1 || d >= c->topo_n_dom[k]) return fail(c, SIMON_EINVAL, "topo_dom[%d][%d] out of range", k, j);
        }
    c->have_nodes = true; c->staged = false; c->have_results = false;
    return SIMON_OK;
}

int simon_load_pods(simon_ctx* c, const simon_pods_soa* pd) {
    if (!c || !pd) return SIMON_EINVAL;
    if (pd->struct_size != sizeof(simon_pods_soa)) return fail(c, SIMON_EINVAL, "simon_pods_soa size mismatch");
    if (!c->have_nodes) return fail(c, SIMON_ESTATE, "load nodes before pods");
    const int P = pd->n_pods;
    if (P < 0 || (P > 0 && (!pd->req_cpu || !pd->req_mem))) return fail(c, SIMON_EINVAL, "pods: missing required arrays");
    c->P = P;
    copy_opt(c->p_req_cpu, pd->req_cpu, P); copy_opt(c->p_req_mem, pd->req_mem, P); copy_opt(c->p_req_eph, pd->req_eph, P);
    if (pd->nz_cpu) copy_opt(c->p_nz_cpu, pd->nz_cpu, P); else c->p_nz_cpu = c->p_req_cpu;
    if (pd->nz_mem) copy_opt(c->p_nz_mem, pd->nz_mem, P); else c->p_nz_mem = c->p_req_mem;
    copy_opt(c->p_scalar, pd->scalar_req, (size_t)c->K * P);
    copy_opt(c->p_cls, pd->pod_class, P);
    copy_opt(c->p_preset, pd->preset_node, P, (int32_t)-1);
    copy_opt(c->p_gate, pd->gate_node, P, (int32_t)-1);
    copy_opt(c->p_gpu_mem, pd->gpu_mem, P); copy_opt(c->p_gpu_cnt, pd->gpu_cnt, P);
    copy_opt(c->p_pin, pd->pin_node, P, (int32_t)-1);
    copy_opt(c->p_gpu_index, pd->gpu_index, P);
    c->has_gpu_index = false;
    for (int p = 0; p < P; ++p) {
        const uint32_t w = c->p_gpu_index[p];
        if (!w) continue;
        c->has_gpu_index = true;
        bool ended = false;                            // nibbles: 1 + device id, 0 ends the list; ids 0 .. SIMON_MAX_GPU_DEV - 1
        for (int i = 0; i < 8; ++i) {
            const uint32_t nib = (w >> (4 * i)) & 15u;
            if (!nib) ended = true;
            else if (ended || nib > SIMON_MAX_GPU_DEV) return fail(c, SIMON_EINVAL, "pod %d: gpu_index 0x%x is not a packed device-id list", p, w);
        }
    }
    c->has_pin = false;
    for (int p = 0; p < P; ++p) {
        if (c->p_pin[p] >= c->N) return fail(c, SIMON_EINVAL, "pod %d: pin_node %d out of range", p, c->p_pin[p]);
        if (c->p_pin[p] >= 0 && c->p_preset[p] >= 0) return fail(c, SIMON_EINVAL, "pod %d: both preset_node and pin_node", p);
        c->has_pin = c->has_pin || c->p_pin[p] >= 0;
    }
    c->p_entries.clear(); c->p_priority.clear(); c->init_min_priority = 0x7fffffff;   // (ABI v6 additions: set after the pods)
    c->have_pods = true; c->staged = false; c->have_results = false;
    return SIMON_OK;
}

int simon_set_scalar_entries(simon_ctx* c, const uint8_t* entries) {
    if (!c) return SIMON_EINVAL;
    if (!c->have_pods) return fail(c, SIMON_ESTATE, "set_scalar_entries: load pods first");
    if (entries) {
        c->p_entries.assign(entries, entries + c->P);
        for (int p = 0; p < c->P; ++p) {
            if (c->p_entries[p] & 0x70u & ~((1u << SIMON_MAX_SCALAR) - 1u)) return fail(c, SIMON_EINVAL, "pod %d: scalar_entries 0x%x names a resource beyond SIMON_MAX_SCALAR", p, c->p_entries[p]);
            for (int k = 0; k < c->K; ++k)                    // a non-zero quantity IS an entry
                if (c->p_scalar[(size_t)k * c->P + p] != 0) c->p_entries[p] |= (uint8_t)(1u << k);
        }
    } else c->p_entries.clear();
    c->staged = false; c->wide_staged = false; c->have_results = false;
    return SIMON_OK;
}

int simon_set_pod_priorities(simon_ctx* c, const int32_t* priority, int32_t init_min_priority) {
    if (!c) return SIMON_EINVAL;
    if (!c->have_pods) return fail(c, SIMON_ESTATE, "set_pod_priorities: load pods first");
    if (priority) c->p_priority.assign(priority, priority + c->P); else c->p_priority.clear();
    c->init_min_priority = priority ? init_min_priority : 0x7fffffff;
    c->prio_staged = false; c->have_risk = false;
    return SIMON_OK;
}

int simon_load_class_tables(simon_ctx* c, const simon_class_tables* tb) {
    if (!c || !tb) return SIMON_EINVAL;
    if (tb->struct_size != sizeof(simon_class_tables)) return fail(c, SIMON_EINVAL, "simon_class_tables size mismatch");
    if (!c->have_nodes) return fail(c, SIMON_ESTATE, "load nodes before class tables");
    if (tb->n_pod_classes <= 0 || tb->n_node_classes <= 0 || !tb->simon_raw) return fail(c, SIMON_EINVAL, "class tables: bad sizes");
    c->Cp = tb->n_pod_classes; c->Cn = tb->n_node_classes;
    const size_t words = (size_t)(c->N + 63) / 64;
    c->has_mask = tb->static_mask != nullptr;
    copy_opt(c->static_mask, tb->static_mask, c->has_mask ? (size_t)c->Cp * words : 0);
    copy_opt(c->static_reason, tb->static_reason, tb->static_reason ? (size_t)c->Cp * c->N : 0);
    copy_opt(c->simon_raw, tb->simon_raw, (size_t)c->Cp * c->Cn);
    copy_opt(c->const_score, tb->const_score, (size_t)c->Cp);
    const size_t cells = (size_t)c->Cp * c->Cn;
    c->has_na = tb->node_affinity_raw != nullptr; c->has_tt = tb->taint_prefer_raw != nullptr; c->has_add = tb->static_add != nullptr;
    copy_opt(c->na_raw, tb->node_affinity_raw, c->has_na ? cells : 0);
    copy_opt(c->tt_raw, tb->taint_prefer_raw, c->has_tt ? cells : 0);
    copy_opt(c->static_add, tb->static_add, c->has_add ? cells : 0);
    for (int64_t x : c->na_raw) if (x < 0 || x >= (1ll << 40)) return fail(c, SIMON_ERANGE, "node_affinity_raw outside [0, 2^40)");
    for (int64_t x : c->tt_raw) if (x < 0 || x >= (1ll << 40)) return fail(c, SIMON_ERANGE, "taint_prefer_raw outside [0, 2^40)");
    for (int64_t x : c->static_add) if (x < 0 || x >= (1ll << 30)) return fail(c, SIMON_ERANGE, "static_add outside [0, 2^30)");
    c->l_spec_of.clear(); c->l_specs.clear();
    if (tb->local_spec_of) {
        if (!c->has_local) return fail(c, SIMON_ESTATE, "local_spec_of given but the nodes carry no Open-Local arrays");
        if (tb->n_local_specs < 0 || (tb->n_local_specs > 0 && !tb->local_specs)) return fail(c, SIMON_EINVAL, "local_specs missing");
        copy_opt(c->l_spec_of, tb->local_spec_of, (size_t)c->Cp);
        c->l_specs.assign(tb->local_specs, tb->local_specs + tb->n_local_specs);
        for (int32_t x : c->l_spec_of) if (x < -1 || x >= tb->n_local_specs) return fail(c, SIMON_EINVAL, "local_spec_of out of range");
        for (const simon_local_spec& sp : c->l_specs) {
            if (sp.n_lvm < 0 || sp.n_lvm > SIMON_MAX_LVOL || sp.n_ssd < 0 || sp.n_ssd > SIMON_MAX_LVOL || sp.n_hdd < 0 || sp.n_hdd > SIMON_MAX_LVOL)
                return fail(c, SIMON_ERANGE, "more than %d Open-Local volumes of one kind", SIMON_MAX_LVOL);
            for (int k = 0; k < sp.n_lvm; ++k) if (sp.lvm_size[k] <= 0) return fail(c, SIMON_EINVAL, "Open-Local volume size <= 0");
            for (int k = 0; k < sp.n_ssd; ++k) if (sp.ssd_size[k] <= 0 || (k && sp.ssd_size[k] < sp.ssd_size[k - 1])) return fail(c, SIMON_EINVAL, "ssd_size must be positive and ascending");
            for (int k = 0; k < sp.n_hdd; ++k) if (sp.hdd_size[k] <= 0 || (k && sp.hdd_size[k] < sp.hdd_size[k - 1])) return fail(c, SIMON_EINVAL, "hdd_size must be positive and ascending");
        }
    }
    c->Tm = tb->n_terms;
    if (c->Tm < 0) return fail(c, SIMON_EINVAL, "n_terms < 0");
    c->term_key.clear(); c->term_set.clear(); c->node_sets.clear(); c->R = 0;
    c->match_off.clear(); c->match_idx.clear(); c->anti_off.clear(); c->anti_idx.clear(); c->aff_off.clear(); c->aff_idx.clear(); c->port_off.clear(); c->port_idx.clear();
    c->class_flags.clear(); c->pref_off.clear(); c->pref_idx.clear(); c->pref_w.clear(); c->own_off.clear(); c->own_idx.clear();
    c->own_w.clear(); c->sh_off.clear(); c->sh_idx.clear(); c->sh_skew.clear(); c->sh_self.clear(); c->sh_set.clear();
    c->ss_off.clear(); c->ss_idx.clear(); c->ss_skew.clear(); c->topo_is_hostname.clear(); c->spread_log.clear();
    c->has_ipa_score = false;
    if (c->Tm > 0) {
        if (!tb->term_topo_key) return fail(c, SIMON_EINVAL, "term_topo_key missing");
        copy_opt(c->term_key, tb->term_topo_key, (size_t)c->Tm);
        for (int t = 0; t < c->Tm; ++t)
            if (c->term_key[t] < 0 || c->term_key[t] >= c->Kt) return fail(c, SIMON_EINVAL, "term %d: topology key out of range", t);
        c->R = tb->n_node_sets;
        if (c->R < 0 || (c->R > 0 && !tb->node_sets)) return fail(c, SIMON_EINVAL, "node_sets missing");
        copy_opt(c->node_sets, tb->node_sets, (size_t)c->R * words);
        copy_opt(c->term_set, tb->term_node_set, (size_t)c->Tm, (int32_t)-1);
        for (int32_t r : c->term_set) if (r < -1 || r >= c->R) return fail(c, SIMON_EINVAL, "term_node_set out of range");
        // one CSR list per role; idx entries are term ids
        struct Csr { const char* name; const int32_t* off; const int32_t* idx; std::vector<int32_t>* h_off; std::vector<int32_t>* h_idx; int cap; };
        Csr lists[] = {
            {"match", tb->match_off, tb->match_idx, &c->match_off, &c->match_idx, 1 << 20},
            {"anti", tb->anti_off, tb->anti_idx, &c->anti_off, &c->anti_idx, SIMON_MAX_TERMS_PER_CLASS * 4},
            {"aff", tb->aff_off, tb->aff_idx, &c->aff_off, &c->aff_idx, SIMON_MAX_TERMS_PER_CLASS * 4},
            {"port", tb->port_off, tb->port_idx, &c->port_off, &c->port_idx, 1 << 16},
            {"pref", tb->pref_off, tb->pref_idx, &c->pref_off, &c->pref_idx, SIMON_MAX_TERMS_PER_CLASS * 4},
            {"own", tb->own_off, tb->own_idx, &c->own_off, &c->own_idx, SIMON_MAX_TERMS_PER_CLASS * 8},
            {"spread_hard", tb->spread_hard_off, tb->spread_hard_idx, &c->sh_off, &c->sh_idx, SIMON_MAX_SPREAD},
            {"spread_soft", tb->spread_soft_off, tb->spread_soft_idx, &c->ss_off, &c->ss_idx, SIMON_MAX_SPREAD},
        };
        for (Csr& l : lists) {
            if (!l.off) continue;
            copy_opt(*l.h_off, l.off, (size_t)c->Cp + 1);
            if ((*l.h_off)[0] != 0) return fail(c, SIMON_EINVAL, "%s_off[0] != 0", l.name);
            for (int k = 0; k < c->Cp; ++k) {
                const int len = (*l.h_off)[k + 1] - (*l.h_off)[k];
                if (len < 0 || len > l.cap) return fail(c, SIMON_EINVAL, "class %d: %s list has %d entries (limit %d)", k, l.name, len, l.cap);
            }
            const size_t cnt = (size_t)(*l.h_off)[c->Cp];
            if (cnt > 0 && !l.idx) return fail(c, SIMON_EINVAL, "%s_idx missing", l.name);
            copy_opt(*l.h_idx, l.idx, cnt);
            for (int32_t t : *l.h_idx) if (t < 0 || t >= c->Tm) return fail(c, SIMON_EINVAL, "%s_idx out of range", l.name);
        }
        if (c->match_off.empty()) c->match_off.assign(c->Cp + 1, 0);
        if (c->anti_off.empty()) c->anti_off.assign(c->Cp + 1, 0);
        copy_opt(c->class_flags, tb->class_flags, tb->class_flags ? (size_t)c->Cp : 0);
        if (!c->pref_idx.empty()) { if (!tb->pref_w) return fail(c, SIMON_EINVAL, "pref_w missing"); copy_opt(c->pref_w, tb->pref_w, c->pref_idx.size()); }
        if (!c->own_idx.empty()) { if (!tb->own_w) return fail(c, SIMON_EINVAL, "own_w missing"); copy_opt(c->own_w, tb->own_w, c->own_idx.size()); }
        c->has_ipa_score = !c->pref_idx.empty() || !c->own_idx.empty();
        // counters are int32 on the device: bound the weighted sums by P * sum |w|
        long long wsum = 0;
        for (int32_t x : c->pref_w) wsum = std::max<long long>(wsum, std::llabs((long long)x));
        long long osum = 0;
        for (int k = 0; k < c->Cp && !c->own_off.empty(); ++k) {
            long long sk = 0;
            for (int e = c->own_off[k]; e < c->own_off[k + 1]; ++e) sk += std::llabs((long long)c->own_w[e]);
            osum = std::max(osum, sk);
        }
        if (osum * std::max(c->P, 1) >= (1ll << 31)) return fail(c, SIMON_ERANGE, "own_w x pods overflows the int32 weight counters");
        (void)wsum;
        if (!c->sh_idx.empty()) {
            if (!tb->spread_hard_skew || !tb->spread_hard_self) return fail(c, SIMON_EINVAL, "spread_hard_skew/self missing");
            copy_opt(c->sh_skew, tb->spread_hard_skew, c->sh_idx.size());
            copy_opt(c->sh_self, tb->spread_hard_self, c->sh_idx.size());
            copy_opt(c->sh_set, tb->spread_hard_set, c->sh_idx.size(), (int32_t)-1);
            for (int32_t r : c->sh_set) if (r < -1 || r >= c->R) return fail(c, SIMON_EINVAL, "spread_hard_set out of range");
        }
        if (!c->ss_idx.empty()) {
            if (!tb->spread_soft_skew || !tb->spread_log) return fail(c, SIMON_EINVAL, "spread_soft_skew / spread_log missing");
            copy_opt(c->ss_skew, tb->spread_soft_skew, c->ss_idx.size());
            copy_opt(c->spread_log, tb->spread_log, (size_t)c->N + 1);
            if ((long long)c->P * SIMON_MAX_SPREAD >= (1ll << 31) - 2) return fail(c, SIMON_ERANGE, "too many pods for the spread stamps");
        }
        copy_opt(c->topo_is_hostname, tb->topo_is_hostname, tb->topo_is_hostname ? (size_t)c->Kt : 0);
    } else {
        c->anti_off.assign(c->Cp + 1, 0); c->match_off.assign(c->Cp + 1, 0);
    }
    c->have_tables = true; c->staged = false; c->have_results = false;
    return SIMON_OK;
}

int simon_load_scenarios(simon_ctx* c, const simon_scenario* scen, int32_t S, const int32_t* orders, int32_t n_orders) {
    if (!c || !scen || S <= 0 || !orders || n_orders <= 0) return c ? fail(c, SIMON_EINVAL, "load_scenarios: bad arguments") : SIMON_EINVAL;
    int rc = stage(c);
    if (rc) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    const int P = c->P;
    c->scen.resize(S);
    int max_n = 0;
    for (int s = 0; s < S; ++s) {
        if (scen[s].n_nodes < 0 || scen[s].n_nodes > c->N) return fail(c, SIMON_EINVAL, "scenario %d: n_nodes %d outside [0,%d]", s, scen[s].n_nodes, c->N);
        if (scen[s].order_id < 0 || scen[s].order_id >= n_orders) return fail(c, SIMON_EINVAL, "scenario %d: order_id out of range", s);
        c->scen[s] = ScenarioDesc{scen[s].n_nodes, scen[s].order_id};
        max_n = std::max(max_n, scen[s].n_nodes);
    }
    // every order must be a sequence of valid pod ids; preset targets must exist in each scenario
    for (size_t i = 0; i < (size_t)n_orders * P; ++i)
        if (orders[i] < 0 || orders[i] >= P) return fail(c, SIMON_EINVAL, "orders[%zu] = %d is not a pod id", i, orders[i]);
    int min_n = c->N;
    for (int s = 0; s < S; ++s) min_n = std::min(min_n, scen[s].n_nodes);
    for (int p = 0; p < P; ++p)
        if (c->p_preset[p] >= 0 && c->p_preset[p] >= min_n && c->p_gate[p] < c->p_preset[p])
            return fail(c, SIMON_EINVAL, "pod %d is preset to node %d which a %d-node scenario lacks (gate it)", p, c->p_preset[p], min_n);
    // LPT launch order: biggest scenarios first (cost ~ ceil(n / T) per pod)
    std::vector<int32_t> perm(S);
    std::iota(perm.begin(), perm.end(), 0);
    std::stable_sort(perm.begin(), perm.end(), [&](int a, int b) { return scen[a].n_nodes > scen[b].n_nodes; });
    HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
    HIP_TRY(c, c->d_scen.upload(c->scen, c->stream));
    HIP_TRY(c, c->d_perm.upload(perm, c->stream));
    HIP_TRY(c, c->d_orders.ensure((size_t)n_orders * P));
    HIP_TRY(c, hipMemcpyAsync(c->d_orders.p, orders, (size_t)n_orders * P * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, c->d_unsched.ensure(S));
    HIP_TRY(c, c->d_used_cpu.ensure(S));
    HIP_TRY(c, c->d_used_mem.ensure(S));
    HIP_TRY(c, c->d_used_vg.ensure(S));
    HIP_TRY(c, c->d_plan.ensure(1));
    c->h_perm = perm;
    c->scen_ni.assign(S, 0);
    c->h_orders.assign(orders, orders + (size_t)n_orders * P);
    c->table_perm_ok = false;
    if (c->variant == SIMON_KERNEL_NARROW && c->table_ok) {
        // placements are recorded by scheduling step and gathered back to pod ids through the inverse orders,
        // which exist only when every order is a permutation of [0, P)
        std::vector<int32_t> inv((size_t)n_orders * P, -1);
        bool is_perm = true;
        for (int o = 0; o < n_orders && is_perm; ++o)
            for (int i = 0; i < P; ++i) {
                int32_t& slot = inv[(size_t)o * P + orders[(size_t)o * P + i]];
                if (slot >= 0) { is_perm = false; break; }
                slot = i;
            }
        // padded class-major size of every scenario; workspace slice of every workgroup (launch order = perm)
        const int Ct = c->Cn_t;
        std::vector<int> ni16(S), ni64(S);
        int top16 = 16, top64 = 64;
        for (int s = 0; s < S; ++s) {
            int a = 0, b = 0;
            for (int d = 0; d < Ct; ++d) {
                const int cnt = c->h_clsprefix[(size_t)scen[s].n_nodes * Ct + d];
                a += (cnt + 15) & ~15; b += (cnt + 63) & ~63;
            }
            ni16[s] = std::max(a, 16); ni64[s] = std::max(b, 64);
            top16 = std::max(top16, ni16[s]); top64 = std::max(top64, ni64[s]);
        }
        if (is_perm) {
            // One-level or two-level summary (simon_table.hip: tcarve)?  The LDS of a workgroup decides how many scenario waves a CU
            // holds (allocation granularity 1 280 B, 160 KB, at most 32 single-wave workgroups: profiles/micro/occupancy_probe.hip),
            // and a batch that offers more runs in rounds.  The two-level form needs a quarter of the LDS but costs an 8-byte load and
            // ~15 VALU per signature and cycle more (+19 % per wave, measured).  Cost model fitted to profiles/README.md (config 3:
            // 256 scenarios 7.7 ms, 4 096 14.0 ms, 8 192 27.5 ms one-level / 32.3 ms two-level; 100 signatures 39.5 / 32.1 ms):
            // a round of w waves per CU takes 1 + 0.055 (w - 1) units up to 16 waves and 0.11 per wave beyond.
            auto fit = [](size_t lds) { const size_t g = (lds + 1279) / 1280 * 1280; return g ? (int)std::min<size_t>(32, kTableLdsPerCU / g) : 32; };
            const int nzk = c->spread ? ((int)c->sp_zkeys.size() | ((c->ipa_fold || c->hard_fold) ? 0x100 : 0) | (Ct > kTableMaxClasses ? 0x400 : 0)) : -1;   // (| 0x100: the second score table of spread_select; | 0x400: CN2's larger one)
            const size_t lds16 = table_lds_bytes(c->n_sigs, top16, Ct, false, false) + c->lds_pad, lds64 = table_lds_bytes(c->n_sigs, top64, Ct, true, c->rest, nzk) + c->lds_pad;
            const bool fine_ok = max_n <= kTableMaxNodes && top16 <= kTableMaxPadded && lds16 <= 64 * 1024, coarse_ok = top64 <= kTableMaxPaddedCoarse && lds64 <= kTableLdsMaxWG && Ct <= 128;   // (129 .. 256 classes: one-level only, simon_table_cls4.hip)
            const int per_cu = (S + c->n_cus - 1) / std::max(c->n_cus, 1);
            auto cost = [&](int fits, double factor) {
                auto round_cost = [](int w) { return 1.0 + 0.055 * (std::min(w, 16) - 1) + 0.11 * std::max(w - 16, 0); };
                double t = 0;
                for (int left = per_cu; left > 0; left -= fits) t += round_cost(std::min(left, fits));
                return t * factor;
            };
            bool coarse = coarse_ok && (!fine_ok || cost(std::max(fit(lds64), 1), 1.2) < 0.95 * cost(std::max(fit(lds16), 1), 1.0));
            if (c->force_coarse >= 0) coarse = c->force_coarse ? coarse_ok : !fine_ok && coarse_ok;
            if (c->rest) coarse = coarse_ok;                        // the REST path is built on the two-level layout
            if (c->n_sigs > 128) coarse = coarse_ok;                // ... and so are the signature groups beyond 128 (simon_table.hip: MANY)
            if (c->spread) coarse = coarse_ok;                      // ... and the SPREAD path (generation 7)
            if (c->fold || c->gfold) coarse = coarse_ok;            // ... and the folded exclusions / GPU share (carried by the two-level instantiations)
            c->table_coarse = coarse;
            for (int s = 0; s < S; ++s) c->scen_ni[s] = coarse ? ni64[s] : ni16[s];
            const int ni_top = coarse ? top64 : top16;
            c->table_ni_top = ni_top;
            std::vector<unsigned long long> ws_off(S);
            size_t off = 0;
            for (int b = 0; b < S; ++b) { ws_off[b] = off; off += table_ws_bytes(c->n_sigs, c->scen_ni[perm[b]], c->nzeq, coarse, Ct, c->rest ? c->rest_M : 0, c->rest ? (int)c->zone_keys.size() : 0, c->spread ? c->sp_TH : 0, c->spread ? c->sp_TZ : 0)
                                                                    + (c->gfold ? (((size_t)c->scen_ni[perm[b]] * 40 + 127) & ~(size_t)127) : 0); }   // + GPU fold: devices by position
            c->ws_total = off;
            HIP_TRY(c, c->d_ws_off.upload(ws_off, c->stream));
            HIP_TRY(c, c->d_inv_orders.upload(inv, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            // more than 128 signatures: two-level layout without the REST path (simon_table.hip: MANY; since round 4 also under generation 7's
            // walks); else generation 2 / all-feature kernel
            c->table_perm_ok = (!c->rest || coarse) && (c->n_sigs <= 128 || coarse) && (!c->spread || coarse) && (!c->fold || coarse) && (!c->gfold || coarse);
        }
    }
    HIP_TRY(c, hipEventRecord(c->ev1, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));  // caller buffers may die after return
    float ms = 0;
    (void)hipEventElapsedTime(&ms, c->ev0, c->ev1);
    c->stats.h2d_ms = ms;
    c->S = S; c->n_orders = n_orders; c->max_n = max_n;
    c->has_ranks = false;
    c->have_results = false;
    return SIMON_OK;
}

int simon_set_node_ranks(simon_ctx* c, const int32_t* rank) {
    if (!c) return SIMON_EINVAL;
    if (!c->staged || c->S <= 0) return fail(c, SIMON_ESTATE, "set_node_ranks: load scenarios first");
    if (!rank) { c->has_ranks = false; return SIMON_OK; }
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t S = c->S, N = c->N;
    std::vector<int32_t> inv(S * N, 0);
    for (size_t s = 0; s < S; ++s) {
        const int n = c->scen[s].n_nodes;
        std::vector<char> seen(n, 0);
        for (int j = 0; j < n; ++j) {
            const int r = rank[s * N + j];
            if (r < 0 || r >= n || seen[r]) return fail(c, SIMON_EINVAL, "set_node_ranks: scenario %d: not a permutation of 0..%d", (int)s, n - 1);
            seen[r] = 1;
            inv[s * N + r] = j;
        }
    }
    std::vector<int32_t> rk(rank, rank + S * N);
    HIP_TRY(c, c->d_node_rank.upload(rk, c->stream));
    HIP_TRY(c, c->d_node_inv.upload(inv, c->stream));
    // the score-table kernel's view of a scenario's own node order: per-class node lists in rank order (class segments at the
    // pool's class offsets; a scenario fills the first clsprefix[n][d] entries of segment d) and a node's index inside its class
    c->table_ranks_ok = false;
    if (c->variant == SIMON_KERNEL_NARROW && c->table_ok && c->table_perm_ok) {
        const int Ct = c->Cn_t;
        std::vector<int32_t> ids(S * N, 0), pos(S * N, 0), fill(Ct);
        for (size_t s = 0; s < S; ++s) {
            const int n = c->scen[s].n_nodes;
            std::fill(fill.begin(), fill.end(), 0);
            for (int r = 0; r < n; ++r) {                       // nodes in rank order
                const int j = inv[s * N + r], d = c->h_ncls_t[j];
                pos[s * N + j] = fill[d];
                ids[s * N + c->h_cls_off[d] + fill[d]++] = j;
            }
        }
        HIP_TRY(c, c->d_rk_ids.upload(ids, c->stream));
        HIP_TRY(c, c->d_rk_pos.upload(pos, c->stream));
        c->table_ranks_ok = true;
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->has_ranks = true;
    c->have_results = false;
    return SIMON_OK;
}

int simon_run_loaded(simon_ctx* c, int32_t want_placement) {
    if (!c) return SIMON_EINVAL;
    if (!c->staged || c->S <= 0) return fail(c, SIMON_ESTATE, "run_loaded: no scenarios loaded");
    HIP_TRY(c, hipSetDevice(c->device));
    const int S = c->S, P = c->P;
    // bit 1: also record the devices Reserve books for every placed GPU pod (only problems with GPU requests have any)
    const bool want_slices = (want_placement & SIMON_WANT_GPU_SLICES) != 0 && c->has_gpu;
    want_placement &= SIMON_WANT_PLACEMENT;
    const bool want_risk = !c->p_priority.empty();               // pods of unequal priority: the flags come from the placement matrix
    if (want_risk) want_placement = SIMON_WANT_PLACEMENT;
    c->have_slices = false; c->have_risk = false;
    if (want_slices) {
        HIP_TRY(c, c->d_gpu_slices.ensure((size_t)S * P));
        HIP_TRY(c, hipMemsetAsync(c->d_gpu_slices.p, 0, (size_t)S * P * 8, c->stream));
    }
    if (want_placement) HIP_TRY(c, c->d_place.ensure((size_t)S * P));
    int T = 0, slots = 0, variant_used = c->variant;
    bool table_used = false;
    size_t lds = 0;
    // per-scenario node ranks: the score-table kernel (own per-class lists) or the all-feature kernel
    bool run_wide = c->variant != SIMON_KERNEL_NARROW || (c->has_ranks && !c->table_ranks_ok);
    if (!run_wide) {
        // workgroup shape: T = 256 (4 waves) with up to 8 node slots per lane covers 2048 nodes;
        // larger pools widen the workgroup.  SIMON_WG overrides (tuning knob).
        T = c->force_T ? c->force_T : (c->max_n <= 2048 ? 256 : c->max_n <= 4096 ? 512 : 1024);
        slots = (std::max(c->max_n, 1) + T - 1) / T;
        const bool too_big = slots > 8;   // pool too large for register residency even at T = 1024: the all-feature kernel takes it
        if (slots == 5) slots = 6;
        if (slots == 7) slots = 8;
        // the score-table kernel: one workgroup (one wave) per scenario, ONE launch, scenarios in LPT order
        const int ni_top = c->table_ni_top;
        // team mode: a small batch of a problem with soft spread constraints gets kTeamWaves waves per scenario (the walks of
        // spread_select and the prologue are split; LDS: + one score table, the canonical indices and the exchange slots)
        const int team_max = c->team_max_s >= 0 ? c->team_max_s : 2 * c->n_cus;
        int team = (c->spread && c->table_coarse && (!c->rest || c->rs) && c->team_mode != 0 && (c->team_mode > 0 || S <= team_max)) ? kTeamWaves : 1;
        auto lds_for = [&](int tm) -> size_t {
            return c->table_ok ? table_lds_bytes(c->n_sigs, ni_top, c->Cn_t, c->table_coarse, c->rest, c->spread ? ((int)c->sp_zkeys.size() | ((c->ipa_fold || c->hard_fold) ? 0x100 : 0) | (tm > 1 ? 0x200 : 0) | (c->Cn_t > kTableMaxClasses ? 0x400 : 0)) : -1) + c->lds_pad : 0;
        };
        if (team > 1 && lds_for(team) > kTableLdsMaxWG) team = 1;           // (its extra table does not fit: the single-wave shape still may)
        size_t table_lds = lds_for(team);
        // Generation 4 with the scenario's workspace in LDS (round 5; simon_table.hip: LDSWS): a batch of at most one scenario per CU -- what
        // a real Applier.Run offers -- of a problem whose byte table + node state fit the CU's LDS next to the summaries.  The one memory
        // round trip of a scheduling cycle becomes an LDS access.
        // The LDS-resident homes (LDSWS, LDSX) trade occupancy for latency: a workgroup asks for up to 159 KB, so a CU holds ONE.  They are
        // the default only where that costs nothing whatever the dispatcher does: a batch of at most one scenario per CU (what a real
        // Applier.Run offers).  Round 5 also took batches of a few workgroups per CU when ceil(S / CUs) of them fit the CU's LDS together --
        // which assumed the dispatcher places exactly that many per CU, and left simon_load_scenarios' fine / coarse cost model (lds16 /
        // lds64, fitted before this LDS growth) describing another launch; those batches keep the HBM homes now (SIMON_LDS_WS=1: whenever
        // one scenario fits).
        auto lds_home = [&](size_t need) -> bool {
            if (need > kTableLdsMaxWG || c->ldsws_mode == 0) return false;
            return c->ldsws_mode > 0 || S <= std::max(c->n_cus, 1);
        };
        bool lds_ws = false;
        if (c->table_ok && !c->table_coarse && !c->rest && !c->spread && !c->fold && !c->gfold && c->n_sigs <= 128 && c->Cn_t <= 128 && team == 1) {
            size_t ws_max = 0;
            for (int s2 = 0; s2 < S; ++s2) ws_max = std::max(ws_max, table_ws_bytes(c->n_sigs, c->scen_ni[s2], c->nzeq, false, c->Cn_t, 0, 0, 0, 0));
            const size_t need = ((table_lds - c->lds_pad + 127) & ~(size_t)127) + ws_max + c->lds_pad;   // (the kernel puts the workspace at the next 128-byte boundary behind tcarve's total)
            if (lds_home(need)) { lds_ws = true; table_lds = need; }
        }
        // Generation 6 with its mask rows, row totals and canonical indices in LDS (round 5; simon_table.hip: LDSX), under the same rule
        bool lds_x = false;
        if (c->table_ok && c->table_coarse && c->rest && !c->spread && team == 1 && c->Cn_t <= kTableMaxClasses && c->n_sigs <= 128) {   // (65 .. 128 node classes: the rows stay in HBM, simon_table_rest2.hip)
            const size_t need = ((table_lds - c->lds_pad + 127) & ~(size_t)127) + table_ldsx_bytes(ni_top, c->rest_M) + c->lds_pad;
            if (lds_home(need)) { lds_x = true; table_lds = need; }
        }
        bool use_table = c->table_ok && c->table_perm_ok && (c->Cn_t <= 128 || (!c->table_coarse && !c->fold && !c->gfold && c->n_sigs <= 128)) && !c->no_cache && !c->force_v1 && c->max_n <= (c->table_coarse ? kTableMaxNodesCoarse : kTableMaxNodes) &&
                               ni_top <= (c->table_coarse ? kTableMaxPaddedCoarse : kTableMaxPadded) && table_lds <= ((c->table_coarse || lds_ws) ? kTableLdsMaxWG : (size_t)64 * 1024);
        // pinned pods (pin_node) are known to the score-table kernel and the all-feature kernel only
        if (c->debug_route)
            fprintf(stderr, "[route] variant %d rest %d spread %d fold %d gfold %d table_ok %d perm_ok %d coarse %d n_sigs %d Cn_t %d ni_top %d lds %zu max_n %d M %d NZ %d TH %d TZ %d\n", c->variant, (int)c->rest,
                    (int)c->spread, (int)c->fold, (int)c->gfold, (int)c->table_ok, (int)c->table_perm_ok, (int)c->table_coarse, c->n_sigs, c->Cn_t, ni_top, table_lds, c->max_n,
                    c->rest_M, (int)c->zone_keys.size(), c->sp_TH, c->sp_TZ);
        const bool needs_table_or_wide = c->has_pin || too_big || c->rest || c->spread || c->fold || c->gfold || !c->raw_fits_lds || c->has_ranks || c->has_static;
        // Beyond 256 signatures generation 2 (register-resident state, every node re-evaluated per cycle: its time does not depend on
        // the signature count) overtakes the score table (measured, profiles/r03: 300 signatures 124 ms against 119 ms, 384: 169 ms) --
        // where it is eligible; otherwise the table (K <= 384) still beats the all-feature kernel by far.
        if (use_table && c->n_sigs > 256 && !needs_table_or_wide && c->fast_ok && !c->force_v1 && T >= 128 && !c->force_table) use_table = false;
        if (needs_table_or_wide && !use_table) run_wide = true;
        if (run_wide) {
            // falls through to the all-feature kernel below
        } else if (use_table) {
            HIP_TRY(c, c->d_ws.ensure(c->ws_total));
            if (const char* fill = getenv("SIMON_WS_FILL"))          // debugging aid: the workspace starts from a byte pattern (a read of something the prologue never wrote shows)
                HIP_TRY(c, hipMemsetAsync(c->d_ws.p, atoi(fill), c->ws_total, c->stream));
            if (want_placement) HIP_TRY(c, c->d_place_step.ensure((size_t)S * P));
            TableCold cold{};
            cold.ncls = c->d_t_ncls.p; cold.rank = c->d_rank.p; cold.cls_off = c->d_cls_off.p;
            cold.clsprefix = c->d_clsprefix.p; cold.a_pods = c->d_a_pods.p;
            cold.i_rq_cpu = c->d_i_rq_cpu.p; cold.i_rq_mem = c->d_i_rq_mem.p; cold.i_nz_cpu = c->d_i_nz_cpu.p; cold.i_nz_mem = c->d_i_nz_mem.p;
            cold.foldx = c->fold ? c->d_foldx.p : nullptr; cold.i_npods = c->d_i_npods.p; cold.sigs = c->d_sigs.p; cold.shapes = c->d_shapes.p; cold.scen = c->d_scen.p;
            cold.static_mask = c->has_mask ? c->d_t_mask.p : nullptr; cold.simon_raw = c->d_t_raw.p;
            cold.unscheduled = c->d_unsched.p; cold.used_cpu = c->d_used_cpu.p; cold.used_mem = c->d_used_mem.p;
            cold.N = c->N;
            cold.gpu_slices = want_slices ? reinterpret_cast<unsigned long long*>(c->d_gpu_slices.p) : nullptr;
            cold.na_raw = c->has_na ? c->d_t_na.p : nullptr; cold.tt_raw = c->has_tt ? c->d_t_tt.p : nullptr; cold.add_raw = c->has_add ? c->d_t_add.p : nullptr;
            if (c->has_ranks) { cold.rk_pos = c->d_rk_pos.p; cold.rk_rank = c->d_node_rank.p; }
            if (c->rest) {
                cold.xrows = c->d_xrows.p; cold.zdom = c->d_zdom.p; cold.xsig = c->d_xsig.p; cold.xalloc = c->d_xalloc.p; cold.i_xused = c->d_i_xused.p;
                cold.gsig = c->d_gsig.p; cold.gpu_cnt = c->d_gpu_cnt.p; cold.gpu_devtot = c->d_gpu_devtot.p; cold.i_gused = c->d_i_gused.p;
            }
            if (c->gfold && !c->rest) { cold.gpu_cnt = c->d_gpu_cnt.p; cold.gpu_devtot = c->d_gpu_devtot.p; cold.i_gused = c->d_i_gused.p; }
            if (c->spread) {
                cold.sp_ent = (const int2*)c->d_sp_ent.p; cold.spread_log = c->d_spread_log.p;
                cold.node_sets = c->d_node_sets.p; cold.set_words = (c->N + 63) / 64; cold.cls_zdom = c->d_cls_zdom.p;
                if (c->rest) cold.sp_word = c->d_sp_word.p;
            }
            const bool tprof = c->table_prof;
            if (tprof) { HIP_TRY(c, c->d_table_prof.ensure((size_t)S * 24)); HIP_TRY(c, hipMemsetAsync(c->d_table_prof.p, 0, (size_t)S * 192, c->stream)); cold.prof = c->d_table_prof.p; }
            HIP_TRY(c, c->d_table_cold.ensure(sizeof cold));
            HIP_TRY(c, hipMemcpyAsync(c->d_table_cold.p, &cold, sizeof cold, hipMemcpyHostToDevice, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));             // `cold` is a stack object
            TableLaunch f{};
            f.cold = reinterpret_cast<const TableCold*>(c->d_table_cold.p);
            f.cls_list = c->has_ranks ? c->d_rk_ids.p : c->d_cls_list.p; f.pods = c->d_podsC.p; f.orders = c->d_orders.p; f.perm = c->d_perm.p;
            f.ws_off = c->d_ws_off.p; f.ws = c->d_ws.p; f.coarse = c->table_coarse; f.rest = c->rest; f.spread = c->spread; f.aff = c->rest && !c->aff_idx.empty(); f.team = team; f.lds_ws = lds_ws; f.lds_x = lds_x;
            f.place_step = want_placement ? c->d_place_step.p : nullptr;
            f.sc = TableScalars{(c->N + 63) / 64, c->Cn_t, c->Cp, P, S, c->n_sigs, c->has_ranks ? c->N : 0, (c->has_na ? 1 : 0) | (c->has_tt ? 2 : 0) | (c->has_add ? 4 : 0) | (want_slices ? 8 : 0) | (c->sig_twins ? 16 : 0) | (c->fold ? 32 : 0) | ((c->spread && (c->ipa_fold || c->hard_fold)) ? 64 : 0) | (c->gfold ? 128 : 0), c->rest ? (int)c->zone_keys.size() : 0, c->rest ? c->rest_M : 0, c->rest ? c->rest_G : 0, c->rest ? c->rest_X : 0, c->spread ? c->sp_TH : 0, c->spread ? c->sp_TZ : 0, c->spread ? (int)c->sp_zkeys.size() : 0, ni_top, c->g_cpu, c->g_mem};
            HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
            HIP_TRY(c, launch_table(f, S, c->has_mask, c->nzeq, c->has_pin, table_lds, c->stream));
            if (want_placement)
                HIP_TRY(c, launch_unpermute(c->d_place_step.p, c->d_inv_orders.p, c->d_scen.p, S, P, c->d_place.p, c->stream));
            HIP_TRY(c, hipEventRecord(c->ev1, c->stream));
            if (tprof) {     // phase profile: mean ticks per scheduling cycle over the batch (profile builds only)
                HIP_TRY(c, hipStreamSynchronize(c->stream));
                std::vector<unsigned long long> hp((size_t)S * 24);
                HIP_TRY(c, hipMemcpy(hp.data(), c->d_table_prof.p, hp.size() * 8, hipMemcpyDeviceToHost));
                double acc[24] = {0};
                for (int s2 = 0; s2 < S; ++s2) for (int q = 0; q < 24; ++q) acc[q] += (double)hp[(size_t)s2 * 24 + q];
                fprintf(stderr, "[SIMON_TABLE_PROF] S=%d ticks/cycle: loop %.0f | row+summary read %.0f | key+wavemax %.0f | tie check %.0f | lds(shape,sn) %.0f | mem(state,row) %.0f | state update %.0f | eval+patch+store %.0f | REST assume %.0f | REST select %.0f | canonical tie-breaks per cycle %.3f\n",
                        S, acc[0] / S / P, acc[1] / S / P, acc[2] / S / P, acc[3] / S / P, acc[4] / S / P, acc[5] / S / P, acc[8] / S / P, acc[9] / S / P, acc[6] / S / P, acc[10] / S / P, acc[7] / S / P);
                if (c->rest)
                    fprintf(stderr, "[SIMON_TABLE_PROF] REST select, ticks/cycle (averaged over ALL pods): pod row %.0f | filter words + summaries %.0f | candidates %.0f | table rows of excluded bests %.0f (needed on %.3f of the cycles) | per-class best %.0f | class term %.0f | totals + tie %.0f | rest %.0f\n",
                            acc[11] / S / P, acc[12] / S / P, acc[13] / S / P, acc[14] / S / P, acc[20] / S / P, acc[15] / S / P, acc[16] / S / P, acc[17] / S / P, acc[10] / S / P);
                if (c->spread)
                    fprintf(stderr, "[SIMON_TABLE_PROF] spread pods, ticks/cycle: entries arrived %.0f | descriptor + first loads %.0f | counters, sizes %.0f | zone counters, Log, raw table %.0f | pass 1 %.0f | extremes, totals table %.0f | pass 2 %.0f | winner %.0f | counter stores %.0f\n",
                            acc[21] / S / P, acc[12] / S / P, acc[13] / S / P, acc[14] / S / P, acc[15] / S / P, acc[16] / S / P, acc[17] / S / P, acc[18] / S / P, acc[19] / S / P);
            }
            variant_used = SIMON_KERNEL_NARROW_CACHE;
            T = 64 * team; slots = (ni_top / (c->table_coarse ? 64 : 16) + 63) / 64; lds = table_lds;
            c->stats.n_launches = 1;
            table_used = true;
        } else if (c->fast_ok && !c->force_v1 && T >= 128) {
            lds = ((size_t)c->Cp * c->Cn * 2 * sizeof(int32_t) + 15) & ~(size_t)15;
            FastLaunch f{};
            f.a_cpu = c->d_a_cpu.p; f.a_mem = c->d_a_mem.p; f.a_pods = c->d_a_pods.p; f.ncls = c->d_ncls.p;
            f.i_rq_cpu = c->d_i_rq_cpu.p; f.i_rq_mem = c->d_i_rq_mem.p; f.i_nz_cpu = c->d_i_nz_cpu.p; f.i_nz_mem = c->d_i_nz_mem.p;
            f.i_npods = c->d_i_npods.p; f.pods = c->d_podsF.p; f.orders = c->d_orders.p; f.scen = c->d_scen.p; f.perm = c->d_perm.p;
            f.static_mask = c->has_mask ? c->d_mask.p : nullptr; f.simon_raw = c->d_raw32.p;
            f.unscheduled = c->d_unsched.p; f.used_cpu = c->d_used_cpu.p; f.used_mem = c->d_used_mem.p;
            f.placement = want_placement ? c->d_place.p : nullptr;
            f.sc = FastScalars{(c->N + 63) / 64, c->Cn, c->Cp, P, S, c->g_cpu, c->g_mem};
            HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
            HIP_TRY(c, launch_fast(f, T, slots, c->has_mask, c->nzeq, lds, c->stream));
            HIP_TRY(c, hipEventRecord(c->ev1, c->stream));
            variant_used = SIMON_KERNEL_NARROW_FAST;
        } else {
            lds = ((size_t)c->Cp * c->Cn * sizeof(int32_t) + 15) & ~(size_t)15;
            NarrowArgs a{};
            a.a_cpu = c->d_a_cpu.p; a.a_mem = c->d_a_mem.p; a.a_pods = c->d_a_pods.p; a.ncls = c->d_ncls.p;
            a.i_rq_cpu = c->d_i_rq_cpu.p; a.i_rq_mem = c->d_i_rq_mem.p; a.i_nz_cpu = c->d_i_nz_cpu.p; a.i_nz_mem = c->d_i_nz_mem.p;
            a.i_npods = c->d_i_npods.p;
            a.pods = c->d_podsN.p; a.orders = c->d_orders.p; a.scen = c->d_scen.p; a.perm = c->d_perm.p;
            a.static_mask = c->has_mask ? c->d_mask.p : nullptr;
            a.simon_raw = c->d_raw32.p;
            a.mask_words = (c->N + 63) / 64; a.Cn = c->Cn; a.Cp = c->Cp; a.P = P; a.S = S;
            a.g_cpu = c->g_cpu; a.g_mem = c->g_mem;
            a.unscheduled = c->d_unsched.p; a.used_cpu = c->d_used_cpu.p; a.used_mem = c->d_used_mem.p;
            a.placement = want_placement ? c->d_place.p : nullptr;
            HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
            HIP_TRY(c, launch_narrow(a, T, slots, c->has_mask, c->rcp_div, lds, c->stream));
            HIP_TRY(c, hipEventRecord(c->ev1, c->stream));
        }
    }
    if (run_wide) {
        if (!c->wide_staged) {   // a NARROW problem whose batch cannot use the cache kernel but has pinned pods
            int rcw = wide_stage(c->wide, *c, c->stream, c->err);
            if (rcw) return rcw;
            c->wide_staged = true;
        }
        variant_used = SIMON_KERNEL_WIDE;
        // Workgroup shape of the all-feature kernel.  A 256-thread group fits twice on a CU (launch bounds 256 x 2: two
        // independent barrier domains per SIMD), which wins as soon as the batch offers two groups per CU; with fewer
        // scenarios than that a CU holds ONE group and the 512-thread shape keeps all four SIMDs at two waves.
        // Measured on config 5 (50 000 pods x 2 500..5 000 nodes): S = 256: 512 threads 536 ms, 256 threads 734 ms;
        // S = 1 024: 512 threads 2 033 ms, 256 threads 1 611 ms.
        int n_cu = 256;
        (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, c->device);
        const bool two_groups = S >= 2 * n_cu && c->max_n <= 8192;          // 32 nodes per lane at most
        T = c->force_T ? c->force_T : (c->max_n <= 1024 || two_groups ? 256 : c->max_n <= 16384 ? 512 : 1024);
        if (T == 128) T = 256;                                  // (SIMON_WG=128: the all-feature kernel is built for 64 / 256 / 512 / 1 024 threads since round 5)
        HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
        int rc = wide_run(c->wide, *c, reinterpret_cast<const WideScenario*>(c->d_scen.p), nullptr, S, c->d_orders.p,
                          c->max_n, T, c->d_unsched.p, c->d_used_cpu.p, c->d_used_mem.p, c->d_used_vg.p,
                          want_placement ? c->d_place.p : nullptr, c->has_ranks ? c->d_node_rank.p : nullptr,
                          c->has_ranks ? c->d_node_inv.p : nullptr, want_slices ? c->d_gpu_slices.p : nullptr, c->stream, c->err);
        if (rc) return rc;
        HIP_TRY(c, hipEventRecord(c->ev1, c->stream));
    }
    if (want_risk) {                                             // (behind ev1: not part of kernel_ms; one wave per scenario, P / 64 steps)
        if (!c->prio_staged) {
            HIP_TRY(c, c->d_prio.ensure((size_t)P));
            HIP_TRY(c, hipMemcpyAsync(c->d_prio.p, c->p_priority.data(), (size_t)P * 4, hipMemcpyHostToDevice, c->stream));
            c->prio_staged = true;
        }
        HIP_TRY(c, c->d_risk.ensure((size_t)S));
        hipLaunchKernelGGL(preempt_risk_kernel_impl, dim3(S), dim3(64), 0, c->stream, c->d_place.p, c->d_orders.p, c->d_scen.p, c->d_prio.p, P,
                           c->init_min_priority, c->d_risk.p);
        HIP_TRY(c, hipGetLastError());
        c->have_risk = true;
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    float ms = 0;
    (void)hipEventElapsedTime(&ms, c->ev0, c->ev1);
    c->stats.kernel_ms = ms;
    c->stats.n_launches = 1;
    c->stats.kernel_variant = variant_used;
    c->stats.kernel_generation = table_used ? (c->spread ? 7 : c->rest ? 6 : c->table_coarse ? 5 : 4) : variant_used == SIMON_KERNEL_NARROW_FAST ? 2 : variant_used == SIMON_KERNEL_NARROW ? 1 : 0;
    c->stats.workgroup_size = T;
    c->stats.slots_per_lane = slots;
    c->stats.lds_bytes = (int64_t)lds;
    c->have_results = true;
    c->have_placement = want_placement != 0;
    c->have_slices = want_slices;
    // generations 1 / 2 never see a GPU request (has_gpu routes the problem to generation 6 or the all-feature kernel)
    return SIMON_OK;
}


// simon_batch_out as the caller filled it, widened to the current layout: a v3 struct (no gpu_slices member) reads as gpu_slices = NULL
}  // extern "C"
static bool batch_out_view(const simon_batch_out* out, simon_batch_out* v) {
    if (out->struct_size == sizeof(simon_batch_out)) { *v = *out; return true; }
    if (out->struct_size != SIMON_BATCH_OUT_SIZE_V3) return false;
    std::memset(v, 0, sizeof *v);
    std::memcpy(v, out, SIMON_BATCH_OUT_SIZE_V3);
    v->struct_size = sizeof *v;
    return true;
}
extern "C" {

int simon_fetch_results(simon_ctx* c, simon_batch_out* caller_out) {
    if (!c || !caller_out) return SIMON_EINVAL;
    simon_batch_out view, *out = &view;
    if (!batch_out_view(caller_out, out)) return fail(c, SIMON_EINVAL, "simon_batch_out size mismatch");
    if (!c->have_results) return fail(c, SIMON_ESTATE, "fetch_results: nothing has run");
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t S = c->S, P = c->P;
    HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
    if (out->unscheduled) HIP_TRY(c, hipMemcpyAsync(out->unscheduled, c->d_unsched.p, S * 4, hipMemcpyDeviceToHost, c->stream));
    if (out->used_cpu) HIP_TRY(c, hipMemcpyAsync(out->used_cpu, c->d_used_cpu.p, S * 8, hipMemcpyDeviceToHost, c->stream));
    if (out->used_mem) HIP_TRY(c, hipMemcpyAsync(out->used_mem, c->d_used_mem.p, S * 8, hipMemcpyDeviceToHost, c->stream));
    if (out->used_vg) {   // only the all-feature kernel tracks Open-Local volume groups
        if (c->stats.kernel_variant == SIMON_KERNEL_WIDE) HIP_TRY(c, hipMemcpyAsync(out->used_vg, c->d_used_vg.p, S * 8, hipMemcpyDeviceToHost, c->stream));
        else memset(out->used_vg, 0, S * 8);
    }
    if (out->placement) {
        if (!c->have_placement) return fail(c, SIMON_ESTATE, "fetch_results: the last run skipped placements");
        HIP_TRY(c, hipMemcpyAsync(out->placement, c->d_place.p, S * P * 4, hipMemcpyDeviceToHost, c->stream));
    }
    if (out->gpu_slices) {
        if (c->have_slices) HIP_TRY(c, hipMemcpyAsync(out->gpu_slices, c->d_gpu_slices.p, S * P * 8, hipMemcpyDeviceToHost, c->stream));
        else if (!c->has_gpu) memset(out->gpu_slices, 0, S * P * 8);      // no GPU request anywhere: nothing was ever booked
        else return fail(c, SIMON_ESTATE, "fetch_results: the last run did not record GPU devices (SIMON_WANT_GPU_SLICES)");
    }
    HIP_TRY(c, hipEventRecord(c->ev1, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    float ms = 0;
    (void)hipEventElapsedTime(&ms, c->ev0, c->ev1);
    c->stats.d2h_ms = ms;
    return SIMON_OK;
}

int simon_fetch_placement(simon_ctx* c, int32_t scenario, int32_t* placement) {
    if (!c || !placement) return SIMON_EINVAL;
    if (!c->have_results || !c->have_placement) return fail(c, SIMON_ESTATE, "fetch_placement: no placements on device");
    if (scenario < 0 || scenario >= c->S) return fail(c, SIMON_EINVAL, "fetch_placement: scenario out of range");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpyAsync(placement, c->d_place.p + (size_t)scenario * c->P, (size_t)c->P * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SIMON_OK;
}

int simon_fetch_gpu_slices(simon_ctx* c, int32_t scenario, uint64_t* slices) {
    if (!c || !slices) return SIMON_EINVAL;
    if (!c->have_results) return fail(c, SIMON_ESTATE, "fetch_gpu_slices: nothing has run");
    if (scenario < 0 || scenario >= c->S) return fail(c, SIMON_EINVAL, "fetch_gpu_slices: scenario out of range");
    if (!c->has_gpu) { memset(slices, 0, (size_t)c->P * 8); return SIMON_OK; }
    if (!c->have_slices) return fail(c, SIMON_ESTATE, "fetch_gpu_slices: the last run did not record GPU devices (SIMON_WANT_GPU_SLICES)");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpyAsync(slices, c->d_gpu_slices.p + (size_t)scenario * c->P, (size_t)c->P * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SIMON_OK;
}

int simon_fetch_preempt_risk(simon_ctx* c, uint8_t* risk) {
    if (!c || !risk) return SIMON_EINVAL;
    if (!c->have_results) return fail(c, SIMON_ESTATE, "fetch_preempt_risk: nothing has run");
    if (!c->have_risk) { memset(risk, 0, (size_t)c->S); return SIMON_OK; }       // one priority for all: DefaultPreemption finds no victim anywhere
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpyAsync(risk, c->d_risk.p, (size_t)c->S, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SIMON_OK;
}

int simon_run_batch(simon_ctx* c, const simon_scenario* scen, int32_t S, const int32_t* orders, int32_t n_orders,
                    simon_batch_out* out) {
    if (!c || !out) return SIMON_EINVAL;
    simon_batch_out view;
    if (!batch_out_view(out, &view)) return fail(c, SIMON_EINVAL, "simon_batch_out size mismatch");
    int rc = simon_load_scenarios(c, scen, S, orders, n_orders);
    if (rc) return rc;
    rc = simon_run_loaded(c, (view.placement ? SIMON_WANT_PLACEMENT : 0) | (view.gpu_slices ? SIMON_WANT_GPU_SLICES : 0));
    if (rc) return rc;
    return simon_fetch_results(c, &view);
}

int simon_min_plan(simon_ctx* c, int32_t max_cpu_pct, int32_t max_mem_pct, simon_plan* best) {
    return simon_min_plan_vg(c, max_cpu_pct, max_mem_pct, 100, best, nullptr);
}

// satisfyResourceSetting + minimum over the batch on the device: leaves the key (n_nodes << 32 | scenario, ~0 = none) in c->d_plan
static int launch_plan(simon_ctx* c, int32_t max_cpu_pct, int32_t max_mem_pct, int32_t max_vg_pct) {
    if (max_vg_pct > 100 || max_vg_pct < 0) max_vg_pct = 100;
    if (!c->have_results) return fail(c, SIMON_ESTATE, "min_plan: nothing has run");
    HIP_TRY(c, hipSetDevice(c->device));
    if (max_cpu_pct > 100 || max_cpu_pct < 0) max_cpu_pct = 100;  // apply.go:698-700
    if (max_mem_pct > 100 || max_mem_pct < 0) max_mem_pct = 100;
    HIP_TRY(c, hipMemsetAsync(c->d_plan.p, 0xFF, sizeof(unsigned long long), c->stream));
    const int blocks = std::min(64, (c->S + 255) / 256);
    hipLaunchKernelGGL(plan_kernel, dim3(blocks), dim3(256), 0, c->stream, c->d_scen.p, c->S, c->d_unsched.p,
                       c->d_used_cpu.p, c->d_used_mem.p, c->d_prefix_cpu.p, c->d_prefix_mem.p, max_cpu_pct, max_mem_pct,
                       c->has_local ? c->d_used_vg.p : nullptr, c->d_prefix_vg.p, max_vg_pct, c->d_plan.p);
    HIP_TRY(c, hipGetLastError());
    return SIMON_OK;
}

int simon_min_plan_device(simon_ctx* c, int32_t max_cpu_pct, int32_t max_mem_pct, int32_t max_vg_pct, void** d_key, void** stream) {
    if (!c || !d_key) return SIMON_EINVAL;
    const int rc = launch_plan(c, max_cpu_pct, max_mem_pct, max_vg_pct);
    if (rc) return rc;
    *d_key = c->d_plan.p;                                          // ordered behind plan_kernel on `stream`: a collective enqueued there needs no host sync
    if (stream) *stream = (void*)c->stream;
    return SIMON_OK;
}

int simon_min_plan_vg(simon_ctx* c, int32_t max_cpu_pct, int32_t max_mem_pct, int32_t max_vg_pct, simon_plan* best,
                      int32_t* vg_pct) {
    if (!c || !best) return SIMON_EINVAL;
    if (vg_pct) *vg_pct = 0;
    const int rcl = launch_plan(c, max_cpu_pct, max_mem_pct, max_vg_pct);
    if (rcl) return rcl;
    unsigned long long key = 0;
    HIP_TRY(c, hipMemcpyAsync(&key, c->d_plan.p, sizeof key, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    memset(best, 0, sizeof *best);
    best->scenario = -1;
    if (key == ~0ull) return SIMON_OK;
    const int s = (int)(key & 0xFFFFFFFFu);
    int64_t uc = 0, um = 0;
    HIP_TRY(c, hipMemcpy(&uc, c->d_used_cpu.p + s, 8, hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy(&um, c->d_used_mem.p + s, 8, hipMemcpyDeviceToHost));
    const int n = c->scen[s].n_nodes;
    int64_t ac = 0, am = 0;
    for (int j = 0; j < n; ++j) { ac += c->alloc_cpu[j]; am += c->alloc_mem[j]; }
    best->found = 1; best->scenario = s; best->n_nodes = n; best->order_id = c->scen[s].order_id;
    best->cpu_pct = (int)((double)uc / (double)ac * 100.0);
    best->mem_pct = (int)((double)(um * 1000) / (double)(am * 1000) * 100.0);
    best->used_cpu = uc; best->used_mem = um;
    if (vg_pct && c->has_local && c->prefix_vg[n] != 0) {
        int64_t uv = 0;
        HIP_TRY(c, hipMemcpy(&uv, c->d_used_vg.p + s, 8, hipMemcpyDeviceToHost));
        *vg_pct = (int)((double)uv / (double)c->prefix_vg[n] * 100.0);
    }
    return SIMON_OK;
}

static int explain_impl(simon_ctx* c, int n_nodes, const int32_t* order, int ranked_scenario, int32_t* failed_pods,
                        uint16_t* fail_codes, int32_t max_failed) {
    int rc = stage(c);
    if (rc) return rc;
    if (n_nodes < 0 || n_nodes > c->N) return fail(c, SIMON_EINVAL, "explain: n_nodes out of range");
    for (int i = 0; i < c->P; ++i) if (order[i] < 0 || order[i] >= c->P) return fail(c, SIMON_EINVAL, "explain: bad order");
    HIP_TRY(c, hipSetDevice(c->device));
    if (!c->wide_staged) {   // NARROW problem: the failure codes still come from the all-feature kernel
        rc = wide_stage(c->wide, *c, c->stream, c->err);
        if (rc) return rc;
        c->wide_staged = true;
    }
    int T = c->force_T ? c->force_T : (n_nodes <= 512 ? 256 : n_nodes <= 4096 ? 512 : 1024);
    if (T == 128) T = 256;
    // with per-scenario node ranks the replay must break ties exactly like the batch did: the ranks of THAT scenario
    const int32_t *rk = nullptr, *iv = nullptr;
    if (ranked_scenario >= 0) {
        rk = c->d_node_rank.p + (size_t)ranked_scenario * c->N;
        iv = c->d_node_inv.p + (size_t)ranked_scenario * c->N;
    }
    c->explain_nodes = n_nodes;
    c->explain_ran = true;
    return wide_explain(c->wide, *c, n_nodes, order, failed_pods, fail_codes, max_failed, T, rk, iv, c->stream, c->err, &c->explain_detail);
}

int simon_explain(simon_ctx* c, simon_scenario scen, const int32_t* order, int32_t* failed_pods, uint16_t* fail_codes,
                  int32_t max_failed) {
    if (!c || !order || !failed_pods || !fail_codes || max_failed <= 0) return c ? fail(c, SIMON_EINVAL, "explain: bad arguments") : SIMON_EINVAL;
    if (c->has_ranks)   // an ad-hoc scenario has no rank row: which loaded scenario's tie-break order would apply is ambiguous
        return fail(c, SIMON_ESTATE, "explain: per-scenario node ranks are loaded; use simon_explain_loaded(scenario index)");
    return explain_impl(c, scen.n_nodes, order, -1, failed_pods, fail_codes, max_failed);
}

int simon_explain_loaded(simon_ctx* c, int32_t scenario, int32_t* failed_pods, uint16_t* fail_codes, int32_t max_failed) {
    if (!c || !failed_pods || !fail_codes || max_failed <= 0) return c ? fail(c, SIMON_EINVAL, "explain_loaded: bad arguments") : SIMON_EINVAL;
    if (!c->staged || c->S <= 0) return fail(c, SIMON_ESTATE, "explain_loaded: no scenarios loaded");
    if (scenario < 0 || scenario >= c->S) return fail(c, SIMON_EINVAL, "explain_loaded: scenario %d outside [0,%d)", scenario, c->S);
    const ScenarioDesc& sd = c->scen[scenario];
    return explain_impl(c, sd.n_nodes, c->h_orders.data() + (size_t)sd.order_id * c->P, c->has_ranks ? scenario : -1,
                        failed_pods, fail_codes, max_failed);
}

int simon_get_stats(simon_ctx* c, simon_stats* st) {
    if (!c || !st) return SIMON_EINVAL;
    *st = c->stats;
    return SIMON_OK;
}

int simon_explain_local_detail(simon_ctx* c, int64_t* detail, int32_t max_failed) {
    if (!c || !detail || max_failed <= 0) return c ? fail(c, SIMON_EINVAL, "explain_local_detail: bad arguments") : SIMON_EINVAL;
    if (!c->explain_ran) return fail(c, SIMON_ESTATE, "explain_local_detail: no simon_explain call yet");
    const size_t row = (size_t)std::max(c->explain_nodes, 0) * 4;
    const size_t rows = row ? std::min<size_t>(c->explain_detail.size() / row, (size_t)max_failed) : 0;
    if (rows) std::memcpy(detail, c->explain_detail.data(), rows * row * sizeof(int64_t));
    return (int)rows;
}

int simon_device_results(simon_ctx* c, void** d_unscheduled, void** d_used_cpu, void** d_used_mem) {
    if (!c) return SIMON_EINVAL;
    if (!c->have_results) return fail(c, SIMON_ESTATE, "device_results: nothing has run");
    if (d_unscheduled) *d_unscheduled = c->d_unsched.p;
    if (d_used_cpu) *d_used_cpu = c->d_used_cpu.p;
    if (d_used_mem) *d_used_mem = c->d_used_mem.p;
    return SIMON_OK;
}

}  // extern "C"
