// simon_cache.hip -- NARROW kernel, third generation: one WAVE = one capacity-planning scenario,
// per-(pod signature, node) score table resident in LDS.
//
// Observation ("further lever"): a scheduling cycle changes ONE node.  For a pod
// stream drawn from K <= 64 distinct request signatures (replicas of workloads), the pair
// (feasible?, LeastAllocated + BalancedAllocation) of every (signature, node) is therefore a
// table with K x n one-byte entries of which only one COLUMN (K bytes) changes per cycle:
//   * filter + score of a pod  = read ONE ROW of the table (n bytes, ds_read_b128: 16 nodes per
//     lane), packed 16-bit max tree, one DPP wave reduction            -> findNodesThatFitPod +
//     prioritizeNodes + selectHost (V/core/generic_scheduler.go:131-209);
//   * assume (V/scheduler.go:371 -> NodeInfo.AddPod, V/framework/types.go:482-508) = lane k
//     re-evaluates signature k on the touched node with EXACTLY the arithmetic of
//     simon_fast.hip (same fp64 sequences => same bits) and stores one byte.
// Nodes are laid out class-major inside a scenario (stable in canonical order, every class
// segment padded to 16) so that the 16 nodes of a lane share their Simon node class: the
// normalised Simon/Open-Gpu-Share term (pkg/simulator/plugin/simon.go:76-101) is then one add per
// lane AFTER the in-lane max.  Ties are broken on the canonical node index carried in the key,
// i.e. determinised selectHost = first maximum in nodeTree.list() order.
// The per-signature count of feasible nodes per node class is maintained incrementally
// (feasibility only ever decreases: Requested grows, free pod slots shrink), which replaces the
// phase-A reduction of the earlier generations by one LDS read + ballot.
//
// No barriers (one wave), no global traffic in the loop except the scalar pod-row stream and one
// coalesced 256-byte placement store per 64 cycles.
//
// Eligibility (host, simon_hip.hip): NARROW preconditions + K <= 64 signatures, <= 64 node
// shapes, Cn <= 32, no zero-capacity node, padded scenario size <= 2047, LDS <= 160 KiB.
#include "simon_cache.h"

#include <algorithm>

namespace simon {

__device__ __forceinline__ int la_term_c(double r, double rc100) {   // == la_term_f of simon_fast.hip
    const double C = 100.0 + 0x1.0p-33;
    return (int)__builtin_fma(-r, rc100, C);
}

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pkmax(unsigned a, unsigned b) {
    u16x2 x = __builtin_bit_cast(u16x2, a), y = __builtin_bit_cast(u16x2, b);
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(x, y));
}

// LDS carve (all offsets multiples of 16).  GLOBAL: the score table and the node state live in a
// per-workgroup HBM workspace (L2 / Infinity-Cache resident) instead of LDS.
struct Carve {
    int tab, state, nz, canon, cnt, raw, sn2, sig, shape, seg, tmp, total;
    int ws_tab, ws_state, ws_nz, ws_total;   // workspace offsets (GLOBAL only)
};
__host__ __device__ inline Carve carve(int K, int stride, int ni_max, int Cn, int Cp, int n_shapes, bool nzeq, bool global) {
    auto al = [](int x) { return (x + 15) & ~15; };
    Carve c;
    int o = 0;
    c.tab = o; o += global ? 0 : al(K * stride);
    c.state = o; o += global ? 0 : ni_max * 16;
    c.nz = o; o += (global || nzeq) ? 0 : al(ni_max * 8);
    c.canon = o; o += al(ni_max * 2);
    c.cnt = o; o += Cn * 64 * 4;
    c.raw = o; o += al(Cp * Cn * 4);
    c.sn2 = o; o += al(Cp * Cn * 4);
    c.sig = o; o += K * 48;
    c.shape = o; o += n_shapes * 48;
    c.seg = o; o += 32 * 4;
    c.tmp = o; o += 32 * 4;
    c.total = o;
    int w = 0;
    c.ws_tab = w; w += (al(K * stride) + 127) & ~127;
    c.ws_state = w; w += (ni_max * 16 + 127) & ~127;
    c.ws_nz = w; w += nzeq ? 0 : ((ni_max * 8 + 127) & ~127);
    c.ws_total = w;
    return c;
}

// 32-way uniform switch over the register-resident node state (REGSTATE)
#define SIMON_ST16(X, o) X(o + 0) X(o + 1) X(o + 2) X(o + 3) X(o + 4) X(o + 5) X(o + 6) X(o + 7) X(o + 8) X(o + 9) X(o + 10) \
    X(o + 11) X(o + 12) X(o + 13) X(o + 14) X(o + 15)

template <int SLOTS, bool HAS_MASK, bool NZEQ, bool GLOBAL, bool REGSTATE>
__global__ __launch_bounds__(64) void cache_kernel(
    const int32_t* __restrict__ ncls, const int32_t* __restrict__ rank, const int32_t* __restrict__ shape_of,
    const int32_t* __restrict__ a_pods, const uint32_t* __restrict__ i_rq_cpu, const uint32_t* __restrict__ i_rq_mem,
    const uint32_t* __restrict__ i_nz_cpu, const uint32_t* __restrict__ i_nz_mem, const int32_t* __restrict__ i_npods,
    const int32_t* __restrict__ clsprefix, const SigRow* __restrict__ sigs, const ShapeRow* __restrict__ shapes,
    const PodRowC* __restrict__ pods, const int32_t* __restrict__ orders, const ScenarioDesc* __restrict__ scen,
    const int32_t* __restrict__ perm, const uint64_t* __restrict__ static_mask, const int32_t* __restrict__ simon_raw,
    int32_t* __restrict__ unscheduled, int64_t* __restrict__ used_cpu, int64_t* __restrict__ used_mem,
    int32_t* __restrict__ place_step, unsigned char* ws, const CacheScalars sc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int Cn = sc.Cn, Cp = sc.Cp, P = sc.P, K = sc.K, stride = sc.stride;
    const Carve cv = carve(K, stride, sc.ni_max, Cn, Cp, sc.n_shapes, NZEQ, GLOBAL);
    unsigned char* const wsb = GLOBAL ? ws + (size_t)blockIdx.x * (size_t)cv.ws_total : nullptr;
    unsigned char* s_tab = GLOBAL ? wsb + cv.ws_tab : smem + cv.tab;
    uint4* s_state = (uint4*)(GLOBAL ? wsb + cv.ws_state : smem + cv.state);   // {Requested cpu, mem, free pod slots, shape | class<<16}
    uint2* s_nz = (uint2*)(GLOBAL ? wsb + cv.ws_nz : smem + cv.nz);            // NonZeroRequested {cpu, mem} (only when !NZEQ)
    unsigned short* s_canon = (unsigned short*)(smem + cv.canon);
    int* s_cnt = (int*)(smem + cv.cnt);              // [Cn][64]: feasible nodes of class d for signature k
    int* s_raw = (int*)(smem + cv.raw);
    int* s_sn2 = (int*)(smem + cv.sn2);
    const SigRow* s_sig = (const SigRow*)(smem + cv.sig);
    const ShapeRow* s_shape = (const ShapeRow*)(smem + cv.shape);
    int* s_seg = (int*)(smem + cv.seg);
    int* s_tmp = (int*)(smem + cv.tmp);

    const int lane = threadIdx.x;
    const int s = perm[blockIdx.x];
    const int n = scen[s].n_nodes;
    const int32_t* __restrict__ order = orders + (size_t)scen[s].order_id * P;

    // ---- prologue 1: clear, tables -> LDS ----------------------------------------------------
    if (!GLOBAL) for (int i = lane; i < cv.canon / 16; i += 64) ((uint4*)smem)[i] = make_uint4(0, 0, 0, 0);   // tab, state, nz
    for (int i = lane; i < Cn * 64; i += 64) s_cnt[i] = 0;
    for (int i = lane; i < Cp * Cn; i += 64) s_raw[i] = simon_raw[i];
    for (int i = lane; i < K * 12; i += 64) ((int*)(smem + cv.sig))[i] = ((const int*)sigs)[i];
    for (int i = lane; i < sc.n_shapes * 12; i += 64) ((int*)(smem + cv.shape))[i] = ((const int*)shapes)[i];
    for (int i = lane; i < sc.ni_max; i += 64) s_canon[i] = 0xFFFFu;
    // class segments: count of class-d nodes among the first n canonical nodes, padded to 16
    int cnt_d = (lane < Cn) ? clsprefix[(size_t)n * Cn + lane] : 0;
    int pad_d = (cnt_d + 15) & ~15;
    int incl = pad_d;
#pragma unroll
    for (int off = 1; off < 32; off <<= 1) {
        const int o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
    }
    if (lane < 32) s_seg[lane] = incl - pad_d;
    const int ni = __builtin_amdgcn_readlane(incl, 31);               // padded scenario size (lanes >= Cn add 0)
    const unsigned bits_all = (unsigned)__ballot(cnt_d > 0);
    __syncthreads();
    // ---- prologue 2: scatter canonical indices into the class-major layout ------------------
    for (int j = lane; j < n; j += 64) s_canon[s_seg[ncls[j]] + rank[j]] = (unsigned short)j;
    // sn2_full[c][d] = 2 * NormalizeScore(raw[c][d]) over the node classes present (simon.go:76-101)
    for (int i = lane; i < Cp * Cn; i += 64) {
        const int c = i / Cn;
        long long lo = 0x7fffffffll, hi = -0x7fffffffll;
        for (int d = 0; d < Cn; ++d)
            if ((bits_all >> d) & 1u) {
                const long long r = s_raw[c * Cn + d];
                lo = r < lo ? r : lo;
                hi = r > hi ? r : hi;
            }
        const long long range = hi - lo;
        s_sn2[i] = (range > 0) ? (int)(2 * ((((long long)s_raw[i] - lo) * 100) / range)) : 0;
    }
    __syncthreads();

    // (feasible, LeastAllocated + BalancedAllocation) of one signature on one node: byte = 0 when
    // NodeResourcesFit fails (fit.go:230-302), else 1 + score.  Same fp64 sequences as
    // simon_fast.hip::eval_slot.
    auto eval_node = [&](double q_req_c, double q_req_m, double q_nz_c, double q_nz_m, bool q_zero, double rq_c, double rq_m,
                         double nzs_c, double nzs_m, int freep, const ShapeRow& sh) -> unsigned {
        const double t_c = rq_c + q_req_c, t_m = rq_m + q_req_m;
        const bool res_ok = (sh.cap_c >= t_c) && (sh.cap_m >= t_m);
        const bool ok = (freep >= 1) && (q_zero || res_ok);
        // resource_allocation.go:91-98: requested = NonZeroRequested + pod non-zero request
        const double r_c = NZEQ ? t_c : nzs_c + q_nz_c;
        const double r_m = NZEQ ? t_m : nzs_m + q_nz_m;
        const bool ge_c = r_c >= sh.cap_c, ge_m = r_m >= sh.cap_m;
        const int la_c = ge_c ? 0 : la_term_c(r_c, sh.rc100_c);           // least_allocated.go:108-117
        const int la_m = ge_m ? 0 : la_term_c(r_m, sh.rc100_m);
        const double cf = div_by_rcp(r_c, sh.cap_c, sh.rc_c);             // balanced_allocation.go:82-119
        const double mf = div_by_rcp(r_m, sh.cap_m, sh.rc_m);
        const int bs = (int)((1.0 - __builtin_fabs(cf - mf)) * 100.0);
        const int base = ((la_c + la_m) >> 1) + ((ge_c || ge_m) ? 0 : bs);
        return ok ? (unsigned)(base + 1) : 0u;
    };

    // ---- prologue 3: node rows + the whole table; lanes = 64 consecutive positions, loop over signatures
    for (int p0 = 0; p0 < ni; p0 += 64) {
        const int p = p0 + lane;
        const unsigned cj = p < ni ? s_canon[p] : 0xFFFFu;
        const bool real = cj != 0xFFFFu;
        const int j = real ? (int)cj : 0;
        const int d = real ? ncls[j] : 0;
        const uint4 st = real ? make_uint4(i_rq_cpu[j], i_rq_mem[j], (unsigned)(a_pods[j] - i_npods[j]),
                                           (unsigned)shape_of[j] | ((unsigned)d << 16))
                              : make_uint4(0, 0, 0, 0);
        uint2 z = make_uint2(0, 0);
        if (!NZEQ && real) z = make_uint2(i_nz_cpu[j], i_nz_mem[j]);
        if (!REGSTATE && p < sc.ni_max) {
            s_state[p] = st;
            if (!NZEQ) s_nz[p] = z;
        }
        const ShapeRow sh = s_shape[st.w & 0xFFFFu];
        // classes present in this chunk form a contiguous range (class-major layout)
        const int dlo = wave_min_i32(real ? d : 0x7fffffff), dhi = wave_max_i32(real ? d : (int)0x80000000);
        for (int k = 0; k < K; ++k) {
            const SigRow q = s_sig[k];
            unsigned b = eval_node(q.req_c, q.req_m, q.nz_c, q.nz_m, q.flags & 1u, (double)st.x, (double)st.y, (double)z.x,
                                   (double)z.y, (int)st.z, sh);
            b = real ? b : 0u;
            if (HAS_MASK) {   // NodeUnschedulable/NodeName/TaintToleration/NodeAffinity: static per (class, node)
                const uint64_t w = static_mask[(size_t)q.cls * sc.mask_words + (j >> 6)];
                b = ((w >> (j & 63)) & 1ull) ? b : 0u;
            }
            if (p < stride) s_tab[(size_t)k * stride + p] = (unsigned char)b;
            for (int dd = dlo; dd <= dhi; ++dd) {
                const int c = __popcll(__ballot(b != 0u && d == dd));
                if (lane == 0 && c) s_cnt[dd * 64 + k] += c;
            }
        }
    }
    if (GLOBAL) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // this lane's signature (lane k re-evaluates signature k on a touched node)
    const int kk = lane < K ? lane : 0;
    const double my_req_c = s_sig[kk].req_c, my_req_m = s_sig[kk].req_m;
    const double my_nz_c = s_sig[kk].nz_c, my_nz_m = s_sig[kk].nz_m;
    const bool my_zero = s_sig[kk].flags & 1u;
    unsigned char* my_col = s_tab + (size_t)kk * stride;

    // node class of this lane's 16-node blocks (class segments are multiples of 16); REGSTATE: the
    // dynamic state of those 16 x SLOTS nodes lives in this lane's registers
    const int nblk = ni >> 4;
    int ncl4[SLOTS];
    unsigned rqc[REGSTATE ? SLOTS * 16 : 1], rqm[REGSTATE ? SLOTS * 16 : 1], fps[REGSTATE ? SLOTS * 16 : 1];
#pragma unroll
    for (int q = 0; q < SLOTS; ++q) {
        const int b = q * 64 + lane;
        const unsigned cj0 = (b < nblk) ? s_canon[b * 16] : 0xFFFFu;
        ncl4[q] = (cj0 != 0xFFFFu) ? ncls[cj0] * 4 : 0;
        if (REGSTATE) {
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const unsigned cj = (b < nblk) ? s_canon[b * 16 + t] : 0xFFFFu;
                const bool real = cj != 0xFFFFu;
                const int j = real ? (int)cj : 0;
                rqc[q * 16 + t] = real ? i_rq_cpu[j] : 0u;
                rqm[q * 16 + t] = real ? i_rq_mem[j] : 0u;
                // free pod slots << 8 (signed, upper 24 bits: decrements never borrow from the shape id) | shape
                fps[q * 16 + t] = real ? (((unsigned)(a_pods[j] - i_npods[j]) << 8) | (unsigned)shape_of[j]) : 0u;
            }
        }
    }

    int unsched = 0, plreg = 0;
    int32_t* __restrict__ place = place_step ? place_step + (size_t)s * P : nullptr;

    // pod stream: 64 rows per vector load (lane l holds step i0 + l), one chunk ahead
    int4 cur = make_int4(0, -1, 0x7fffffff, 0), nxt = cur;
    auto load_chunk = [&](int i0) -> int4 {
        const int idx = i0 + lane;
        if (idx >= P) return make_int4(0, -1, 0x7fffffff, 0);
        const PodRowC r = pods[order[idx]];
        return make_int4(r.sig, r.preset, r.gate, r.cls);
    };
    if (P > 0) nxt = load_chunk(0);

    // table row of the NEXT pod's signature, loaded one cycle ahead; the one byte the current
    // cycle changes in it is patched in registers
    uint4 Rn[SLOTS];
    auto load_row = [&](int k) {
        const uint4* rowp = (const uint4*)(s_tab + (size_t)k * stride);
#pragma unroll
        for (int q = 0; q < SLOTS; ++q) {
            const int b = q * 64 + lane;
            if (SLOTS == 1 || q * 64 < nblk) Rn[q] = rowp[b < nblk ? b : 0];
            else Rn[q] = make_uint4(0, 0, 0, 0);
        }
    };
    if (P > 0) load_row(__builtin_amdgcn_readlane(nxt.x, 0));

    for (int i = 0; i < P; ++i) {
        const int il = i & 63;
        if (il == 0) { cur = nxt; nxt = load_chunk(i + 64); }
        const int r_sig = __builtin_amdgcn_readlane(cur.x, il), r_preset = __builtin_amdgcn_readlane(cur.y, il);
        const int r_gate = __builtin_amdgcn_readlane(cur.z, il), r_cls = __builtin_amdgcn_readlane(cur.w, il);
        const int k_next = il < 63 ? __builtin_amdgcn_readlane(cur.x, (il + 1) & 63) : __builtin_amdgcn_readlane(nxt.x, 0);
        uint4 R[SLOTS];
#pragma unroll
        for (int q = 0; q < SLOTS; ++q) R[q] = Rn[q];
        if (i + 1 < P && !(sc.ablate & 8)) load_row(k_next);

        int res, pstar = -1;
        if (r_gate >= n) {
            res = -2;                                                  // pod not part of this scenario
        } else if (r_preset >= 0) {                                    // addPodToCache path (V/eventhandlers.go:223-236)
            res = r_preset;
            pstar = __builtin_amdgcn_readfirstlane(s_seg[ncls[r_preset]] + rank[r_preset]);
        } else {
            const int k = r_sig;
            // -------- which node classes still have a feasible node for this signature --------
            const int cvv = (lane < Cn) ? s_cnt[lane * 64 + k] : 0;
            const unsigned bits = (unsigned)__ballot(cvv > 0);
            if (bits == 0u) {                                          // FitError: pod deleted, state unchanged
                ++unsched;
                res = -1;
            } else {
                // -------- SimonPlugin/GpuSharePlugin NormalizeScore row (x2: both plugins) ----
                const int* snrow;
                if (bits == bits_all) {
                    snrow = s_sn2 + r_cls * Cn;
                } else {
                    const int c = lane < Cn ? lane : 0;
                    const int rawc = s_raw[r_cls * Cn + c];
                    const bool inb = (lane < Cn) && ((bits >> c) & 1u);
                    const int lo = wave_min_i32(inb ? rawc : 0x7fffffff);
                    const int hi = wave_max_i32(inb ? rawc : (int)0x80000000);
                    const int range = hi - lo;
                    const double rr = range ? 1.0 / (double)range : 0.0;
                    const int sn = range ? (int)__builtin_fma((double)(rawc - lo) * 100.0, rr, 0.5 * rr) : 0;
                    if (lane < 32) s_tmp[lane] = inb ? 2 * sn : 0;
                    __builtin_amdgcn_wave_barrier();
                    snrow = s_tmp;
                }
                // -------- row scan: 16 nodes per lane and slot ---------------------------------
                unsigned key = 0;
#pragma unroll
                for (int q = 0; q < SLOTS; ++q) {
                    const int b = q * 64 + lane;
                    if (SLOTS == 1 || q * 64 < nblk) {
                        const uint4 Rq = R[q];
                        // key16 = byte << 4 | (15 - position): max = best base score, first position on ties
                        const unsigned M8 = 0x00FF00FFu;
#define SIMON_LO(w, q4) ((((w) & M8) << 4) | (unsigned)((15 - (q4)) | ((15 - ((q4) + 2)) << 16)))
#define SIMON_HI(w, q4) (((((w) >> 8) & M8) << 4) | (unsigned)((15 - ((q4) + 1)) | ((15 - ((q4) + 3)) << 16)))
                        unsigned m0 = pkmax(SIMON_LO(Rq.x, 0), SIMON_HI(Rq.x, 0));
                        unsigned m1 = pkmax(SIMON_LO(Rq.y, 4), SIMON_HI(Rq.y, 4));
                        unsigned m2 = pkmax(SIMON_LO(Rq.z, 8), SIMON_HI(Rq.z, 8));
                        unsigned m3 = pkmax(SIMON_LO(Rq.w, 12), SIMON_HI(Rq.w, 12));
#undef SIMON_LO
#undef SIMON_HI
                        m0 = pkmax(pkmax(m0, m1), pkmax(m2, m3));
                        const unsigned m16 = max(m0 & 0xFFFFu, m0 >> 16);
                        const unsigned M = m16 >> 4;                   // 0: no feasible node here; else 1 + LA + BA
                        const int p = b * 16 + 15 - (int)(m16 & 15u);
                        const unsigned canon = s_canon[b < nblk ? p : 0];
                        const int sn2 = *(const int*)((const char*)snrow + ncl4[q]);
                        // total = BA + LA + Simon + GpuShare (weights: registry.go:118-131, utils.go:321-333)
                        const unsigned kq = ((M - 1u + (unsigned)sn2) << 22) | ((2047u - canon) << 11) | (unsigned)p;
                        key = max(key, (M != 0u && b < nblk) ? kq : 0u);
                    }
                }
                key = wave_max_u32(key);
                pstar = (int)(key & 2047u);
                res = (int)(2047u - ((key >> 11) & 2047u));           // first maximum in canonical order
            }
        }
        // -------- assume: NodeInfo.AddPod (V/framework/types.go:482-508) + table column ---------
        if (pstar >= 0) {
            const SigRow& cur_sig = s_sig[r_sig];
            const unsigned add_c = (unsigned)cur_sig.req_c, add_m = (unsigned)cur_sig.req_m;
            const int blk = pstar >> 4, owner = blk & 63, pos = pstar & 15;
            const unsigned old = (sc.ablate & 1) ? 1u : my_col[pstar]; // this signature's byte before the cycle
            unsigned st_c, st_m, shape_id, dcls;
            int st_f;
            double nzc = 0.0, nzm = 0.0;
            if (REGSTATE) {
                unsigned v_c = 0, v_m = 0, v_f = 0;
                const int t_idx = (blk >> 6) * 16 + pos;
#define SIMON_CASE(t)                                                                                       \
    case (t):                                                                                               \
        v_c = (unsigned)__builtin_amdgcn_readlane((int)rqc[(t) < SLOTS * 16 ? (t) : 0], owner) + add_c;     \
        v_m = (unsigned)__builtin_amdgcn_readlane((int)rqm[(t) < SLOTS * 16 ? (t) : 0], owner) + add_m;     \
        v_f = (unsigned)__builtin_amdgcn_readlane((int)fps[(t) < SLOTS * 16 ? (t) : 0], owner) - 256u;      \
        if (lane == owner) { rqc[(t) < SLOTS * 16 ? (t) : 0] = v_c; rqm[(t) < SLOTS * 16 ? (t) : 0] = v_m; fps[(t) < SLOTS * 16 ? (t) : 0] = v_f; } \
        break;
                switch (t_idx) {
                    SIMON_ST16(SIMON_CASE, 0)
                    SIMON_ST16(SIMON_CASE, 16)
                    default: break;
                }
#undef SIMON_CASE
                st_c = v_c; st_m = v_m;
                st_f = (int)v_f >> 8;                                  // |free pod slots| + P < 2^22 checked on the host
                shape_id = v_f & 0xFFu;
                dcls = (unsigned)__builtin_amdgcn_readlane(ncl4[0], owner) >> 2;
                if (SLOTS > 1 && (blk >> 6)) dcls = (unsigned)__builtin_amdgcn_readlane(ncl4[SLOTS - 1], owner) >> 2;
            } else {
                uint4 st = (sc.ablate & 4) ? make_uint4(1, 1, 50, 0) : s_state[pstar];
                st.x += add_c;
                st.y += add_m;
                st.z -= 1u;
                if (!NZEQ) {
                    uint2 z = s_nz[pstar];
                    z.x += (unsigned)cur_sig.nz_c;
                    z.y += (unsigned)cur_sig.nz_m;
                    if (lane == 0) s_nz[pstar] = z;
                    nzc = (double)z.x; nzm = (double)z.y;
                }
                if (lane == 0 && !(sc.ablate & 4)) s_state[pstar] = st;
                st_c = st.x; st_m = st.y; st_f = (int)st.z; shape_id = st.w & 0xFFFFu; dcls = st.w >> 16;
            }
            const ShapeRow sh = s_shape[shape_id];
            const unsigned nb_raw = (sc.ablate & 16) ? 100u : eval_node(my_req_c, my_req_m, my_nz_c, my_nz_m, my_zero, (double)st_c,
                                                                        (double)st_m, nzc, nzm, st_f, sh);
            // -------- patch the prefetched row of the next pod (its load preceded this cycle's stores) ----
            if (i + 1 < P) {
                const int dwi = pos >> 2, sh8 = (pos & 3) * 8, qq = SLOTS > 1 ? (blk >> 6) : 0;
                unsigned w = dwi == 0 ? Rn[0].x : dwi == 1 ? Rn[0].y : dwi == 2 ? Rn[0].z : Rn[0].w;
                if (SLOTS > 1) {
                    const unsigned w1 = dwi == 0 ? Rn[SLOTS - 1].x : dwi == 1 ? Rn[SLOTS - 1].y : dwi == 2 ? Rn[SLOTS - 1].z : Rn[SLOTS - 1].w;
                    w = qq ? w1 : w;
                }
                const unsigned wo = (unsigned)__builtin_amdgcn_readlane((int)w, owner);
                const unsigned old_n = (wo >> sh8) & 0xFFu;
                const unsigned new_n = old_n ? (unsigned)__builtin_amdgcn_readlane((int)nb_raw, k_next) : 0u;
                const unsigned wn = (wo & ~(0xFFu << sh8)) | (new_n << sh8);
                const bool mine = lane == owner;
                switch (qq * 4 + dwi) {
                    case 0: Rn[0].x = mine ? wn : Rn[0].x; break;
                    case 1: Rn[0].y = mine ? wn : Rn[0].y; break;
                    case 2: Rn[0].z = mine ? wn : Rn[0].z; break;
                    case 3: Rn[0].w = mine ? wn : Rn[0].w; break;
                    case 4: Rn[SLOTS - 1].x = mine ? wn : Rn[SLOTS - 1].x; break;
                    case 5: Rn[SLOTS - 1].y = mine ? wn : Rn[SLOTS - 1].y; break;
                    case 6: Rn[SLOTS - 1].z = mine ? wn : Rn[SLOTS - 1].z; break;
                    default: Rn[SLOTS - 1].w = mine ? wn : Rn[SLOTS - 1].w; break;
                }
            }
            // -------- table column + feasible-node counters (off the critical path) -------------------
            const unsigned nb = old ? nb_raw : 0u;                     // static mask / monotone infeasibility
            if (lane < K && nb != old && !(sc.ablate & 2)) {
                my_col[pstar] = (unsigned char)nb;
                if (!nb) s_cnt[dcls * 64 + lane] -= 1;
            }
#ifdef SIMON_CACHE_DRAIN_STORES
            if (GLOBAL) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
            __builtin_amdgcn_wave_barrier();
        }
        // -------- placement, recorded by STEP (coalesced); simon_hip.hip permutes to pod ids ---
        plreg = (il == lane) ? res : plreg;
        if (place && il == 63) place[(i & ~63) + lane] = plreg;
    }
    if (place && (P & 63) && lane < (P & 63)) place[(P & ~63) + lane] = plreg;

    // ---- epilogue: sum of Requested over the scenario's nodes -------------------------------
    long long uc = 0, um = 0;
    if (REGSTATE) {
#pragma unroll
        for (int t = 0; t < SLOTS * 16; ++t) { uc += rqc[t]; um += rqm[t]; }
    } else {
        for (int p = lane; p < ni; p += 64) { const uint4 st = s_state[p]; uc += st.x; um += st.y; }
    }
    uc = wave_sum_i64(uc);
    um = wave_sum_i64(um);
    if (lane == 0) {
        unscheduled[s] = unsched;
        used_cpu[s] = uc * (long long)sc.g_cpu;
        used_mem[s] = um * (long long)sc.g_mem;
    }
}

// placement[s][pod] = place_step[s][inv_order[order_id(s)][pod]]: gather (scattered reads hit L2,
// stores coalesced)
__global__ __launch_bounds__(256) void unpermute_kernel(const int32_t* __restrict__ place_step,
                                                        const int32_t* __restrict__ inv_orders,
                                                        const ScenarioDesc* __restrict__ scen, int P,
                                                        int32_t* __restrict__ placement) {
    const int s = blockIdx.y;
    const int32_t* __restrict__ inv = inv_orders + (size_t)scen[s].order_id * P;
    const int32_t* __restrict__ src = place_step + (size_t)s * P;
    int32_t* __restrict__ dst = placement + (size_t)s * P;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) dst[p] = src[inv[p]];
}

template <int SLOTS, bool M, bool Z, bool G, bool R>
static hipError_t launch_smz(const CacheLaunch& a, int n_blocks, size_t lds, hipStream_t st) {
    auto kern = cache_kernel<SLOTS, M, Z, G, R>;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(n_blocks), dim3(64), lds, st, a.ncls, a.rank, a.shape_of, a.a_pods, a.i_rq_cpu,
                       a.i_rq_mem, a.i_nz_cpu, a.i_nz_mem, a.i_npods, a.clsprefix, a.sigs, a.shapes, a.pods, a.orders,
                       a.scen, a.perm, a.static_mask, a.simon_raw, a.unscheduled, a.used_cpu, a.used_mem, a.place_step,
                       a.ws, a.sc);
    return hipGetLastError();
}

template <int SLOTS, bool G>
static hipError_t launch_s(const CacheLaunch& a, int n_blocks, bool m, bool z, size_t lds, hipStream_t st) {
    if (G && z && a.reg_state) {   // node state in registers: NonZeroRequested == Requested, HBM workspace for the table
        return m ? launch_smz<SLOTS, true, true, G, G>(a, n_blocks, lds, st) : launch_smz<SLOTS, false, true, G, G>(a, n_blocks, lds, st);
    }
    if (m) return z ? launch_smz<SLOTS, true, true, G, false>(a, n_blocks, lds, st) : launch_smz<SLOTS, true, false, G, false>(a, n_blocks, lds, st);
    return z ? launch_smz<SLOTS, false, true, G, false>(a, n_blocks, lds, st) : launch_smz<SLOTS, false, false, G, false>(a, n_blocks, lds, st);
}

size_t cache_lds_bytes(int K, int stride, int ni_max, int Cn, int Cp, int n_shapes, bool nzeq, bool global) {
    return (size_t)carve(K, stride, ni_max, Cn, Cp, n_shapes, nzeq, global).total;
}
size_t cache_ws_bytes(int K, int stride, int ni_max, int Cn, int Cp, int n_shapes, bool nzeq) {
    return (size_t)carve(K, stride, ni_max, Cn, Cp, n_shapes, nzeq, true).ws_total;
}

// n_blocks scenarios; scenario of block b = a.perm[b]; a.ws != nullptr selects the HBM-workspace variant
hipError_t launch_cache(const CacheLaunch& a, int n_blocks, bool has_mask, bool nzeq, size_t lds_bytes, hipStream_t st) {
    const int slots = (a.sc.ni_max / 16 + 63) / 64;
    const bool g = a.ws != nullptr;
    switch (slots) {
        case 1: return g ? launch_s<1, true>(a, n_blocks, has_mask, nzeq, lds_bytes, st) : launch_s<1, false>(a, n_blocks, has_mask, nzeq, lds_bytes, st);
        case 2: return g ? launch_s<2, true>(a, n_blocks, has_mask, nzeq, lds_bytes, st) : launch_s<2, false>(a, n_blocks, has_mask, nzeq, lds_bytes, st);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_unpermute(const int32_t* place_step, const int32_t* inv_orders, const ScenarioDesc* scen, int S, int P,
                            int32_t* placement, hipStream_t st) {
    if (S <= 0 || P <= 0) return hipSuccess;
    const int bx = std::min(64, (P + 255) / 256);
    for (int s0 = 0; s0 < S; s0 += 65535) {   // gridDim.y limit
        const int ns = std::min(65535, S - s0);
        hipLaunchKernelGGL(unpermute_kernel, dim3(bx, ns), dim3(256), 0, st, place_step + (size_t)s0 * P, inv_orders,
                           scen + s0, P, placement + (size_t)s0 * P);
    }
    return hipGetLastError();
}

}  // namespace simon
