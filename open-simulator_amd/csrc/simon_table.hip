// simon_table.hip -- cpu+memory scenario kernel, generation 4: one WAVE = one capacity-planning scenario, a
// PRE-KEYED (signature, node) score table in canonical node order.
//
// What generation 3 (simon_cache.hip) established: a scheduling cycle changes ONE node, pods come from K distinct
// request signatures, so (feasible, score) of every (signature, node) is a table of which one column changes per
// cycle, and a per-(signature, block of 16 nodes) summary in LDS answers findNodesThatFitPod + prioritizeNodes +
// selectHost (V/core/generic_scheduler.go:131-209) with one LDS read per 16 nodes.  Its cycle, measured
// (profiles/README.md): ~250 dependent instructions and five waits (feasible-class counters, summary row, canonical
// index of the block's best node, tile row + node state from L2, node shape) -- 4 500 cycles of latency per pod.
//
// This generation removes three of the waits and a third of the instructions by changing WHAT the table holds:
//   * an entry is the final arg-max key of its node for its signature, ready to be max-reduced:
//       e = (1 + LeastAllocated + BalancedAllocation + 2 x normalised Simon score) << 4 | (15 - position in block)
//     (0 = NodeResourcesFit or a static filter fails).  The Simon / Open-Gpu-Share term (pkg/simulator/plugin/simon.go:
//     76-101: min-max normalised over the FEASIBLE nodes) depends on (signature, node class) and on which node classes
//     still hold a feasible node for the signature.  That set only ever shrinks (Requested grows, pod slots shrink), so
//     the term is FOLDED into the entries and a signature's row is re-based in place on the rare cycle after a class
//     lost its last feasible node (at most Cn - 1 times per signature and scenario);
//   * nodes stay in canonical nodeTree order (position = node index): the first maximum in canonical order is the
//     maximum of (e >> 4) << 12 | (4095 - position) -- no class-major permutation, no canonical-index lookup, no
//     per-class term in the scan;
//   * the block summary is the plain packed-u16 maximum of the 16 pre-keyed entries (no unpacking of bytes).
// Per pod: [dirty signature? re-base its row] -> one u16 LDS read per 16 nodes -> 4 VALU -> one DPP wave max -> node.
// assume (V/scheduler.go:371 -> NodeInfo.AddPod, V/framework/types.go:482-508): the wave loads the node's 16-byte state
// and lane k the 32-byte table row (signature k, touched block) in ONE memory round trip, overlapped with the LDS reads
// of the node's shape and class term; lane k re-evaluates signature k with exactly the fp64 sequences of
// simon_fast.hip / simon_cache.hip, patches its entry, re-reduces its row and stores the summary entry.
//
// Limits: K <= 128 signatures (two per lane), <= 4095 nodes, <= 64 node classes, <= 256 node shapes, NARROW
// preconditions (simon_hip.hip::choose_variant), no zero-capacity node, orders are permutations.
#include "simon_table.h"

#include <algorithm>

namespace simon {

typedef unsigned int u32x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x2t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pkmax_t(unsigned a, unsigned b) {
    u16x2t x = __builtin_bit_cast(u16x2t, a), y = __builtin_bit_cast(u16x2t, b);
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(x, y));
}

__device__ __forceinline__ int la_term_t(double r, double rc100) {   // == la_term_f of simon_fast.hip (exactness: DESIGN.md 5.2)
    const double C = 100.0 + 0x1.0p-33;
    return (int)__builtin_fma(-r, rc100, C);
}

// max over each 16-lane row (result in every lane of the row)
__device__ __forceinline__ unsigned row16_max_t(unsigned v) {
    v = max(v, (unsigned)SIMON_DPP(0, (int)v, 0xB1, 0xF));
    v = max(v, (unsigned)SIMON_DPP(0, (int)v, 0x4E, 0xF));
    v = max(v, (unsigned)SIMON_DPP(0, (int)v, 0x141, 0xF));
    v = max(v, (unsigned)SIMON_DPP(0, (int)v, 0x140, 0xF));
    return v;
}

struct TCarve {
    int sum, node, sn, cnt, raw, shape, tmp, total;   // LDS offsets (multiples of 16)
    int ws_tile, ws_state, ws_nz, ws_total;           // HBM workspace offsets per scenario
    int nbp;
};
__host__ __device__ inline TCarve tcarve(int K, int ni_max, int Cn, int Cp, int n_shapes, bool nzeq) {
    auto al = [](int x) { return (x + 15) & ~15; };
    TCarve c;
    const int nblk = ni_max / 16;
    c.nbp = cache_nbp(nblk);
    int o = 0;
    c.sum = o; o += al(K * c.nbp * 2);
    c.node = o; o += al(ni_max * 2);
    c.sn = o; o += al(K * Cn * 2);
    c.cnt = o; o += al(K * Cn * 4);
    c.raw = o; o += al(Cp * Cn * 4);
    c.shape = o; o += n_shapes * 48;
    c.tmp = o; o += 64 * 4;
    c.total = o;
    int w = 0;
    c.ws_tile = w; w += (nblk * K * 32 + 127) & ~127;
    c.ws_state = w; w += (ni_max * 16 + 127) & ~127;
    c.ws_nz = w; w += nzeq ? 0 : ((ni_max * 8 + 127) & ~127);
    c.ws_total = w;
    return c;
}

// KQ: signatures per lane (1: K <= 64, 2: K <= 128).  HAS_PIN: the stream holds pinned pods (own instantiation: the extra
// branch costs the common kernel time).
template <bool HAS_MASK, bool NZEQ, bool HAS_PIN, int KQ>
__global__ __launch_bounds__(64) void table_kernel(
    const int32_t* __restrict__ ncls, const int32_t* __restrict__ shape_of, const int32_t* __restrict__ a_pods,
    const uint32_t* __restrict__ i_rq_cpu, const uint32_t* __restrict__ i_rq_mem, const uint32_t* __restrict__ i_nz_cpu,
    const uint32_t* __restrict__ i_nz_mem, const int32_t* __restrict__ i_npods, const SigRow* __restrict__ sigs,
    const ShapeRow* __restrict__ shapes, const PodRowC* __restrict__ pods, const int32_t* __restrict__ orders,
    const ScenarioDesc* __restrict__ scen, const int32_t* __restrict__ perm, const uint64_t* __restrict__ static_mask,
    const int32_t* __restrict__ simon_raw, int32_t* __restrict__ unscheduled, int64_t* __restrict__ used_cpu,
    int64_t* __restrict__ used_mem, int32_t* __restrict__ place_step, unsigned char* ws, const TableScalars sc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int Cn = sc.Cn, Cp = sc.Cp, P = sc.P, K = sc.K;
    const TCarve cv = tcarve(K, sc.ni_max, Cn, Cp, sc.n_shapes, NZEQ);
    const int nbp = cv.nbp;
    unsigned char* const wsb = ws + (size_t)blockIdx.x * (size_t)cv.ws_total;
    unsigned short* g_tile = (unsigned short*)(wsb + cv.ws_tile);   // [block][K][16] pre-keyed entries
    uint4* g_state = (uint4*)(wsb + cv.ws_state);                   // {Requested cpu, mem, free pod slots, unused}
    uint2* g_nz = (uint2*)(wsb + cv.ws_nz);                         // NonZeroRequested {cpu, mem} (only when !NZEQ)
    unsigned short* s_sum = (unsigned short*)(smem + cv.sum);       // [K][nbp]: max entry of (signature, block)
    unsigned short* s_node = (unsigned short*)(smem + cv.node);     // [ni]: shape id | node class << 8
    unsigned short* s_sn = (unsigned short*)(smem + cv.sn);         // [K][Cn]: the class term currently folded into row k
    int* s_cnt = (int*)(smem + cv.cnt);                             // [K][Cn]: feasible nodes of class d for signature k
    int* s_raw = (int*)(smem + cv.raw);                             // [Cp][Cn] Simon raw scores
    const ShapeRow* s_shape = (const ShapeRow*)(smem + cv.shape);
    int* s_tmp = (int*)(smem + cv.tmp);

    const int lane = threadIdx.x;
    const int s = perm[blockIdx.x];
    const int n = scen[s].n_nodes;
    const int ni = (n + 15) & ~15, nblk = ni >> 4;
    const int32_t* __restrict__ order = orders + (size_t)scen[s].order_id * P;

    // ---- prologue 1: clear, tables -> LDS ----------------------------------------------------
    for (int i = lane; i < cv.node / 16; i += 64) ((uint4*)smem)[i] = make_uint4(0, 0, 0, 0);    // summary
    for (int i = lane; i < K * Cn; i += 64) { s_cnt[i] = 0; s_sn[i] = 0; }
    for (int i = lane; i < Cp * Cn; i += 64) s_raw[i] = simon_raw[i];
    for (int i = lane; i < sc.n_shapes * 12; i += 64) ((int*)(smem + cv.shape))[i] = ((const int*)shapes)[i];
    for (int p = lane; p < ni; p += 64) s_node[p] = p < n ? (unsigned short)((unsigned)shape_of[p] | ((unsigned)ncls[p] << 8)) : (unsigned short)0;
    __syncthreads();

    // (feasible, LeastAllocated + BalancedAllocation) of one signature on one node: 0 when NodeResourcesFit fails
    // (fit.go:230-302), else 1 + score.  Same fp64 sequences as simon_fast.hip::eval_slot / simon_cache.hip.
    auto eval_node = [&](double q_req_c, double q_req_m, double q_nz_c, double q_nz_m, bool q_zero, double rq_c, double rq_m,
                         double nzs_c, double nzs_m, int freep, const ShapeRow& sh) -> unsigned {
        const double t_c = rq_c + q_req_c, t_m = rq_m + q_req_m;
        const bool res_ok = (sh.cap_c >= t_c) && (sh.cap_m >= t_m);
        const bool ok = (freep >= 1) && (q_zero || res_ok);
        // resource_allocation.go:91-98: requested = NonZeroRequested + pod non-zero request
        const double r_c = NZEQ ? t_c : nzs_c + q_nz_c;
        const double r_m = NZEQ ? t_m : nzs_m + q_nz_m;
        const bool ge_c = r_c >= sh.cap_c, ge_m = r_m >= sh.cap_m;
        const int la_c = ge_c ? 0 : la_term_t(r_c, sh.rc100_c);           // least_allocated.go:108-117
        const int la_m = ge_m ? 0 : la_term_t(r_m, sh.rc100_m);
        const double cf = div_by_rcp(r_c, sh.cap_c, sh.rc_c);             // balanced_allocation.go:82-119
        const double mf = div_by_rcp(r_m, sh.cap_m, sh.rc_m);
        const int bs = (int)((1.0 - __builtin_fabs(cf - mf)) * 100.0);
        const int base = ((la_c + la_m) >> 1) + ((ge_c || ge_m) ? 0 : bs);
        return ok ? (unsigned)(base + 1) : 0u;
    };

    // ---- prologue 2: node rows, table and summary; lanes = 64 consecutive nodes (4 blocks) --------
    for (int p0 = 0; p0 < ni; p0 += 64) {
        const int p = p0 + lane;
        const bool real = p < n;
        const int j = real ? p : 0;
        const unsigned nd = real ? (unsigned)s_node[p] : 0u;
        const uint4 st = real ? make_uint4(i_rq_cpu[j], i_rq_mem[j], (unsigned)(a_pods[j] - i_npods[j]), 0u) : make_uint4(0, 0, 0, 0);
        uint2 z = make_uint2(0, 0);
        if (!NZEQ && real) z = make_uint2(i_nz_cpu[j], i_nz_mem[j]);
        if (p < ni) {
            g_state[p] = st;
            if (!NZEQ) g_nz[p] = z;
        }
        const ShapeRow sh = s_shape[nd & 0xFFu];
        unsigned short* tp = g_tile + ((size_t)(p >> 4) * K) * 16 + (p & 15);
        for (int k = 0; k < K; ++k) {
            const SigRow q = sigs[k];
            unsigned b = eval_node(q.req_c, q.req_m, q.nz_c, q.nz_m, q.flags & 1u, (double)st.x, (double)st.y, (double)z.x,
                                   (double)z.y, (int)st.z, sh);
            b = real ? b : 0u;
            if (HAS_MASK) {   // NodeUnschedulable/NodeName/TaintToleration/NodeAffinity: static per (class, node)
                const uint64_t w = static_mask[(size_t)q.cls * sc.mask_words + (j >> 6)];
                b = ((w >> (j & 63)) & 1ull) ? b : 0u;
            }
            const unsigned e = b ? ((b << 4) | (unsigned)(15 - (p & 15))) : 0u;   // class term still 0: every signature starts dirty
            if (p < ni) tp[k * 16] = (unsigned short)e;
            const unsigned m16 = row16_max_t(e);
            if ((lane & 15) == 0 && p < ni) s_sum[k * nbp + (p >> 4)] = (unsigned short)m16;
            if (b) atomicAdd(&s_cnt[k * Cn + (int)(nd >> 8)], 1);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // this lane's signatures: lane l re-evaluates signatures l (and l + 64) on a touched node
    int kk[KQ];
    bool kvalid[KQ];
    double my_req_c[KQ], my_req_m[KQ], my_nz_c[KQ], my_nz_m[KQ];
    bool my_zero[KQ];
    unsigned my_add_c[KQ], my_add_m[KQ], my_addz_c[KQ], my_addz_m[KQ];
    unsigned my_dirty = 0;
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
        kvalid[q] = lane + 64 * q < K;
        kk[q] = kvalid[q] ? lane + 64 * q : 0;
        const SigRow r = sigs[kk[q]];
        my_req_c[q] = r.req_c; my_req_m[q] = r.req_m; my_nz_c[q] = r.nz_c; my_nz_m[q] = r.nz_m;
        my_zero[q] = r.flags & 1u;
        my_add_c[q] = (unsigned)r.req_c; my_add_m[q] = (unsigned)r.req_m;
        my_addz_c[q] = (unsigned)r.nz_c; my_addz_m[q] = (unsigned)r.nz_m;
        if (kvalid[q]) my_dirty |= 1u << q;
    }
    // per-lane constant of the arg-max key: lane handles blocks lane, lane + 64, ... ; key low field = 4095 - position
    int cb[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) cb[q] = 4080 - 16 * (q * 64 + lane);
    unsigned koff[KQ];
#pragma unroll
    for (int q = 0; q < KQ; ++q) koff[q] = (unsigned)kk[q] * 32u;

    // Re-base row k after the set of node classes with a feasible node changed: SimonPlugin / GpuSharePlugin
    // NormalizeScore (pkg/simulator/plugin/simon.go:76-101) over the classes present, x 2 (both plugins, weight 1 each).
    auto renormalise = [&](int k, int c) {
        const int dd = lane < Cn ? lane : 0;
        const int cn = (lane < Cn) ? s_cnt[k * Cn + dd] : 0;
        const bool inb = cn > 0;
        const int rawc = s_raw[c * Cn + dd];
        const int lo = wave_min_i32(inb ? rawc : 0x7fffffff);
        const int hi = wave_max_i32(inb ? rawc : (int)0x80000000);
        const bool any = hi >= lo;
        const int range = any ? hi - lo : 0;
        const double rr = range ? 1.0 / (double)range : 0.0;
        const int sn = (inb && range) ? 2 * (int)__builtin_fma((double)(rawc - lo) * 100.0, rr, 0.5 * rr) : 0;
        if (lane < Cn) {
            s_tmp[dd] = sn - (int)s_sn[k * Cn + dd];
            s_sn[k * Cn + dd] = (unsigned short)sn;
        }
        __syncthreads();
        for (int p0 = 0; p0 < ni; p0 += 64) {
            const int p = p0 + lane;
            const bool in = p < ni;
            unsigned short* ep = g_tile + ((size_t)((in ? p : 0) >> 4) * K + k) * 16 + (p & 15);
            unsigned e = in ? (unsigned)*ep : 0u;
            if (e) {
                e = (unsigned)((int)e + s_tmp[s_node[p] >> 8] * 16);
                *ep = (unsigned short)e;
            }
            const unsigned m16 = row16_max_t(e);
            if ((lane & 15) == 0 && in) s_sum[k * nbp + (p >> 4)] = (unsigned short)m16;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };

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

    for (int i = 0; i < P; ++i) {
        const int il = i & 63;
        if (il == 0) { cur = nxt; nxt = load_chunk(i + 64); }
        const int r_sig = __builtin_amdgcn_readlane(cur.x, il), r_preset = __builtin_amdgcn_readlane(cur.y, il);
        const int r_gate = __builtin_amdgcn_readlane(cur.z, il), r_cls = __builtin_amdgcn_readlane(cur.w, il);

        int res, pstar = -1;
        if (r_gate >= n) {
            res = -2;                                                  // pod not part of this scenario
        } else if (r_preset >= 0) {                                    // addPodToCache path (V/eventhandlers.go:223-236)
            res = r_preset;
            pstar = r_preset;
        } else if (HAS_PIN && r_preset <= -2) {                        // pinned pod (simon_pods_soa.pin_node, stored as -2 - node):
            const int pin = -2 - r_preset;                             // its node affinity admits ONE node; the table entry of
            res = -1;                                                  // (signature, node) holds static filters + fit
            if (pin < n) {
                const unsigned short e = g_tile[((size_t)(pin >> 4) * K + r_sig) * 16 + (pin & 15)];
                if (e != 0) { res = pin; pstar = pin; }
            }
            if (res < 0) ++unsched;
        } else {
            const int k = r_sig;
            const unsigned dq = (unsigned)__builtin_amdgcn_readlane((int)my_dirty, k & 63);
            if ((dq >> (k >> 6)) & 1u) {                               // a node class lost its last feasible node for k (or first use)
                renormalise(k, r_cls);
                if (lane == (k & 63)) my_dirty &= ~(1u << (k >> 6));
            }
            // -------- summary scan: one pre-keyed u16 per block of 16 nodes ---------------------------
            const unsigned short* srow = s_sum + k * nbp;
            unsigned key = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (q == 0 || nblk > 64 * q) {
                    const int b = q * 64 + lane;
                    const bool in = b < nblk;
                    const unsigned m16 = srow[in ? b : 0];
                    // (total + 1) << 12 | (4095 - position), position = 16 b + 15 - (m16 & 15)
                    const unsigned kq = ((m16 >> 4) << 12) + (m16 & 15u) + (unsigned)cb[q];
                    key = max(key, (m16 != 0u && in) ? kq : 0u);
                }
            }
            key = wave_max_u32(key);
            if (key == 0u) {                                           // FitError: pod deleted, state unchanged
                ++unsched;
                res = -1;
            } else {
                pstar = 4095 - (int)(key & 4095u);                     // first maximum in canonical order
                res = pstar;
            }
        }
        // -------- assume: NodeInfo.AddPod (V/framework/types.go:482-508) + table column + summary ----
        if (pstar >= 0) {
            const int blk = pstar >> 4, pos = pstar & 15;
            const int dwi = pos >> 1, sh16 = (pos & 1) * 16;
            uint4 st = g_state[pstar];
            // row of (signature kk[q], touched block): uniform block base + this lane's constant byte offset (saddr + voffset)
            const unsigned char* const blk_base = (const unsigned char*)g_tile + (size_t)blk * (size_t)(K * 32);
            const uint4* rowp[KQ];
            uint4 A[KQ], B[KQ];
#pragma unroll
            for (int q = 0; q < KQ; ++q) {
                rowp[q] = (const uint4*)(blk_base + koff[q]);
                A[q] = rowp[q][0];
                B[q] = rowp[q][1];
            }
            uint2 z = make_uint2(0, 0);
            if (!NZEQ) z = g_nz[pstar];
            const unsigned nd = s_node[pstar];
            const int dcls = (int)(nd >> 8);
            const ShapeRow sh = s_shape[nd & 0xFFu];
            unsigned snq[KQ];
#pragma unroll
            for (int q = 0; q < KQ; ++q) snq[q] = s_sn[kk[q] * Cn + dcls];
            const int sl = r_sig & 63;
            const bool hiq = KQ > 1 && (r_sig >> 6);
            st.x += (unsigned)(hiq ? __builtin_amdgcn_readlane((int)my_add_c[KQ - 1], sl) : __builtin_amdgcn_readlane((int)my_add_c[0], sl));
            st.y += (unsigned)(hiq ? __builtin_amdgcn_readlane((int)my_add_m[KQ - 1], sl) : __builtin_amdgcn_readlane((int)my_add_m[0], sl));
            st.z -= 1u;
            double nzc = 0.0, nzm = 0.0;
            if (!NZEQ) {
                z.x += (unsigned)(hiq ? __builtin_amdgcn_readlane((int)my_addz_c[KQ - 1], sl) : __builtin_amdgcn_readlane((int)my_addz_c[0], sl));
                z.y += (unsigned)(hiq ? __builtin_amdgcn_readlane((int)my_addz_m[KQ - 1], sl) : __builtin_amdgcn_readlane((int)my_addz_m[0], sl));
                if (lane == 0) g_nz[pstar] = z;
                nzc = (double)z.x; nzm = (double)z.y;
            }
            if (lane == 0) g_state[pstar] = st;
            const double rq_c = (double)st.x, rq_m = (double)st.y;
#pragma unroll
            for (int q = 0; q < KQ; ++q) {
                const unsigned nb_raw = eval_node(my_req_c[q], my_req_m[q], my_nz_c[q], my_nz_m[q], my_zero[q], rq_c, rq_m, nzc, nzm,
                                                  (int)st.z, sh);
                u32x8 T = {A[q].x, A[q].y, A[q].z, A[q].w, B[q].x, B[q].y, B[q].z, B[q].w};
                const unsigned wsel = T[dwi];
                const unsigned old16 = (wsel >> sh16) & 0xFFFFu;          // this signature's entry before the cycle
                const unsigned nb = old16 ? nb_raw : 0u;                  // static mask / monotone infeasibility
                const unsigned new16 = nb ? (((nb + snq[q]) << 4) | (unsigned)(15 - pos)) : 0u;
                // One wave: its vector memory accesses are served in order, so the next cycle's loads of this row / state
                // observe these stores; no cache maintenance, no wait.
                if (kvalid[q] && new16 != old16) {
                    ((unsigned short*)rowp[q])[pos] = (unsigned short)new16;
                    T[dwi] = (wsel & ~(0xFFFFu << sh16)) | (new16 << sh16);
                    unsigned m = pkmax_t(pkmax_t(pkmax_t(T[0], T[1]), pkmax_t(T[2], T[3])), pkmax_t(pkmax_t(T[4], T[5]), pkmax_t(T[6], T[7])));
                    m = max(m & 0xFFFFu, m >> 16);
                    s_sum[kk[q] * nbp + blk] = (unsigned short)m;
                    if (!nb) {                                            // the node stopped being feasible for this signature
                        const int cidx = kk[q] * Cn + dcls;
                        const int left = s_cnt[cidx] - 1;
                        s_cnt[cidx] = left;
                        if (left == 0) my_dirty |= 1u << q;               // the class term of row k changes: re-base before its next use
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        // -------- placement, recorded by STEP (coalesced); simon_hip.hip permutes to pod ids ---
        plreg = (il == lane) ? res : plreg;
        if (place && il == 63) place[(i & ~63) + lane] = plreg;
    }
    if (place && (P & 63) && lane < (P & 63)) place[(P & ~63) + lane] = plreg;

    // ---- epilogue: sum of Requested over the scenario's nodes -------------------------------
    long long uc = 0, um = 0;
    for (int p = lane; p < n; p += 64) { const uint4 st = g_state[p]; uc += st.x; um += st.y; }
    uc = wave_sum_i64(uc);
    um = wave_sum_i64(um);
    if (lane == 0) {
        unscheduled[s] = unsched;
        used_cpu[s] = uc * (long long)sc.g_cpu;
        used_mem[s] = um * (long long)sc.g_mem;
    }
}

template <bool M, bool Z, bool PIN, int KQ>
static hipError_t launch_t4(const TableLaunch& a, int n_blocks, size_t lds, hipStream_t st) {
    auto kern = table_kernel<M, Z, PIN, KQ>;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(n_blocks), dim3(64), lds, st, a.ncls, a.shape_of, a.a_pods, a.i_rq_cpu, a.i_rq_mem,
                       a.i_nz_cpu, a.i_nz_mem, a.i_npods, a.sigs, a.shapes, a.pods, a.orders, a.scen, a.perm,
                       a.static_mask, a.simon_raw, a.unscheduled, a.used_cpu, a.used_mem, a.place_step, a.ws, a.sc);
    return hipGetLastError();
}

size_t table_lds_bytes(int K, int ni_max, int Cn, int Cp, int n_shapes, bool nzeq) {
    return (size_t)tcarve(K, ni_max, Cn, Cp, n_shapes, nzeq).total;
}
size_t table_ws_bytes(int K, int ni_max, int Cn, int Cp, int n_shapes, bool nzeq) {
    return (size_t)tcarve(K, ni_max, Cn, Cp, n_shapes, nzeq).ws_total;
}

template <bool M, bool Z, bool PIN>
static hipError_t launch_t3(const TableLaunch& a, int n_blocks, size_t lds, hipStream_t st) {
    return a.sc.K > 64 ? launch_t4<M, Z, PIN, 2>(a, n_blocks, lds, st) : launch_t4<M, Z, PIN, 1>(a, n_blocks, lds, st);
}
template <bool M, bool Z>
static hipError_t launch_t2(const TableLaunch& a, int n_blocks, bool has_pin, size_t lds, hipStream_t st) {
    return has_pin ? launch_t3<M, Z, true>(a, n_blocks, lds, st) : launch_t3<M, Z, false>(a, n_blocks, lds, st);
}

hipError_t launch_table(const TableLaunch& a, int n_blocks, bool has_mask, bool nzeq, bool has_pin, size_t lds_bytes, hipStream_t st) {
    if (has_mask) return nzeq ? launch_t2<true, true>(a, n_blocks, has_pin, lds_bytes, st) : launch_t2<true, false>(a, n_blocks, has_pin, lds_bytes, st);
    return nzeq ? launch_t2<false, true>(a, n_blocks, has_pin, lds_bytes, st) : launch_t2<false, false>(a, n_blocks, has_pin, lds_bytes, st);
}

}  // namespace simon
