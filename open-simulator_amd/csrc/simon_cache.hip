// simon_cache.hip -- NARROW kernel, third generation: one WAVE = one capacity-planning scenario,
// a two-level (signature, node) score table instead of re-evaluating every node for every pod.
//
// Observation ("further lever"): a scheduling cycle changes ONE node.  For a pod
// stream drawn from K <= 64 distinct request signatures (replicas of workloads), the pair
// (feasible?, LeastAllocated + BalancedAllocation) of every (signature, node) is a table of
// one-byte entries (0 = NodeResourcesFit fails, else 1 + score <= 201) of which only the K bytes
// of the touched node change per cycle.  Two levels:
//   * TILES in an HBM workspace (L2-resident): tile[block of 16 nodes][signature][16 bytes].  The
//     K bytes of one node are 16 bytes apart inside one <= 1 KiB tile, so the update of a cycle is
//     ONE coalesced 16-byte-per-lane load (lane k: signature k's 16 nodes of the touched block)
//     and one byte store per lane into the same few cache lines -- no scattered accesses;
//   * SUMMARY in LDS: sum[signature][block] = max over the block of (byte << 4 | 15 - position)
//     as u16, i.e. the best base score of the block and the FIRST node that reaches it.
// Per pod (findNodesThatFitPod + prioritizeNodes + selectHost, V/core/generic_scheduler.go:131-209):
// read one summary row (2 bytes per 16 nodes), add the normalised Simon/Open-Gpu-Share term of
// the block's node class (pkg/simulator/plugin/simon.go:76-101; nodes are laid out class-major,
// stable in canonical order, every class padded to 16, so a block has ONE class), build
// key = total << 22 | (2047 - canonical index) << 11 | position, one DPP wave max = first maximum
// in nodeTree.list() order (determinised selectHost).  assume (V/scheduler.go:371 ->
// NodeInfo.AddPod, V/framework/types.go:482-508): lane k re-evaluates signature k on the touched
// node with EXACTLY the fp64 sequences of simon_fast.hip, patches its 16-byte tile row, stores
// the byte, re-reduces the 16 bytes and stores the new summary entry.
// The per-signature count of feasible nodes per node class is maintained incrementally
// (feasibility only ever decreases: Requested grows, free pod slots shrink), which replaces the
// phase-A reduction of the earlier generations by one LDS read + ballot.
//
// No barriers (one wave).  Global traffic per cycle: one 16 B state row, <= 1 KiB of tile, K byte
// stores into that tile; placements leave as one coalesced 256 B store per 64 cycles.
//
// Eligibility (host, simon_hip.hip): NARROW preconditions + K <= 64 signatures, <= 256 node
// shapes, Cn <= 32, no zero-capacity node, padded scenario size <= 2032, orders are permutations.
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

// max over the 16 bytes of one tile row of (byte << 4 | 15 - position)
__device__ __forceinline__ unsigned block_key16(const uint4 R) {
    const unsigned M8 = 0x00FF00FFu;
#define SIMON_LO(w, q4) ((((w) & M8) << 4) | (unsigned)((15 - (q4)) | ((15 - ((q4) + 2)) << 16)))
#define SIMON_HI(w, q4) (((((w) >> 8) & M8) << 4) | (unsigned)((15 - ((q4) + 1)) | ((15 - ((q4) + 3)) << 16)))
    unsigned m0 = pkmax(SIMON_LO(R.x, 0), SIMON_HI(R.x, 0));
    unsigned m1 = pkmax(SIMON_LO(R.y, 4), SIMON_HI(R.y, 4));
    unsigned m2 = pkmax(SIMON_LO(R.z, 8), SIMON_HI(R.z, 8));
    unsigned m3 = pkmax(SIMON_LO(R.w, 12), SIMON_HI(R.w, 12));
#undef SIMON_LO
#undef SIMON_HI
    m0 = pkmax(pkmax(m0, m1), pkmax(m2, m3));
    return max(m0 & 0xFFFFu, m0 >> 16);
}

// max over each 16-lane row (result in every lane of the row)
__device__ __forceinline__ unsigned row16_max_u32(unsigned v) {
    v = max(v, (unsigned)SIMON_DPP(0, (int)v, 0xB1, 0xF));
    v = max(v, (unsigned)SIMON_DPP(0, (int)v, 0x4E, 0xF));
    v = max(v, (unsigned)SIMON_DPP(0, (int)v, 0x141, 0xF));
    v = max(v, (unsigned)SIMON_DPP(0, (int)v, 0x140, 0xF));
    return v;
}

struct Carve {
    int sum, canon, cnt, raw, sn2, shape, seg, tmp, total;   // LDS offsets (multiples of 16)
    int ws_tile, ws_state, ws_nz, ws_total;                  // HBM workspace offsets per scenario
    int nbp, Kp;
};
__host__ __device__ inline Carve carve(int K, int ni_max, int Cn, int Cp, int n_shapes, bool nzeq) {
    auto al = [](int x) { return (x + 15) & ~15; };
    Carve c;
    const int nblk = ni_max / 16;
    c.nbp = cache_nbp(nblk);
    c.Kp = (K + 3) & ~3;
    if (c.Kp < 4) c.Kp = 4;
    int o = 0;
    c.sum = o; o += al(K * c.nbp * 2);
    c.canon = o; o += al(ni_max * 2);
    c.cnt = o; o += Cn * 64 * 4;
    c.raw = o; o += al(Cp * Cn * 4);
    c.sn2 = o; o += al(Cp * Cn * 4);
    c.shape = o; o += n_shapes * 48;
    c.seg = o; o += 32 * 4;
    c.tmp = o; o += 32 * 4;
    c.total = o;
    int w = 0;
    c.ws_tile = w; w += (nblk * c.Kp * 16 + 127) & ~127;
    c.ws_state = w; w += (ni_max * 16 + 127) & ~127;
    c.ws_nz = w; w += nzeq ? 0 : ((ni_max * 8 + 127) & ~127);
    c.ws_total = w;
    return c;
}

// HAS_PIN: the stream holds pinned pods (pin_node); a separate instantiation because the extra branch costs the
// common kernel 2.6 % (same-box A/B)
template <bool HAS_MASK, bool NZEQ, bool HAS_PIN>
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
    const int Cn = sc.Cn, Cp = sc.Cp, P = sc.P, K = sc.K;
#ifdef SIMON_CACHE_ABLATION
    const int ablate = sc.ablate;      // timing experiments (env SIMON_CACHE_ABLATE): results are wrong when non-zero
#else
    constexpr int ablate = 0;          // the product build carries no experiment switches (they cost SGPRs in the hot loop)
#endif
    const Carve cv = carve(K, sc.ni_max, Cn, Cp, sc.n_shapes, NZEQ);
    const int nbp = cv.nbp, Kp = cv.Kp;
    unsigned char* const wsb = ws + (size_t)blockIdx.x * (size_t)cv.ws_total;
    unsigned char* g_tile = wsb + cv.ws_tile;                 // [block][Kp][16] bytes
    uint4* g_state = (uint4*)(wsb + cv.ws_state);             // {Requested cpu, mem, free pod slots, shape | class<<16}
    uint2* g_nz = (uint2*)(wsb + cv.ws_nz);                   // NonZeroRequested {cpu, mem} (only when !NZEQ)
    unsigned short* s_sum = (unsigned short*)(smem + cv.sum); // [K][nbp]
    unsigned short* s_canon = (unsigned short*)(smem + cv.canon);
    int* s_cnt = (int*)(smem + cv.cnt);                       // [Cn][64]: feasible nodes of class d for signature k
    int* s_raw = (int*)(smem + cv.raw);
    int* s_sn2 = (int*)(smem + cv.sn2);
    const ShapeRow* s_shape = (const ShapeRow*)(smem + cv.shape);
    int* s_seg = (int*)(smem + cv.seg);
    int* s_tmp = (int*)(smem + cv.tmp);

    const int lane = threadIdx.x;
    const int s = perm[blockIdx.x];
    const int n = scen[s].n_nodes;
    const int32_t* __restrict__ order = orders + (size_t)scen[s].order_id * P;

    // ---- prologue 1: clear, tables -> LDS ----------------------------------------------------
    for (int i = lane; i < cv.canon / 16; i += 64) ((uint4*)smem)[i] = make_uint4(0, 0, 0, 0);   // summary
    for (int i = lane; i < Cn * 64; i += 64) s_cnt[i] = 0;
    for (int i = lane; i < Cp * Cn; i += 64) s_raw[i] = simon_raw[i];
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
    const int nblk = ni >> 4;
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

    // ---- prologue 3: node rows, tiles and summary; lanes = 64 consecutive positions (4 blocks) ----
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
        if (p < ni) {
            g_state[p] = st;
            if (!NZEQ) g_nz[p] = z;
        }
        const ShapeRow sh = s_shape[st.w & 0xFFFFu];
        // classes present in this chunk form a contiguous range (class-major layout)
        const int dlo = wave_min_i32(real ? d : 0x7fffffff), dhi = wave_max_i32(real ? d : (int)0x80000000);
        unsigned char* tp = g_tile + ((size_t)(p >> 4) * Kp) * 16 + (p & 15);
        for (int k = 0; k < K; ++k) {
            const SigRow q = sigs[k];
            unsigned b = eval_node(q.req_c, q.req_m, q.nz_c, q.nz_m, q.flags & 1u, (double)st.x, (double)st.y, (double)z.x,
                                   (double)z.y, (int)st.z, sh);
            b = real ? b : 0u;
            if (HAS_MASK) {   // NodeUnschedulable/NodeName/TaintToleration/NodeAffinity: static per (class, node)
                const uint64_t w = static_mask[(size_t)q.cls * sc.mask_words + (j >> 6)];
                b = ((w >> (j & 63)) & 1ull) ? b : 0u;
            }
            if (p < ni) tp[k * 16] = (unsigned char)b;
            const unsigned m16 = row16_max_u32((b << 4) | (unsigned)(15 - (p & 15)));
            if ((lane & 15) == 0 && p < ni) s_sum[k * nbp + (p >> 4)] = (unsigned short)m16;
            for (int dd = dlo; dd <= dhi; ++dd) {
                const int c = __popcll(__ballot(b != 0u && d == dd));
                if (lane == 0 && c) s_cnt[dd * 64 + k] += c;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // this lane's signature (lane k re-evaluates signature k on a touched node)
    const int kk = lane < K ? lane : 0;
    const SigRow mysig = sigs[kk];
    const double my_req_c = mysig.req_c, my_req_m = mysig.req_m, my_nz_c = mysig.nz_c, my_nz_m = mysig.nz_m;
    const bool my_zero = mysig.flags & 1u;
    const unsigned my_add_c = (unsigned)mysig.req_c, my_add_m = (unsigned)mysig.req_m;
    const unsigned my_addz_c = (unsigned)mysig.nz_c, my_addz_m = (unsigned)mysig.nz_m;
    unsigned short* my_sum = s_sum + kk * nbp;

    // node class of this lane's two blocks (class segments are multiples of 16)
    int ncl4[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int b = q * 64 + lane;
        const unsigned cj0 = (b < nblk) ? s_canon[b * 16] : 0xFFFFu;
        ncl4[q] = (cj0 != 0xFFFFu) ? ncls[cj0] * 4 : 0;
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
            pstar = __builtin_amdgcn_readfirstlane(s_seg[ncls[r_preset]] + rank[r_preset]);
        } else if (HAS_PIN && r_preset <= -2) {                        // pinned pod (simon_pods_soa.pin_node, stored as -2 - node):
            const int pin = -2 - r_preset;                             // its node affinity admits ONE node; the table byte of
            res = -1;                                                  // (signature, node) holds static filters + fit
            if (pin < n) {
                const int pp = __builtin_amdgcn_readfirstlane(s_seg[ncls[pin]] + rank[pin]);
                const unsigned char byte = g_tile[((size_t)(pp >> 4) * Kp + r_sig) * 16 + (pp & 15)];
                if (byte != 0) { res = pin; pstar = pp; }
            }
            if (res < 0) ++unsched;
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
                // -------- summary scan: one u16 per block of 16 nodes, two blocks per lane ---------
                const unsigned short* srow = s_sum + k * nbp;
                unsigned key = 0;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int b = q * 64 + lane;
                    if (q == 0 || nblk > 64) {
                        const bool in = b < nblk;
                        const unsigned m16 = srow[in ? b : 0];
                        const unsigned M = m16 >> 4;                   // 0: no feasible node in the block; else 1 + LA + BA
                        const int p = b * 16 + 15 - (int)(m16 & 15u);
                        const unsigned canon = s_canon[in ? p : 0];
                        const int sn2 = *(const int*)((const char*)snrow + ncl4[q]);
                        // total = BA + LA + Simon + GpuShare (weights: registry.go:118-131, utils.go:321-333)
                        const unsigned kq = ((M - 1u + (unsigned)sn2) << 22) | ((2047u - canon) << 11) | (unsigned)p;
                        key = max(key, (M != 0u && in) ? kq : 0u);
                    }
                }
                key = wave_max_u32(key);
                pstar = (int)(key & 2047u);
                res = (int)(2047u - ((key >> 11) & 2047u));           // first maximum in canonical order
            }
        }
        // -------- assume: NodeInfo.AddPod (V/framework/types.go:482-508) + tile row + summary ----
        if (pstar >= 0) {
            const int blk = pstar >> 4, pos = pstar & 15;
            const int dwi = pos >> 2, sh8 = (pos & 3) * 8;
            uint4* my_row = (uint4*)(g_tile + ((size_t)blk * Kp + kk) * 16);
            uint4 T = (ablate & 1) ? make_uint4(0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u) : *my_row;
            uint4 st = (ablate & 4) ? make_uint4(1, 1, 50, 0) : g_state[pstar];
            st.x += (unsigned)__builtin_amdgcn_readlane((int)my_add_c, r_sig);
            st.y += (unsigned)__builtin_amdgcn_readlane((int)my_add_m, r_sig);
            st.z -= 1u;
            double nzc = 0.0, nzm = 0.0;
            if (!NZEQ) {
                uint2 z = g_nz[pstar];
                z.x += (unsigned)__builtin_amdgcn_readlane((int)my_addz_c, r_sig);
                z.y += (unsigned)__builtin_amdgcn_readlane((int)my_addz_m, r_sig);
                if (lane == 0) g_nz[pstar] = z;
                nzc = (double)z.x; nzm = (double)z.y;
            }
            if (lane == 0 && !(ablate & 4)) g_state[pstar] = st;
            const ShapeRow sh = s_shape[st.w & 0xFFFFu];
            const unsigned nb_raw = (ablate & 16) ? 100u : eval_node(my_req_c, my_req_m, my_nz_c, my_nz_m, my_zero, (double)st.x,
                                                                        (double)st.y, nzc, nzm, (int)st.z, sh);
            const unsigned wsel = dwi == 0 ? T.x : dwi == 1 ? T.y : dwi == 2 ? T.z : T.w;
            const unsigned old = (wsel >> sh8) & 0xFFu;                // this signature's byte before the cycle
            const unsigned nb = old ? nb_raw : 0u;                     // static mask / monotone infeasibility
            const unsigned wnew = (wsel & ~(0xFFu << sh8)) | (nb << sh8);
            T.x = dwi == 0 ? wnew : T.x;
            T.y = dwi == 1 ? wnew : T.y;
            T.z = dwi == 2 ? wnew : T.z;
            T.w = dwi == 3 ? wnew : T.w;
            // One wave: its vector memory accesses are served in order, so the next cycle's loads of this
            // tile row / state row observe these stores (same guarantee the LLVM memory model gives
            // wavefront-scope ordering); no cache maintenance, no wait.
            if (lane < K && nb != old) {
                if (!(ablate & 2)) ((unsigned char*)my_row)[pos] = (unsigned char)nb;
                my_sum[blk] = (unsigned short)block_key16(T);
                if (!nb) s_cnt[(st.w >> 16) * 64 + lane] -= 1;
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
    for (int p = lane; p < ni; p += 64) { const uint4 st = g_state[p]; uc += st.x; um += st.y; }
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

template <bool M, bool Z, bool PIN>
static hipError_t launch_mzp(const CacheLaunch& a, int n_blocks, size_t lds, hipStream_t st) {
    auto kern = cache_kernel<M, Z, PIN>;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(n_blocks), dim3(64), lds, st, a.ncls, a.rank, a.shape_of, a.a_pods, a.i_rq_cpu,
                       a.i_rq_mem, a.i_nz_cpu, a.i_nz_mem, a.i_npods, a.clsprefix, a.sigs, a.shapes, a.pods, a.orders,
                       a.scen, a.perm, a.static_mask, a.simon_raw, a.unscheduled, a.used_cpu, a.used_mem, a.place_step,
                       a.ws, a.sc);
    return hipGetLastError();
}

size_t cache_lds_bytes(int K, int ni_max, int Cn, int Cp, int n_shapes, bool nzeq) {
    return (size_t)carve(K, ni_max, Cn, Cp, n_shapes, nzeq).total;
}
size_t cache_ws_bytes(int K, int ni_max, int Cn, int Cp, int n_shapes, bool nzeq) {
    return (size_t)carve(K, ni_max, Cn, Cp, n_shapes, nzeq).ws_total;
}

// n_blocks scenarios; scenario of block b = a.perm[b]; a.ws = [n_blocks][cache_ws_bytes]
template <bool M, bool Z>
static hipError_t launch_mz(const CacheLaunch& a, int n_blocks, bool has_pin, size_t lds, hipStream_t st) {
    return has_pin ? launch_mzp<M, Z, true>(a, n_blocks, lds, st) : launch_mzp<M, Z, false>(a, n_blocks, lds, st);
}

hipError_t launch_cache(const CacheLaunch& a, int n_blocks, bool has_mask, bool nzeq, bool has_pin, size_t lds_bytes, hipStream_t st) {
    if (has_mask) return nzeq ? launch_mz<true, true>(a, n_blocks, has_pin, lds_bytes, st) : launch_mz<true, false>(a, n_blocks, has_pin, lds_bytes, st);
    return nzeq ? launch_mz<false, true>(a, n_blocks, has_pin, lds_bytes, st) : launch_mz<false, false>(a, n_blocks, has_pin, lds_bytes, st);
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
