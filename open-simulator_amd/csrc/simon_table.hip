// simon_table.hip -- cpu+memory scenario kernel, generations 4 to 6: one WAVE = one capacity-planning scenario over a
// (signature, node) score table, with the class term folded into the block summaries.
//
// What generation 3 (round 1's simon_cache.hip, since replaced by this file) established: a scheduling cycle changes ONE node, pods come from K distinct
// request signatures, so (feasible, LeastAllocated + BalancedAllocation) of every (signature, node) is a table of
// bytes of which one column changes per cycle; nodes are laid out class-major (stable in canonical order, every node
// class padded to 16, so a block of 16 nodes has ONE Simon node class) and a per-(signature, block) summary in LDS
// answers findNodesThatFitPod + prioritizeNodes + selectHost (V/core/generic_scheduler.go:131-209) with one LDS read
// per 16 nodes.  Its cycle, measured (profiles/README.md): ~250 dependent instructions and five waits (feasible-class
// counters, summary row, canonical index of the block's best node, tile row + node state from L2, node shape).
//
// This generation keeps the byte table (its footprint decides whether 4 096 scenarios stay in the Infinity Cache: a
// 16-bit table was measured at 39 GB of HBM traffic per step against 6.5 GB) and changes what the SUMMARY holds:
//   * sum[k][b] = (best byte of the block + class term of (signature k, class of block b)) << 4 | 15 - position.
//     The Simon / Open-Gpu-Share term (pkg/simulator/plugin/simon.go:76-101: min-max normalised over the FEASIBLE
//     nodes) depends on (signature, node class) and on which node classes still hold a feasible node for the
//     signature.  That set only ever shrinks (Requested grows, pod slots shrink), so the term is folded in and a
//     signature's summary row is re-based -- 2 LDS entries per lane -- on the rare cycle after a class lost its last
//     feasible node (at most Cn - 1 times per signature and scenario).  The scan needs no class term, no feasible-class
//     ballot, no counters;
//   * the arg-max key is total << 12 | 4095 - POSITION.  Inside a class, position order is canonical order; across
//     classes it is not, so after the reduction the wave checks whether blocks of ANOTHER class tie with the winner
//     (one compare + ballot) and only then resolves the tie through the canonical indices (static per-class node lists,
//     L2-hot) -- the canonical index leaves the critical path;
//   * shape ids live in LDS (one byte per position), so the shape row is fetched while the table row and the node state
//     are in flight -- one memory round trip per cycle, nothing dependent behind it.
// Per pod: [dirty signature? re-base its summary row] -> one u16 LDS read per 16 nodes -> 4 VALU -> one DPP wave max
// -> tie check -> node.  assume (V/scheduler.go:371 -> NodeInfo.AddPod, V/framework/types.go:482-508): the wave loads
// the node's 12-byte state and lane k the 16-byte table row (signature k, touched block); lane k re-evaluates signature
// k with exactly the fp64 sequences of simon_fast.hip, patches its byte, re-reduces its row and
// stores the summary entry.
//
// Generation 5 (template COARSE, see tcarve): two-level summary -- 64-position entries in LDS, per-16 entries and the
// feasible-node counters in HBM -- for batches whose one-level summary would starve the CU of waves (many signatures).
// Generation 6 (template REST, see table_kernel): pods with per-node filters the table cannot hold (Open-Gpu-Share devices,
// required anti-affinity on node-level topology keys) scan their signature's rows under per-block position masks.
//
// Limits: K <= 384 signatures (two per lane in registers, further groups of 128 from memory), padded scenario size <= 4096 positions (<= 4095 nodes; two-level: 8192 /
// 8191), <= 128 node classes (two per lane where a select / walk is lane-shaped: CN2; 129 .. 256 without rows and walks on the one-level layout: simon_table_cls4.hip), NARROW preconditions (simon_hip.hip::choose_variant), no zero-capacity node, orders are
// permutations; REST: <= 32 GPU requests, <= 120 terms, <= 63 mask rows per pod.
#include "simon_table.h"

#include <algorithm>
#include <type_traits>

// Phase profile of the scheduling cycle (profiles/build_variant.sh ... -DSIMON_TABLE_PROFILE): s_memtime stamps at the phase
// boundaries, accumulated per wave, written to TableCold::prof ([workgroup][12] ticks).  Not compiled into the product build.
#ifdef SIMON_TABLE_PROFILE
#define TPROF_DECL unsigned long long tp_prev = __builtin_readcyclecounter(), tp_acc[24] = {0};
#define TPROF(i) do { const unsigned long long t_ = __builtin_readcyclecounter(); tp_acc[i] += t_ - tp_prev; tp_prev = t_; } while (0)
#define TPROF_WAIT_LDS asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define TPROF_WAIT_MEM asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define TPROF_DECL
#define TPROF(i) do { } while (0)
#define TPROF_WAIT_LDS do { } while (0)
#define TPROF_WAIT_MEM do { } while (0)
#endif

namespace simon {

// Pointers that arrive through TableCold -- a struct in device memory -- are generic to the compiler: every access through them was a
// FLAT instruction (round 6: 40 .. 90 per kernel, among them the REST select's Simon row and the pod's filter entries).  A flat load
// counts on vmcnt AND lgkmcnt, so the next wait for an LDS read also waited for the memory round trip.  gp() names the address
// space: the tables behind TableCold are global memory, always.
#ifdef SIMON_TABLE_LDS_TU
#define SIMON_AS1   // (generation 4 with the workspace in LDS: its cycle touches no global memory, and the unit keeps round 5's code layout -- a single wave per CU is sensitive to it: same-box A/B 5.54 -> 5.80 ms on config 3 x 64 with identical loop bodies)
#else
#define SIMON_AS1 __attribute__((address_space(1)))
#endif
template <class T> using GPtr = T SIMON_AS1*;
template <class T> __device__ __forceinline__ GPtr<T> gp(T* p) { return (GPtr<T>)p; }
// (HIP's uint2 / uint4 / int2 are classes: their copy constructors take generic references, so vectors travel as native ones)
typedef unsigned u32x4n __attribute__((ext_vector_type(4)));
typedef unsigned u32x2n __attribute__((ext_vector_type(2)));
template <class T> __device__ __forceinline__ uint4 ldg_u4(GPtr<T> p) { const u32x4n v = *(GPtr<const u32x4n>)p; return make_uint4(v.x, v.y, v.z, v.w); }
template <class T> __device__ __forceinline__ uint2 ldg_u2(GPtr<T> p) { const u32x2n v = *(GPtr<const u32x2n>)p; return make_uint2(v.x, v.y); }
template <class T> __device__ __forceinline__ int2 ldg_i2(GPtr<T> p) { const u32x2n v = *(GPtr<const u32x2n>)p; return make_int2((int)v.x, (int)v.y); }

typedef unsigned short u16x2t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pkmax_t(unsigned a, unsigned b) {
    u16x2t x = __builtin_bit_cast(u16x2t, a), y = __builtin_bit_cast(u16x2t, b);
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(x, y));
}

__device__ __forceinline__ int la_term_t(double r, double rc100) {   // == la_term_f of simon_fast.hip (exactness: DESIGN.md 5.2)
    const double C = 100.0 + 0x1.0p-33;
    return (int)__builtin_fma(-r, rc100, C);
}

// max over the 16 bytes of one table row of (byte << 4 | 15 - position)
__device__ __forceinline__ unsigned block_key16_t(const uint4 R) {
    const unsigned M8 = 0x00FF00FFu;
#define SIMON_LO(w, q4) ((((w) & M8) << 4) | (unsigned)((15 - (q4)) | ((15 - ((q4) + 2)) << 16)))
#define SIMON_HI(w, q4) (((((w) >> 8) & M8) << 4) | (unsigned)((15 - ((q4) + 1)) | ((15 - ((q4) + 3)) << 16)))
    unsigned m0 = pkmax_t(SIMON_LO(R.x, 0), SIMON_HI(R.x, 0));
    unsigned m1 = pkmax_t(SIMON_LO(R.y, 4), SIMON_HI(R.y, 4));
    unsigned m2 = pkmax_t(SIMON_LO(R.z, 8), SIMON_HI(R.z, 8));
    unsigned m3 = pkmax_t(SIMON_LO(R.w, 12), SIMON_HI(R.w, 12));
#undef SIMON_LO
#undef SIMON_HI
    m0 = pkmax_t(pkmax_t(m0, m1), pkmax_t(m2, m3));
    return max(m0 & 0xFFFFu, m0 >> 16);
}

// Block key of a table row whose byte at position `pos` is being replaced: the v_perm selectors that expand the row's bytes to
// u16 pairs take the touched byte from `nb` (selector 4 = byte 0 of the first operand) instead of the row -- the patch costs no
// instruction.  kSel[pos] = 8 selectors: dword i pair lo (bytes 0, 1), pair hi (bytes 2, 3); 0x0c = constant 0.
struct SelTab { unsigned v[16][8]; };
constexpr SelTab make_sel_tab() {
    SelTab t{};
    for (int pos = 0; pos < 16; ++pos)
        for (int i = 0; i < 4; ++i)
            for (int h = 0; h < 2; ++h) {
                unsigned b0 = (unsigned)(2 * h), b1 = (unsigned)(2 * h + 1);            // byte indices inside dword i
                if (pos == 4 * i + 2 * h) b0 = 4;
                if (pos == 4 * i + 2 * h + 1) b1 = 4;
                t.v[pos][2 * i + h] = 0x0c000c00u | b0 | (b1 << 16);
            }
    return t;
}
static __device__ __constant__ SelTab kSelTab = make_sel_tab();   // (static: simon_table_team.hip compiles this file a second time)
#define kSel (kSelTab.v)

__device__ __forceinline__ unsigned pk_key(unsigned pair, unsigned c) {   // (x << 4) + c on both halves in ONE instruction
    unsigned d;
    asm("v_pk_mad_u16 %0, %1, 16, %2 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(pair), "v"(c));
    return d;
}
__device__ __forceinline__ unsigned block_key16_patched(const uint4 R, unsigned nb, const uint4 sa, const uint4 sb) {
#define SIMON_C(q4) ((unsigned)((15 - (q4)) | ((15 - ((q4) + 1)) << 16)))
    const unsigned m0 = pkmax_t(pk_key(__builtin_amdgcn_perm(nb, R.x, sa.x), SIMON_C(0)), pk_key(__builtin_amdgcn_perm(nb, R.x, sa.y), SIMON_C(2)));
    const unsigned m1 = pkmax_t(pk_key(__builtin_amdgcn_perm(nb, R.y, sa.z), SIMON_C(4)), pk_key(__builtin_amdgcn_perm(nb, R.y, sa.w), SIMON_C(6)));
    const unsigned m2 = pkmax_t(pk_key(__builtin_amdgcn_perm(nb, R.z, sb.x), SIMON_C(8)), pk_key(__builtin_amdgcn_perm(nb, R.z, sb.y), SIMON_C(10)));
    const unsigned m3 = pkmax_t(pk_key(__builtin_amdgcn_perm(nb, R.w, sb.z), SIMON_C(12)), pk_key(__builtin_amdgcn_perm(nb, R.w, sb.w), SIMON_C(14)));
#undef SIMON_C
    const unsigned m = pkmax_t(pkmax_t(m0, m1), pkmax_t(m2, m3));
    return max(m & 0xFFFFu, m >> 16);
}

// max over each 16-lane row (result in every lane of the row)
__device__ __forceinline__ unsigned row16_max_t(unsigned v) {
    v = max(v, (unsigned)SIMON_DPP(0, (int)v, 0xB1, 0xF));
    v = max(v, (unsigned)SIMON_DPP(0, (int)v, 0x4E, 0xF));
    v = max(v, (unsigned)SIMON_DPP(0, (int)v, 0x141, 0xF));
    v = max(v, (unsigned)SIMON_DPP(0, (int)v, 0x140, 0xF));
    return v;
}

// wave64 sum / or on the DPP pattern of simon_device.h's wave_max (identity 0 in the lanes a row broadcast does not reach)
__device__ __forceinline__ int wave_sum_i32_t(int v) {
    v += SIMON_DPP(0, v, 0xB1, 0xF);
    v += SIMON_DPP(0, v, 0x4E, 0xF);
    v += SIMON_DPP(0, v, 0x141, 0xF);
    v += SIMON_DPP(0, v, 0x140, 0xF);
    v += SIMON_DPP(0, v, 0x142, 0xA);
    v += SIMON_DPP(0, v, 0x143, 0xC);
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ unsigned wave_or_u32_t(unsigned v) {
    v |= (unsigned)SIMON_DPP(0, (int)v, 0xB1, 0xF);
    v |= (unsigned)SIMON_DPP(0, (int)v, 0x4E, 0xF);
    v |= (unsigned)SIMON_DPP(0, (int)v, 0x141, 0xF);
    v |= (unsigned)SIMON_DPP(0, (int)v, 0x140, 0xF);
    v |= (unsigned)SIMON_DPP(0, (int)v, 0x142, 0xA);
    v |= (unsigned)SIMON_DPP(0, (int)v, 0x143, 0xC);
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

struct NodeState { unsigned rq_c, rq_m, freep; };   // Requested cpu / mem (gcd units), free pod slots: 12 B per position

struct TCarve {
    int sum, sn, cnt, shape, seg, tmp, ext, zdom, stash, tab, tabi, tab2, xch, total;   // LDS offsets (multiples of 16)
    int nbp;
};
// summary row pitch in u16 entries: >= nblk with an odd pitch in dwords or an odd pitch in entries, so that the K column
// entries of one block (pitch * 2 bytes apart) spread over the LDS banks (a 2-way conflict on a 16-bit store costs nothing)
__host__ __device__ inline int table_nbp(int nblk) { return (nblk & 1) || (nblk & 3) == 2 ? nblk : nblk + 1; }
// LDS per workgroup decides how many scenario waves a CU holds: 160 KB / 16 = 10 KB (measured, profiles/micro/occupancy_probe.hip:
// 10 240 B -> 16 workgroups per CU, 12 288 B -> 12), so everything but the summary is kept tiny: no per-position arrays (the shape
// of a node follows from its class), Simon raw scores stay in global memory (read on the rare re-base only).
//
// COARSE (two-level summary): the LDS entry covers 64 positions instead of 16 -- a quarter of the LDS, so that batches with many
// signatures keep 16+ scenario waves per CU -- and the per-16 entries move to the scenario's HBM workspace, where only the assume
// reads them (4 entries = 8 bytes per signature, fetched with the table row, off the dependent chain); the feasible-node counters
// move there too (touched on the rare cycle a node becomes infeasible for a signature).  Classes are then padded to 64 positions.
__host__ __device__ inline TCarve tcarve(int K, int ni_max, int Cn, bool coarse, bool rest, int nzk = -1, bool cls4 = false) {   // (cls4: 129 .. 256 classes, simon_table_cls4.hip -- a constant to every kernel)
    auto al = [](int x) { return (x + 15) & ~15; };
    TCarve c;
    c.nbp = table_nbp(ni_max / (coarse ? 64 : 16));
    int o = 0;
    c.sum = o; o += al(K * c.nbp * 2);
    c.sn = o; o += al(K * Cn * 2);
    // (REST, round 6: the feasible-node counters stay in LDS on the two-level layout as well.  On a gpushare cluster that fills up -- config
    // 5 -- some (signature, node) byte drops to 0 on almost every cycle, and the counter's read-modify-write in the HBM workspace was a
    // dependent memory round trip inside the refresh; config 5: 84 signatures x 8 classes = 2.7 KB; with hundreds of signatures the host's LDS bound decides.)
#ifdef SIMON_REST_CNT_HBM
    (void)rest;
    c.cnt = o; o += coarse ? 0 : al(K * Cn * 4);
#else
    c.cnt = o; o += ((coarse && !rest) || cls4) ? 0 : al(K * Cn * 4);   // (129 .. 256 classes, simon_table_cls4.hip: counters in the workspace, 32-byte shapes)
#endif
    c.shape = o; o += Cn * (cls4 ? 32 : 48);
    c.seg = o; o += al((Cn + 1) * 4);
    c.tmp = o; o += al(Cn * 4);
    c.ext = o; o += cls4 ? al(K * 8) : 0;                              // cls4: [K] extremes of the Simon raw scores at the last re-base (table_kernel: the interior-class test)
    // SPREAD (nzk >= 0): zone domain of a class per zone-like key; for spread_select two bytes per position and the score table of
    // the pod being placed
    const bool ipa = nzk >= 0 && (nzk & 0x100);                       // the problem has preferred pod (anti-)affinity terms: a second table
    const bool team = nzk >= 0 && (nzk & 0x200);                      // several waves per scenario (table_kernel: NW > 1): totals table of its own + exchange slots
    const bool cn2 = nzk >= 0 && (nzk & 0x400);                       // 65 .. 128 node classes under the walks (table_kernel: CN2): a score table of kSpreadTabMax2 entries
    if (nzk >= 0) nzk &= 0xFF;
    c.zdom = o; o += nzk >= 0 ? al((nzk > 0 ? nzk : 1) * Cn) : 0;
    c.stash = o; o += nzk >= 0 ? al(ni_max * 2) : 0;
    c.tab = o; o += nzk >= 0 ? (cn2 ? kSpreadTabMax2 : kSpreadTabMax) * 4 : 0;
    c.tabi = o; o += ipa ? (cn2 ? kSpreadTabMax2 : kSpreadTabMax) * 4 : 0;
    c.tab2 = o; o += team ? (cn2 ? kSpreadTabMax2 : kSpreadTabMax) * 4 : 0;
    c.xch = o; o += team ? kTeamWavesMax * 6 * 4 : 0;
    c.total = o;
    return c;
}
// HBM workspace of ONE scenario with `ni` padded positions: byte table [ni / 16][K][16], node state [ni] x 12 B, (when
// NonZeroRequested differs from Requested) [ni] x 8 B, and (COARSE) the per-16 summary entries [ni / 64][K][4] u16 and the
// feasible-node counters [K][Cn] i32
// ... and (REST, M mask rows) the position masks [ni / 16][M] u16 and the GPU devices of every position: used [ni][8], per-device
// total [ni], device count [ni] (u32 each), extra-resource Requested [ni][8] and allocatable [ni][8], and the topology domain of every
// position under the NZ keys that are not node-level [NZ][ni] u16
__host__ __device__ inline size_t table_ws_of(int K, int ni, bool nzeq, bool coarse, int Cn, int M, int NZ, int TH = 0, int TZ = 0, bool cls4 = false) {
    size_t w = ((size_t)(ni / 16) * K * 16 + 127) & ~(size_t)127;
    w += ((size_t)ni * 12 + 127) & ~(size_t)127;
    if (!nzeq) w += ((size_t)ni * 8 + 127) & ~(size_t)127;
    if (coarse) w += (((size_t)(ni / 64) * K * 8 + 127) & ~(size_t)127) + (((size_t)K * Cn * 4 + 127) & ~(size_t)127);
    else if (cls4) w += ((size_t)K * Cn * 4 + 127) & ~(size_t)127;   // (simon_table_cls4.hip: the feasible-node counters)
    if (M > 0) w += (((size_t)(ni / 16) * M * 2 + 127) & ~(size_t)127) + (size_t)ni * (40 + 64) + (size_t)NZ * ni * 2 + (((size_t)M * 4 + 127) & ~(size_t)127);
    // SPREAD: matching pods per (hostname-key term, position) u8 [TH][ni], per (zone-key term, domain) u32 [TZ][16]
    if (TH > 0 || TZ > 0) w += (((size_t)TH * ni + 127) & ~(size_t)127) + (((size_t)TZ * 16 * 4 + 127) & ~(size_t)127) + (((size_t)TH + 127) & ~(size_t)127)
                              + (((size_t)ni * 2 + 127) & ~(size_t)127);   // + the rank of every position (per-scenario node order)
    return w;
}

// Open-Gpu-Share on gcd-normalised 32-bit quantities (all < 2^31): GpuNodeInfo.AllocateGpuId, pkg/type/open-gpu-share/cache/
// gpunodeinfo.go:232-290 -- the feasibility question Filter asks (open-gpu-share.go:74-78) and the commit Reserve performs
// (:147-188), the same walk as simon_wide.hip::gpu_feasible / gpu_commit_regs.
__device__ __forceinline__ bool gpu_fits_t(const unsigned (&u)[8], int cnt, unsigned tot, unsigned req, int num) {
    if (req == 0u || num <= 0 || cnt <= 0) return false;
    if (num == 1) {                                       // tightest fit exists iff some device has room (:255-267)
        bool ok = false;
#pragma unroll
        for (int d = 0; d < 8; ++d) ok = ok || (d < cnt && (int)tot - (int)u[d] >= (int)req);
        return ok;
    }
    int got = 0;                                          // two-pointer greedy (:268-287): device d takes floor(idle / req) slices
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        if (d >= cnt) continue;
        int idle = (int)tot - (int)u[d];
        while (idle >= (int)req && got < num) { ++got; idle -= (int)req; }
    }
    return got == num;
}
// Returns the booking as simon_batch_out.gpu_slices wants it: byte d = slices on device d (the gpu-index annotation the plugin writes,
// pkg/type/open-gpu-share/utils/pod.go:117-127: ids ascending, one per slice).
__device__ __forceinline__ unsigned long long gpu_commit_t(unsigned (&u)[8], int cnt, unsigned tot, unsigned req, int num) {
    if (req == 0u || num <= 0 || cnt <= 0) return 0ull;
    if (num == 1) {                                       // tightest fit, lowest id on ties (:255-267)
        int cand = -1, cand_idle = 0;
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            const int idle = (int)tot - (int)u[d];
            if (d < cnt && idle >= (int)req && (cand < 0 || idle < cand_idle)) { cand = d; cand_idle = idle; }
        }
#pragma unroll
        for (int d = 0; d < 8; ++d) u[d] += (d == cand) ? req : 0u;
        return cand >= 0 ? 1ull << (8 * cand) : 0ull;
    }
    unsigned w[8];
    int got = 0;
    unsigned long long slices = 0;
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        w[d] = u[d];
        if (d >= cnt) continue;
        int idle = (int)tot - (int)w[d];
        while (idle >= (int)req && got < num) { ++got; idle -= (int)req; w[d] += req; slices += 1ull << (8 * d); }
    }
    if (got == num) {
#pragma unroll
        for (int d = 0; d < 8; ++d) u[d] = w[d];
        return slices;
    }
    return 0ull;
}

// KQ: signatures per lane (1: K <= 64, 2: K <= 128).  HAS_PIN: the stream holds pinned pods (own instantiation: the extra
// branch costs the common kernel time).
// NBQ: blocks per lane (1, 2 or 4: padded scenario sizes up to 1024 / 2048 / 4096 positions) -- a template parameter so that the
// scan is straight-line code (as run-time conditions the four reads became four dependent LDS round trips).
// fitsRequest's ephemeral-storage and extended-resource checks (fit.go:264-299) of one request on one node: component 0 is checked
// whatever the request, an extended resource only when the pod names it (non-zero)
__device__ __forceinline__ bool xres_fits_t(const unsigned (&req)[5], const unsigned (&used)[5], const unsigned (&alloc)[5]) {
    bool ok = !(alloc[0] < req[0] + used[0]);
#pragma unroll
    for (int r = 1; r < 5; ++r) ok = ok && !(req[r] != 0u && alloc[r] < req[r] + used[r]);
    return ok;
}

#ifndef SIMON_SPREAD_BATCH1
#define SIMON_SPREAD_BATCH1 6
#endif
#ifndef SIMON_SPREAD_BATCH2
#define SIMON_SPREAD_BATCH2 6
#endif
#ifndef SIMON_SPREAD_BATCH4
#define SIMON_SPREAD_BATCH4 2
#endif
// units of 64 positions per batch of loads in spread_select: pass 1 and pass 2 of the common shape, both passes of the general one
constexpr int kSpreadBatch1 = SIMON_SPREAD_BATCH1, kSpreadBatch2 = SIMON_SPREAD_BATCH2, kSpreadBatch4 = SIMON_SPREAD_BATCH4;
// NBQ counts summary ENTRIES per lane: 16 positions each, or 64 with COARSE (tcarve, above).
// REST: some pods carry filters the (signature, node) table cannot hold -- Open-Gpu-Share device memory, required anti-affinity on
// a node-level topology key (both directions).  Those filters live as per-block POSITION MASKS xm[block][row] (u16, bit = node
// excluded): one row per GPU signature, two per term (a pod matching the term sits there / a pod requiring it sits there).  A
// table-only pod takes the summary path unchanged.  A REST pod scans its signature's table rows, 16 positions per lane and step,
// with the bytes of excluded positions cleared, keeps the best node per class (LDS max), normalises the Simon term over the
// classes that kept a node (simon.go:76-101 runs on the nodes feasible under ALL filters) and picks the first maximum in
// canonical order.  assume additionally sets the pod's term rows and, for a GPU pod, commits the devices and refreshes the bit
// of every GPU signature (lane g = signature g) on that node.
// RANKED: per-scenario node order (simon_set_node_ranks).  A template parameter although it only touches rare paths (tie-breaks,
// preset pods, the 64-step placement flush): as a run-time flag it cost config 5 4 % (same-box A/B, profiles/README.md).
// AFF: some pod class carries required-affinity entries (REST only; own instantiation for the same reason: +5.7 % on config 5 as
// run-time tests of the entries' bit 31).
// MANY: more than 128 signatures (KQ = 2, two-level summary; since round 6 also with the REST rows): signatures 0 .. 127 live in lane registers as usual, the rest is
// refreshed from TableCold::sigs in up to two further groups of 128 whose table rows are fetched WITH group 0's (one memory round
// trip per cycle; own instantiation: the extra rows cost ~40 VGPRs the K <= 128 kernels must not pay); groups beyond those
// (385 .. 1 023 signatures, round 4) follow one round trip each.
// SPREAD (generation 7): pod classes with soft PodTopologySpread constraints (ScheduleAnyway: the system defaults every pod a Service /
// ReplicaSet / StatefulSet selects gets, podtopologyspread/plugin.go:39-50) -- see spread_select.
// NW (team mode, SPREAD only): waves per scenario.  A batch with fewer scenarios than the chip has wave slots -- what a real
// Applier.Run offers: a handful of candidate sizes (pkg/apply/apply.go:203-259) -- leaves most SIMDs idle when one wave owns a
// scenario, and a pod with soft spread constraints walks EVERY position twice (spread_select).  With NW > 1 the workgroup is NW waves:
// wave 0 (the leader) runs the scheduling cycle exactly as the single-wave kernel does; for a spread pod all NW waves walk a
// contiguous share of the units each, and the extremes of pass 1 and the best key of pass 2 are combined through LDS (three
// s_barriers per spread pod).  The waves of one workgroup share the CU's vector L1 (no tgsplit), so the leader's stores are
// visible to the helpers after the barrier's s_waitcnt without cache maintenance.  The prologue (K x n table) is split the same way.
// LDSWS (round 5): the scenario's WORKSPACE -- byte table, node state -- lives in LDS behind the summaries instead of in HBM.  For the
// batches a real Applier.Run offers (a handful of sizes: at most one scenario per CU, so the CU's 160 KB are this scenario's) on problems
// whose table fits (config 3's pool: 1 600 positions x 42 signatures = 86 KB; config 2: 12 KB) the one memory round trip of a scheduling
// cycle -- the winner's table rows and state from L2 -- becomes an LDS access.  Same code: the pointers are derived from the LDS base in
// this instantiation, the compiler addresses them as LDS.  Generation 4 only (one-level summary, no REST / SPREAD / folds).
// LDSX (round 5): generation 6's position-mask rows (g_xm), their totals and the canonical index of every position live in LDS -- the
// same idea for the batches of a gpushare sweep (config 5 x 256: one scenario per CU, 71 KB of rows): the REST select's filter words and
// the canonical tie-break of the device / device-less twin classes (every other cycle of config 5) stop being memory round trips.
// CN2 (round 6): 65 .. 128 internal node classes under generation 7's walks -- the zone split multiplies the node shapes by the zones, so 25
// shapes in 3 zones used to leave the table.  spread_select keeps its per-class values one per lane; here lane l also holds class 64 + l
// (as the prologue and the re-base do for the instantiations without walks since round 4).  Own instantiations (the second set of
// per-class registers would spill in the 128-VGPR kernels that serve <= 64 classes); single wave, no preferred / hard terms in the walk.
template <bool HAS_MASK, bool NZEQ, bool HAS_PIN, int KQ, int NBQ, bool COARSE, bool REST, bool RANKED, bool AFF, bool MANY, bool SPREAD, int NW = 1, bool LDSWS = false, bool LDSX = false, bool CN2 = false>
#ifndef SIMON_SPREAD_WAVES
#define SIMON_SPREAD_WAVES 4
#endif
#ifndef SIMON_SPREAD_IPA_WAVES
#define SIMON_SPREAD_IPA_WAVES 3
#endif
// (the SPREAD instantiations hold a batch of loads in registers: kept to 128 VGPRs = four scenario waves per SIMD, the same as the others)
// (team mode runs at one or two workgroups per CU = one or two waves per SIMD: no register budget to keep.  Tried on top and dropped,
// same-box A/B profiles/r04/r04e_*, r04f_*: the canonical indices fetched with pass 1's loads and stashed in LDS for pass 2 (+4 %),
// the spread entries fetched one pod ahead (+-1 %), 1 / 3 / 4 waves per SIMD as the register budget (within 2 %), 8 and 16 waves
// per scenario (slower on every batch size, r04b_*))
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(SPREAD ? (NW > 1 ? 2 : AFF ? SIMON_SPREAD_IPA_WAVES : SIMON_SPREAD_WAVES) : 1))) void table_kernel(
    const TableCold* __restrict__ cold_, const int32_t* __restrict__ cls_list_pool, const PodRowC* __restrict__ pods,
    const int32_t* __restrict__ orders, const int32_t* __restrict__ perm, const unsigned long long* __restrict__ ws_off,
    int32_t* __restrict__ place_step, unsigned char* ws, const TableScalars sc) {
    // Pointers the hot loop never touches live in a device-resident struct: as kernel arguments (24 pointers) they kept the
    // loop at the SGPR limit and spilled into VGPR lanes.
    GPtr<const TableCold> const cold = gp(cold_);
    GPtr<const int32_t> __restrict__ const ncls = gp(cold->ncls);
    GPtr<const int32_t> __restrict__ const cls_off = gp(cold->cls_off); GPtr<const int32_t> __restrict__ const clsprefix = gp(cold->clsprefix);
    GPtr<const int32_t> __restrict__ const a_pods = gp(cold->a_pods); GPtr<const uint32_t> __restrict__ const i_rq_cpu = gp(cold->i_rq_cpu);
    GPtr<const uint32_t> __restrict__ const i_rq_mem = gp(cold->i_rq_mem); GPtr<const uint32_t> __restrict__ const i_nz_cpu = gp(cold->i_nz_cpu);
    GPtr<const uint32_t> __restrict__ const i_nz_mem = gp(cold->i_nz_mem); GPtr<const int32_t> __restrict__ const i_npods = gp(cold->i_npods);
    const SigRow* __restrict__ const sigs = cold->sigs; const ShapeRow* __restrict__ const shapes = cold->shapes;   // (structs: copied through generic references)
    const ScenarioDesc* __restrict__ const scen = cold->scen; GPtr<const uint64_t> __restrict__ const static_mask = gp(cold->static_mask);
    GPtr<const int32_t> __restrict__ const simon_raw = gp(cold->simon_raw);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int Cn = sc.Cn, Cp = sc.Cp, P = sc.P, K = sc.K;
    constexpr int UB = COARSE ? 6 : 4;                                // log2(positions per summary entry)
    constexpr int UNIT = 1 << UB;
    constexpr unsigned UMASK = UNIT - 1;
    constexpr int KB = COARSE ? 13 : 12;                              // width of the position field of the arg-max key
    constexpr unsigned PMASK = (1u << KB) - 1u;
    static_assert(!REST || COARSE, "the REST path is built on the two-level layout");
    // 129 .. 256 internal node classes (end of round 6): where no select is lane-shaped -- no REST rows, no SPREAD walks -- the prologue's scan and the class terms'
    // re-base take further groups of 64 classes.  The re-base sits INSIDE the scheduling cycle's register budget (as a run-time branch it cost every kernel of the
    // family 2 .. 8 VGPRs, the 127-VGPR ones their fourth wave per SIMD), so it is compiled into eight kernels of their own: simon_table_cls4.hip.  One-level layout
    // only: a class segment is padded to a summary unit, and 129 units of 64 positions are more than the two-level table holds.  What grows with the class count is
    // kept out of LDS where the cycle does not read it (tcarve): the feasible-node counters live in the scenario's workspace (read-modify-written on the rare cycle
    // a byte drops to 0), a class shape is 32 B (the two 100/cap factors are one IEEE multiplication each, as the host computes them) -- BASELINE config 3's 42
    // signatures on 160 shapes ask for 38.6 KB instead of 68 KB, four scenarios per CU instead of two.
    constexpr bool kCls4 = CN2 && !REST && !SPREAD;
#ifdef SIMON_REST_CNT_HBM
    constexpr bool CNT_LDS = !COARSE;                                 // A/B builds: round 5's home of the two-level layout's counters
#else
    constexpr bool CNT_LDS = (!COARSE || REST) && !kCls4;             // feasible-node counters per (signature, class) in LDS (tcarve)
#endif
#ifdef SIMON_TIE_SPECULATE
    constexpr bool TIE_FIRST = false;                                 // A/B builds: every instantiation speculates
#else
    constexpr bool TIE_FIRST = REST || MANY;                          // MANY: a lost speculation would reload three groups of rows
#endif
    static_assert(!MANY || (KQ == 2 && COARSE && !CN2 && !LDSX), "MANY = groups of 128 signatures on the two-level layout (since round 6 also under the REST select: its rows and counters are indexed by signature already)");
    constexpr int NG = MANY ? 2 : 0;                                  // further signature groups (K <= 128 (1 + NG))
    static_assert(!SPREAD || COARSE, "SPREAD is built on the two-level layout");
    constexpr bool RS = REST && SPREAD;                               // generation 7's walks over generation 6's position-mask rows (round 6)
    // (AFF names two things: required-affinity entries to the REST half, preferred / hard terms in the walk to the SPREAD half.  REST && SPREAD
    // && AFF is launched for the latter -- spread_supported keeps required affinity out -- and carries the former's code unused.)
    static_assert(!RS || (!LDSX && !CN2), "REST && SPREAD: <= 64 classes");
    static_assert(!(SPREAD && MANY) || NBQ == 2, "SPREAD with more than 128 signatures: the two-blocks-per-lane instantiations only");
    static_assert(NW == 1 || SPREAD, "team mode exists for the SPREAD instantiations");
    static_assert(!CN2 || (SPREAD && !MANY && !REST) || (REST && !LDSX && !SPREAD) || (!REST && !SPREAD && !COARSE && !MANY && !LDSWS), "CN2 = two node classes per lane in spread_select (<= 128 signatures) or in rest_select (rows in HBM); without rows and walks: 129 .. 256 classes on the one-level layout (simon_table_cls4.hip)");
    constexpr int TABMAX = CN2 ? kSpreadTabMax2 : kSpreadTabMax;
    const TCarve cv = tcarve(K, sc.ni_max, Cn, COARSE, REST, SPREAD ? (sc.NZK | ((sc.static_tables & 64) ? 0x100 : 0) | (NW > 1 ? 0x200 : 0) | (CN2 ? 0x400 : 0)) : -1, kCls4);
    const int TH = SPREAD ? sc.TH : 0, TZ = SPREAD ? sc.TZ : 0, NZK = SPREAD ? sc.NZK : 0;
    signed char* s_zdom = (signed char*)(smem + cv.zdom);          // SPREAD: [NZK][Cn]
    unsigned short* s_stash = (unsigned short*)(smem + cv.stash);  // SPREAD: [positions] count | table byte << 8 of the pod being placed
    int* s_tabi = (int*)(smem + cv.tabi);                           // SPREAD (problems with preferred pod (anti-)affinity): [class << lg | count] InterPodAffinity raw score
    int* s_tab = (int*)(smem + cv.tab);                             // SPREAD: [class << lg | count] raw score, then class term + 2 x score
    int* s_tot = NW > 1 ? (int*)(smem + cv.tab2) : s_tab;          // ... team mode: the totals get a table of their own (every wave writes all of it)
    int* s_xch = (int*)(smem + cv.xch);                             // team mode: [NW][4] extremes of pass 1, then [NW][2] best key / position of pass 2
    const int M = REST ? sc.M : 0, G = REST ? sc.G : 0, X = REST ? sc.X : 0, NZ = REST ? sc.NZ : 0;
    const int nbp = cv.nbp;
    unsigned short* s_sn = (unsigned short*)(smem + cv.sn);         // [K][Cn]: the class term (<= 822) currently folded into row k
    int* s_cnt = (int*)(smem + cv.cnt);                             // [K][Cn]: feasible nodes of class d for signature k (!COARSE)
    const ShapeRow* s_shape = (const ShapeRow*)(smem + cv.shape);   // [Cn]: shape of a node class (a class shares its allocatable); kCls4: 32 B rows (capacity, reciprocal)
    auto shape_of = [&](int d) -> ShapeRow {
        if constexpr (kCls4) {
            const double* r = (const double*)(smem + cv.shape) + 4 * d;
            return ShapeRow{r[0], r[1], r[2], r[3], 100.0 * r[2], 100.0 * r[3]};      // (simon_hip.hip: rc100 = 100.0 * rc, IEEE; the library is built without contraction)
        } else return s_shape[d];
    };
    int* s_seg = (int*)(smem + cv.seg);                             // [Cn + 1]: first position of a class segment
    int* s_tmp = (int*)(smem + cv.tmp);
    int2* s_ext = (int2*)(smem + cv.ext);                           // kCls4: [K] (lo, hi) of the Simon raw scores over the classes that held a feasible node at row k's last re-base

    const unsigned Krow = (unsigned)K * 16u;                          // bytes of one block's rows
    // Byte offset of (block of 16 positions, signature k) = tile_blk(block) + k * KS.  The table is [block][K][16]: a placement
    // refreshes one block's K rows, contiguous.  SPREAD keeps [unit of 64][K][64] instead: its select walks ONE signature's bytes of
    // every position for each pod, and 64 contiguous bytes per unit waste a quarter of the cache lines that 4 x 16 do.
    const unsigned KS = SPREAD ? 64u : 16u;
    auto tile_blk = [&](unsigned blk) -> unsigned { return SPREAD ? (blk >> 2) * (Krow * 4u) + (blk & 3u) * 16u : blk * Krow; };
    const int tid = threadIdx.x;
    const int lane = NW == 1 ? tid : (tid & 63);
    const int wv = NW == 1 ? 0 : __builtin_amdgcn_readfirstlane(tid >> 6);   // wave of the team
    const bool lead = NW == 1 || wv == 0;                                     // the wave that runs the scheduling cycle
    constexpr int TT = 64 * NW;                                                 // threads of the workgroup
    // the leader alone works on LDS it shares with nobody at that moment: LDS operations of ONE wave execute in order
    auto lead_sync = [&]() {
        if constexpr (NW == 1) __syncthreads();
        else { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
    };
    const int s = __builtin_amdgcn_readfirstlane(perm[blockIdx.x]);
    const int n = __builtin_amdgcn_readfirstlane(scen[s].n_nodes);
    // Per-scenario node order (simon_set_node_ranks: the scenario's own nodeTree order): the per-class node lists, a node's index
    // inside its class and the canonical index used by tie-breaks come from the scenario's own arrays; the kernel is otherwise
    // unchanged (positions are class-major in RANK order, so position order inside a class is still canonical order).
    // (the launch passes the scenario-major array and its stride N, or the pool's lists and stride 0: one kernel argument either way)
    // (the launch passes the scenario-major array and its stride N, or the pool's lists: one kernel argument either way)
    constexpr bool ranked = RANKED;
    const unsigned rk_off = RANKED ? (unsigned)s * (unsigned)sc.rk_stride : 0u;
    const int32_t* __restrict__ const cls_list = cls_list_pool;
    static_assert(!LDSWS || (!COARSE && !REST && !SPREAD && !MANY && NW == 1), "the LDS-resident workspace serves generation 4");
    static_assert(!LDSX || (REST && NW == 1 && !LDSWS), "LDS-resident mask rows serve generation 6");
    unsigned char* const wsb = LDSWS ? smem + ((cv.total + 127) & ~127) : ws + ws_off[blockIdx.x];
    const int32_t* __restrict__ order = orders + (size_t)__builtin_amdgcn_readfirstlane(scen[s].order_id) * P;

    // ---- prologue 1: clear, tables -> LDS, class segments --------------------------------------
    for (int i = tid; i < K * Cn; i += TT) { if (CNT_LDS) s_cnt[i] = 0; s_sn[i] = 0; }
    if constexpr (kCls4) { for (int i = tid; i < Cn * 8; i += TT) ((int*)(smem + cv.shape))[i] = ((const int*)shapes)[(i >> 3) * 12 + (i & 7)]; }
    else for (int i = tid; i < Cn * 12; i += TT) ((int*)(smem + cv.shape))[i] = ((const int*)shapes)[i];
    // count of class-d nodes among the first n canonical nodes, padded to 16 (COARSE: to 64, one class per summary entry)
    const int cnt_d = (lane < Cn) ? clsprefix[(size_t)n * Cn + lane] : 0;
    const int pad_d = (cnt_d + (UNIT - 1)) & ~(UNIT - 1);
    int incl = pad_d;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(incl, off, 64);
        if (lane >= off) incl += o;
    }
    int ni = __builtin_amdgcn_readlane(incl, 63);                     // padded scenario size (classes beyond Cn add nothing)
    if (lead && lane < Cn) s_seg[lane] = incl - pad_d;
    // 65 .. 128 internal node classes (round 4; the instantiations without REST rows and SPREAD walks -- the host admits them there
    // only): lane l also owns class 64 + l.  Every per-class quantity of those instantiations lives in LDS / the workspace by class
    // index already; what is lane-shaped is this prefix scan, the count lookup of the prologue and the class terms' re-base.
    int cnt_d2 = 0;
    if (Cn > 64) {
        cnt_d2 = (64 + lane < Cn) ? clsprefix[(size_t)n * Cn + 64 + lane] : 0;
        const int pad2 = (cnt_d2 + (UNIT - 1)) & ~(UNIT - 1);
        int incl2 = pad2;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(incl2, off, 64);
            if (lane >= off) incl2 += o;
        }
        if (lead && 64 + lane < Cn) s_seg[64 + lane] = ni + incl2 - pad2;
        ni += __builtin_amdgcn_readlane(incl2, 63);
        // 129 .. 256 classes (end of round 6, same instantiations): the further groups of 64 extend the scan; their counts are not kept in
        // registers (count_of_class reads the prefix table for them -- the prologue alone asks)
        if constexpr (kCls4) {
#pragma nounroll
        for (int g = 2; g * 64 < Cn; ++g) {
            const int cg = (g * 64 + lane < Cn) ? clsprefix[(size_t)n * Cn + g * 64 + lane] : 0;
            const int padg = (cg + (UNIT - 1)) & ~(UNIT - 1);
            int inclg = padg;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int o = __shfl_up(inclg, off, 64);
                if (lane >= off) inclg += o;
            }
            if (lead && g * 64 + lane < Cn) s_seg[g * 64 + lane] = ni + inclg - padg;
            ni += __builtin_amdgcn_readlane(inclg, 63);
        }
        }
    }
    if (lead && lane == 0) s_seg[Cn] = ni;                            // the sentinel: with Cn == 64 no lane Cn exists to write it
    auto count_of_class = [&](int d) -> int {                        // nodes of class d in this scenario (every lane active: cross-lane reads)
        const int lo = __shfl(cnt_d, d & 63, 64);
        if (Cn <= 64) return lo;
        const int hi = __shfl(cnt_d2, d & 63, 64);
        int r = d < 64 ? lo : hi;
        if (kCls4 && d >= 128) r = clsprefix[(size_t)n * Cn + (d < Cn ? d : 0)];   // (classes 128 ..: from the prefix table)
        return r;
    };
    const int nblk = ni >> 4, nun = ni >> UB;                         // table blocks (16 positions); summary entries
    unsigned char* g_tile = wsb;                                      // [block][K][16] bytes: 0 = infeasible, else 1 + LA + BA
    NodeState* g_state = (NodeState*)(wsb + (((size_t)nblk * Krow + 127) & ~(size_t)127));
    uint2* g_nz = (uint2*)((unsigned char*)g_state + (((size_t)ni * 12 + 127) & ~(size_t)127));   // NonZeroRequested (only when !NZEQ)
    // COARSE: per-16 entries [ni / 64][K][4] u16 and feasible-node counters [K][Cn] behind the node state
    unsigned short* g_fine = (unsigned short*)((unsigned char*)g_nz + (NZEQ ? 0 : (((size_t)ni * 8 + 127) & ~(size_t)127)));
    int* g_cnt = (int*)((unsigned char*)g_fine + (COARSE ? (((size_t)(ni >> 6) * K * 8 + 127) & ~(size_t)127) : (size_t)0));   // (kCls4: the counters alone)
    // REST: position masks [nblk][M], GPU devices by position
    unsigned short* const g_xm_mem = (unsigned short*)((unsigned char*)g_cnt + (((size_t)K * Cn * 4 + 127) & ~(size_t)127));   // (the HBM slice keeps its layout)
    // LDSX: rows [M][nblk] behind tcarve's total, then the row totals [M], then the canonical index of every position
    const unsigned ldsx_x = ((unsigned)cv.total + 127u) & ~127u;
    const unsigned ldsx_rt = ldsx_x + ((((unsigned)sc.ni_max >> 4) * (unsigned)M * 2u + 127u) & ~127u);
    const unsigned ldsx_cn = ldsx_rt + (((unsigned)M * 4u + 127u) & ~127u);
    unsigned short* g_xm = LDSX ? (unsigned short*)(smem + ldsx_x) : g_xm_mem;
    unsigned short* const sx_canon = (unsigned short*)(smem + ldsx_cn);
    unsigned* g_gused = (unsigned*)((unsigned char*)g_xm_mem + (((size_t)nblk * M * 2 + 127) & ~(size_t)127));
    unsigned* g_gtot = g_gused + (size_t)ni * 8;
    int* g_gcnt = (int*)(g_gtot + ni);
    unsigned* g_xused = (unsigned*)(g_gcnt + ni);                     // [ni][8]: Requested ephemeral storage, extended resources
    unsigned* g_xalloc = g_xused + (size_t)ni * 8;                    // [ni][8]: their allocatable
    unsigned short* g_pdom = (unsigned short*)(g_xalloc + (size_t)ni * 8);   // [NZ][ni]: domain under a zone-like key, 0xFFFF = no label
    unsigned* g_rowtot = LDSX ? (unsigned*)(smem + ldsx_rt) : (unsigned*)(g_pdom + (size_t)NZ * ni);       // [M]: pods that set the row so far (term totals of required affinity)
    // SPREAD (without REST: M == 0, the block starts where the REST rows would; with REST: behind the row totals): placed pods a term's selector matches
    unsigned char* g_hrow = RS ? (unsigned char*)(g_pdom + (size_t)NZ * ni) + (((size_t)M * 4 + 127) & ~(size_t)127) : (unsigned char*)g_xm_mem;   // [TH][ni] per position (hostname-like key: domain = node)
    unsigned* g_zcnt = (unsigned*)(g_hrow + (((size_t)TH * ni + 127) & ~(size_t)127));   // [TZ][16] per domain of a zone-like key
    unsigned char* g_hmax = (unsigned char*)(g_zcnt + (((size_t)TZ * 16 + 31) & ~(size_t)31));   // [TH] largest counter of a hostname-key row
    unsigned short* g_canon = (unsigned short*)(g_hmax + (((size_t)TH + 127) & ~(size_t)127));    // [ni] RANKED: rank of the position's node in the scenario's order
    // GPU fold (TableScalars::static_tables & 128; the two-level instantiations without REST rows): the devices of every position behind
    // the scenario's workspace -- used [ni][8], per-device total [ni], device count [ni] (gcd units)
    constexpr bool kGpuFoldable = COARSE && !REST && HAS_PIN;
    const bool gfold = kGpuFoldable && (sc.static_tables & 128);
    unsigned* g_fu = (unsigned*)(wsb + table_ws_of(K, ni, NZEQ, COARSE, Cn, M, NZ, TH, TZ, kCls4));
    unsigned* g_ft = g_fu + (size_t)ni * 8;
    int* g_fc = (int*)(g_ft + ni);
    // [K][nbp] in LDS: (best byte + class term) << 4 | 15 - position of a block of 16; COARSE: ... << 6 | 63 - position of 64 positions
    unsigned short* s_sum = (unsigned short*)(smem + cv.sum);
    for (int i = tid; i < K * nbp / 2; i += TT) ((unsigned*)s_sum)[i] = 0u;
    if (tid == 0 && ((K * nbp) & 1)) s_sum[K * nbp - 1] = 0;
    if (!CNT_LDS) for (int i = tid; i < K * Cn; i += TT) g_cnt[i] = 0;
    if constexpr (NW > 1) __threadfence();                            // (the zeroes are in L2 before another wave's atomics add to them)
    __syncthreads();

    // class of a position (segments are contiguous): number of segment ENDS at or below it
    auto class_of_pos = [&](int p) -> int {
        int d = 0;
        for (int e = 1; e <= Cn; ++e) d += (p >= s_seg[e]) ? 1 : 0;
        return d < Cn ? d : Cn - 1;
    };

    // (feasible, LeastAllocated + BalancedAllocation) of one signature on one node: 0 when NodeResourcesFit fails
    // (fit.go:230-302), else 1 + score.  Same fp64 sequences as simon_fast.hip::eval_slot.
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
        const double cf = div_by_rcp1(r_c, sh.cap_c, sh.rc_c);            // balanced_allocation.go:82-119; operands < 2^31 (simon_device.h)
        const double mf = div_by_rcp1(r_m, sh.cap_m, sh.rc_m);
        const int bs = (int)((1.0 - __builtin_fabs(cf - mf)) * 100.0);
        const int base = ((la_c + la_m) >> 1) + ((ge_c || ge_m) ? 0 : bs);
        return ok ? (unsigned)(base + 1) : 0u;
    };

    // ---- prologue 2: node rows, table and summary; lanes = 64 consecutive positions (4 blocks) ----
    for (int p0 = wv * 64; p0 < ni; p0 += TT) {                          // (team mode: chunk i belongs to wave i % NW)
        const int p = p0 + lane;
        const int d = class_of_pos(p < ni ? p : 0);
        const int r = p - s_seg[d];
        // the cross-lane read runs with EVERY lane active (as the right operand of && it ran only in the lanes with p < ni: in the last,
        // partial chunk the lane holding class d's count could be inactive and read as 0 -- found by tests/fuzz_table.py)
        const int cnt_of_d = count_of_class(d);
        const bool real = p < ni && r < cnt_of_d;
        const int j = real ? cls_list[rk_off + (unsigned)(cls_off[d] + r)] : 0;   // r-th node of class d in canonical order
        const NodeState st = real ? NodeState{i_rq_cpu[j], i_rq_mem[j], (unsigned)(a_pods[j] - i_npods[j])} : NodeState{0, 0, 0};
        if constexpr (SPREAD && RANKED) {                                 // spread_select breaks its ties by rank: one coalesced read per unit
            // (no term with a counter row -- preferred terms whose weights cancel --: the workspace holds no SPREAD block, table_ws_of, and no pod walks)
            if (p < ni && (TH | TZ) != 0) g_canon[p] = (unsigned short)(real ? gp(cold->rk_rank)[(size_t)s * (size_t)cold->N + j] : 8191);
        }
        uint2 z = make_uint2(0, 0);
        if (!NZEQ && real) z = make_uint2(i_nz_cpu[j], i_nz_mem[j]);
        if (p < ni) {
            g_state[p] = st;
            if (!NZEQ) g_nz[p] = z;
        }
        const ShapeRow sh = shape_of(d);
        unsigned char* tp = g_tile + (tile_blk((unsigned)(p >> 4)) + (unsigned)(p & 15));
        unsigned gu[8] = {0, 0, 0, 0, 0, 0, 0, 0}, gtot = 0;                // GPU fold: this position's devices
        int gcnt = 0;
        if constexpr (kGpuFoldable) {
            if (gfold) {
                if (real) {
                    gcnt = gp(cold->gpu_cnt)[j]; gtot = gp(cold->gpu_devtot)[j];
#pragma unroll
                    for (int e = 0; e < 8; ++e) gu[e] = gp(cold->i_gused)[(size_t)j * 8 + e];
                }
                if (p < ni) {
                    g_fc[p] = gcnt; g_ft[p] = gtot;
                    *(uint4*)(g_fu + (size_t)p * 8) = make_uint4(gu[0], gu[1], gu[2], gu[3]);
                    *(uint4*)(g_fu + (size_t)p * 8 + 4) = make_uint4(gu[4], gu[5], gu[6], gu[7]);
                }
            }
        }
        for (int k = 0; k < K; ++k) {
            const SigRow q = sigs[k];
            unsigned b = eval_node(q.req_c, q.req_m, q.nz_c, q.nz_m, q.flags & 1u, (double)st.rq_c, (double)st.rq_m, (double)z.x,
                                   (double)z.y, (int)st.freep, sh);
            b = real ? b : 0u;
            if constexpr (kGpuFoldable) {                                 // a GPU signature starts infeasible where its request does not fit the devices
                if (gfold && q.pad[1] != 0) b = gpu_fits_t(gu, gcnt, gtot, (unsigned)q.pad[0], q.pad[1]) ? b : 0u;
            }
            if (HAS_MASK && static_mask != nullptr) {   // NodeUnschedulable/NodeName/TaintToleration/NodeAffinity: static per (class, node); a prologue-only test:
                                                        // since round 5 every launch takes the HAS_MASK instantiation (half the kernels) and a null pointer says "no mask"
                const uint64_t w = static_mask[(size_t)q.cls * sc.mask_words + (j >> 6)];
                b = ((w >> (j & 63)) & 1ull) ? b : 0u;
            }
            if (p < ni) tp[(unsigned)k * KS] = (unsigned char)b;
            // class term still 0: every signature starts dirty and is re-based at its first use
            const unsigned m16 = row16_max_t(b ? ((b << 4) | (unsigned)(15 - (p & 15))) : 0u);
            if constexpr (COARSE) {                                   // the 64 lanes are ONE summary entry of ONE class
                if ((lane & 15) == 0) g_fine[((size_t)(p0 >> 6) * K + k) * 4 + (lane >> 4)] = (unsigned short)m16;
                const unsigned cand = m16 ? (((m16 & 0xFFF0u) << 2) | ((3u - (unsigned)(lane >> 4)) << 4) | (m16 & 15u)) : 0u;
                const unsigned c64 = wave_max_u32(cand);
                const int nfe = __popcll(__ballot(b != 0));
                if (lane == 0) {
                    s_sum[k * nbp + (p0 >> 6)] = (unsigned short)c64;
                    if (nfe) {
                        if constexpr (CNT_LDS && NW == 1) s_cnt[k * Cn + d] += nfe;   // (REST: one wave, lane 0 alone)
                        else if constexpr (CNT_LDS) atomicAdd(&s_cnt[k * Cn + d], nfe);   // (REST && SPREAD in team mode: chunks of one class on several waves)
                        else if constexpr (NW == 1) g_cnt[k * Cn + d] += nfe;   // lane 0 alone, chunk after chunk: plain (see the refresh)
                        else atomicAdd(&g_cnt[k * Cn + d], nfe);           // chunks of one class on several waves (L1 is invalidated below)
                    }
                }
            } else {
                if ((lane & 15) == 0 && p < ni) s_sum[k * nbp + (p >> 4)] = (unsigned short)m16;
                if constexpr (kCls4) {                                // (performed in L2: L1 is invalidated behind the prologue; 16 positions = one class: one atomic per group)
                    const int n16 = __popc((unsigned)((__ballot(b != 0) >> (lane & 48)) & 0xFFFFull));
                    if ((lane & 15) == 0 && n16) atomicAdd(&g_cnt[k * Cn + d], n16);
                }
                else if (b) atomicAdd(&s_cnt[k * Cn + d], 1);
            }

        }
    }
    if constexpr (REST) {
        for (int i = lane; i < nblk * M; i += 64) g_xm[i] = 0;            // no pod placed yet: every term row clear
        for (int i = lane; i < M; i += 64) g_rowtot[i] = 0u;
        // the pool's GPU devices by position, and the row of every GPU signature: bit set = it does not fit the node now
        GPtr<const int32_t> __restrict__ const gpu_cnt = gp(cold->gpu_cnt);
        GPtr<const uint32_t> __restrict__ const gpu_devtot = gp(cold->gpu_devtot);
        GPtr<const uint32_t> __restrict__ const i_gused = gp(cold->i_gused);
        GPtr<const uint2> __restrict__ const gsig = gp(cold->gsig);
        for (int p0 = 0; p0 < ni; p0 += 64) {
            const int p = p0 + lane;
            const int d = class_of_pos(p);
            const int r = p - s_seg[d];
            const int cnt_of_d = count_of_class(d);
            const bool real = r < cnt_of_d;
            const int j = real ? cls_list[rk_off + (unsigned)(cls_off[d] + r)] : 0;
            const int gc = real ? gpu_cnt[j] : 0;
            const unsigned tot = real ? gpu_devtot[j] : 0u;
            unsigned u[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) u[e] = real ? i_gused[(size_t)j * 8 + e] : 0u;
            if constexpr (LDSX) sx_canon[p] = (unsigned short)(real ? (ranked ? gp(cold->rk_rank)[(size_t)s * (size_t)cold->N + j] : j) : (int)PMASK);   // (the rank with per-scenario node order: ties compare it)
            g_gcnt[p] = gc; g_gtot[p] = tot;
            *(uint4*)(g_gused + (size_t)p * 8) = make_uint4(u[0], u[1], u[2], u[3]);
            *(uint4*)(g_gused + (size_t)p * 8 + 4) = make_uint4(u[4], u[5], u[6], u[7]);
            for (int g = 0; g < G; ++g) {
                const uint2 sg = ldg_u2(gsig + g);
                const bool fits = real && gpu_fits_t(u, gc, tot, sg.x, (int)sg.y);
                const unsigned long long bal = __ballot(!fits);
                if ((lane & 15) == 0) g_xm[(size_t)g * nblk + (p >> 4)] = (unsigned short)((bal >> (lane & 48)) & 0xFFFFull);
            }
            if (X > 0) {                                                  // extra resources of the position, row of every request
                const uint4 a0 = real ? ldg_u4(gp(cold->xalloc) + (size_t)j * 8) : make_uint4(0, 0, 0, 0);
                const unsigned a4 = real ? gp(cold->xalloc)[(size_t)j * 8 + 4] : 0u;
                const uint4 u0 = real ? ldg_u4(gp(cold->i_xused) + (size_t)j * 8) : make_uint4(0, 0, 0, 0);
                const unsigned u4 = real ? gp(cold->i_xused)[(size_t)j * 8 + 4] : 0u;
                *(uint4*)(g_xalloc + (size_t)p * 8) = a0; *(uint4*)(g_xalloc + (size_t)p * 8 + 4) = make_uint4(a4, 0, 0, 0);
                *(uint4*)(g_xused + (size_t)p * 8) = u0; *(uint4*)(g_xused + (size_t)p * 8 + 4) = make_uint4(u4, 0, 0, 0);
                const unsigned al[5] = {a0.x, a0.y, a0.z, a0.w, a4}, us[5] = {u0.x, u0.y, u0.z, u0.w, u4};
                for (int x = 0; x < X; ++x) {
                    const uint4 q0 = ldg_u4(gp(cold->xsig) + (size_t)x * 8);
                    const unsigned rq[5] = {q0.x, q0.y, q0.z, q0.w, gp(cold->xsig)[(size_t)x * 8 + 4]};
                    const bool fits = real && xres_fits_t(rq, us, al);
                    const unsigned long long bal = __ballot(!fits);
                    if ((lane & 15) == 0) g_xm[(size_t)(G + x) * nblk + (p >> 4)] = (unsigned short)((bal >> (lane & 48)) & 0xFFFFull);
                }
            }
            for (int z = 0; z < NZ; ++z) {
                const int dz = real ? gp(cold->zdom)[(size_t)z * cold->N + j] : -1;
                g_pdom[(size_t)z * ni + p] = (unsigned short)(dz >= 0 ? dz : 0xFFFF);
                // the key's label row (the last NZ rows): bit set = the node lacks the label -- required affinity fails there whatever
                // has been placed (filtering.go:357-360)
                const unsigned long long bal = __ballot(dz < 0);
                if ((lane & 15) == 0) g_xm[(size_t)(M - NZ + z) * nblk + (p >> 4)] = (unsigned short)((bal >> (lane & 48)) & 0xFFFFull);
            }
        }
    }
    if constexpr (SPREAD) {
        for (int i = tid; i < (NZK > 0 ? NZK : 1) * Cn; i += TT) s_zdom[i] = NZK > 0 ? gp(cold->cls_zdom)[i] : (signed char)0;
        for (size_t i = tid; i < ((size_t)TH * ni + 3) / 4; i += TT) ((unsigned*)g_hrow)[i] = 0u;     // no pod placed yet
        for (int i = tid; i < TZ * 16; i += TT) g_zcnt[i] = 0u;
        for (int i = tid; i < TH; i += TT) g_hmax[i] = 0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (NW > 1 || kCls4) __threadfence();                   // the counters were added to in L2 (atomics): no stale L1 line may serve the plain loads of the loop
    __syncthreads();
    if constexpr (NW > 1) __threadfence();

    // this lane's signatures: lane l re-evaluates signatures l (and l + 64) on a touched node
    int kk[KQ];
    bool kvalid[KQ];
    double my_req_c[KQ], my_req_m[KQ], my_nz_c[KQ], my_nz_m[KQ];
    bool my_zero[KQ];
    int my_tc[KQ];                                                    // kCls4: the signatures' table class (row of simon_raw)
    unsigned my_add_c[KQ], my_add_m[KQ], my_addz_c[KQ], my_addz_m[KQ], koff[KQ];
    unsigned my_greq[KQ];                                             // GPU fold: this lane's signatures' GPU requests (SigRow::pad)
    int my_gnum[KQ];
    unsigned my_dirty = 0;
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
        kvalid[q] = lane + 64 * q < K;
        kk[q] = kvalid[q] ? lane + 64 * q : 0;
        const SigRow r = sigs[kk[q]];
        my_req_c[q] = r.req_c; my_req_m[q] = r.req_m; my_nz_c[q] = r.nz_c; my_nz_m[q] = r.nz_m;
        my_zero[q] = r.flags & 1u;
        if constexpr (kCls4) my_tc[q] = r.cls;
        my_greq[q] = (unsigned)r.pad[0]; my_gnum[q] = kvalid[q] ? r.pad[1] : 0;
        my_add_c[q] = (unsigned)r.req_c; my_add_m[q] = (unsigned)r.req_m;
        my_addz_c[q] = (unsigned)r.nz_c; my_addz_m[q] = (unsigned)r.nz_m;
        koff[q] = (unsigned)kk[q] * KS;
    }
    for (int j = 0; j * 64 + lane < K; ++j) my_dirty |= 1u << j;      // bit j: signature lane + 64 j starts dirty (first use re-bases it)
    // this lane's summary entries (lane, lane + 64, ...): constant of the arg-max key (low field = PMASK - position), node class,
    // and offset of the entry's class segment into the static per-class node lists (index of position p = boff + p; packed with
    // the class: one readlane fetches both for the winning entry)
    int cb[NBQ], bcls[NBQ], binfo[NBQ];
#pragma unroll
    for (int q = 0; q < NBQ; ++q) {
        const int b = q * 64 + lane;
        cb[q] = (((int)(PMASK >> UB)) - b) << UB;                         // PMASK - position = (last entry - b) << UB | UNIT - 1 - pos
        bcls[q] = class_of_pos(b < nun ? b * UNIT : 0);
        binfo[q] = (cls_off[bcls[q]] - s_seg[bcls[q]] + 8192) | (bcls[q] << 16);
    }

    // (offset into cls_list + 8192) | class << 16 of block bw: both candidate lanes are read, a scalar select picks (no branch)
    auto winner_info = [&](int bw) -> int {
        const int l = bw & 63, h = bw >> 6;
        int info = __builtin_amdgcn_readlane(binfo[0], l);
        if (NBQ > 1) { const int i1 = __builtin_amdgcn_readlane(binfo[NBQ > 1 ? 1 : 0], l); info = h == 1 ? i1 : info; }
        if (NBQ > 2) {
            const int i2 = __builtin_amdgcn_readlane(binfo[NBQ > 2 ? 2 : 0], l), i3 = __builtin_amdgcn_readlane(binfo[NBQ > 3 ? 3 : 0], l);
            info = h == 2 ? i2 : h == 3 ? i3 : info;
        }
        return info;
    };

    // The part of a pod's score that depends on (table class c, node class of lane dd) and on WHICH classes hold a feasible node
    // (`inb`), every lane calling:
    //   2 x SimonPlugin / GpuSharePlugin NormalizeScore (pkg/simulator/plugin/simon.go:76-101: min-max over the feasible nodes)
    //   + NodeAffinity preferred terms   (DefaultNormalizeScore(100, false), helper/normalize_score.go:27-54: 100 raw / max)
    //   + TaintToleration PreferNoSchedule (DefaultNormalizeScore(100, true): 100 - 100 raw / max, 100 everywhere when max = 0)
    //   + already weighted static scores (NodePreferAvoidPods), small ones only (simon_hip.hip: static_tables_fit).
    // floor(100 x / m) = (int)fma(100 x, 1/m, 0.5/m) for 0 <= x <= m < 2^30 (the la_term argument, simon_device.h).
    auto class_term = [&](bool inb, int rawc, int c, int dd) -> int {
        const int lo = wave_min_i32(inb ? rawc : 0x7fffffff);
        const int hi = wave_max_i32(inb ? rawc : (int)0x80000000);
        const int range = hi >= lo ? hi - lo : 0;
        const double rr = range ? 1.0 / (double)range : 0.0;
        int term = (inb && range) ? 2 * (int)__builtin_fma((double)(rawc - lo) * 100.0, rr, 0.5 * rr) : 0;
        GPtr<const TableCold> cc = cold;
        asm volatile("" : "+s"(cc));                                  // rare path: its pointers are fetched here, and only when present
        if (sc.static_tables & 1) {
            const int v = gp(cc->na_raw)[c * Cn + dd];
            const int mx = wave_max_i32(inb ? v : 0);
            const double r = mx ? 1.0 / (double)mx : 0.0;
            term += (inb && mx) ? (int)__builtin_fma((double)v * 100.0, r, 0.5 * r) : 0;
        }
        if (sc.static_tables & 2) {
            const int v = gp(cc->tt_raw)[c * Cn + dd];
            const int mx = wave_max_i32(inb ? v : 0);
            const double r = mx ? 1.0 / (double)mx : 0.0;
            term += inb ? (mx ? 100 - (int)__builtin_fma((double)v * 100.0, r, 0.5 * r) : 100) : 0;
        }
        if (sc.static_tables & 4) term += inb ? gp(cc->add_raw)[c * Cn + dd] : 0;
        return term;
    };
    // class_term for two classes per lane (CN2: lane l holds classes d0 = l and d1 = 64 + l): the same formulas with the extremes and
    // maxima taken over both halves (renormalise's form for more than 64 classes), every lane calling
    auto class_term2 = [&](bool in0, bool in1, int raw0, int raw1, int c, int d0, int d1, int& t0, int& t1) {
        const int lo = min(wave_min_i32(in0 ? raw0 : 0x7fffffff), wave_min_i32(in1 ? raw1 : 0x7fffffff));
        const int hi = max(wave_max_i32(in0 ? raw0 : (int)0x80000000), wave_max_i32(in1 ? raw1 : (int)0x80000000));
        const int range = hi >= lo ? hi - lo : 0;
        const double rr = range ? 1.0 / (double)range : 0.0;
        t0 = (in0 && range) ? 2 * (int)__builtin_fma((double)(raw0 - lo) * 100.0, rr, 0.5 * rr) : 0;
        t1 = (in1 && range) ? 2 * (int)__builtin_fma((double)(raw1 - lo) * 100.0, rr, 0.5 * rr) : 0;
        GPtr<const TableCold> cc = cold;
        if (sc.static_tables & 1) {
            const int a0 = gp(cc->na_raw)[c * Cn + d0], a1 = gp(cc->na_raw)[c * Cn + d1];
            const int mx = max(wave_max_i32(in0 ? a0 : 0), wave_max_i32(in1 ? a1 : 0));
            const double r = mx ? 1.0 / (double)mx : 0.0;
            t0 += (in0 && mx) ? (int)__builtin_fma((double)a0 * 100.0, r, 0.5 * r) : 0;
            t1 += (in1 && mx) ? (int)__builtin_fma((double)a1 * 100.0, r, 0.5 * r) : 0;
        }
        if (sc.static_tables & 2) {
            const int a0 = gp(cc->tt_raw)[c * Cn + d0], a1 = gp(cc->tt_raw)[c * Cn + d1];
            const int mx = max(wave_max_i32(in0 ? a0 : 0), wave_max_i32(in1 ? a1 : 0));
            const double r = mx ? 1.0 / (double)mx : 0.0;
            t0 += in0 ? (mx ? 100 - (int)__builtin_fma((double)a0 * 100.0, r, 0.5 * r) : 100) : 0;
            t1 += in1 ? (mx ? 100 - (int)__builtin_fma((double)a1 * 100.0, r, 0.5 * r) : 100) : 0;
        }
        if (sc.static_tables & 4) { t0 += in0 ? gp(cc->add_raw)[c * Cn + d0] : 0; t1 += in1 ? gp(cc->add_raw)[c * Cn + d1] : 0; }
    };
    // Re-base summary row k after the set of node classes with a feasible node changed.
    auto renormalise = [&](int k, int c) {
        if (kCls4 && Cn > 128) {
            // 129 .. 256 classes: lane l evaluates classes l, 64 + l, 128 + l, 192 + l -- class_term's formulas with the extremes and maxima taken over all four
            // groups.  Every load of a pass is in flight at once (the first form walked the groups in a rolled loop: four dependent L2 round trips per pass, and
            // with 159 classes of three nodes each a signature is re-based a hundred times per scenario -- config 3 on 160 shapes spent a third of its time here).
            int dq[4], cnq[4], rawq[4], tq[4];
            bool inq[4];
            int lo = 0x7fffffff, hi = (int)0x80000000;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const bool v = g * 64 + lane < Cn;
                dq[g] = v ? g * 64 + lane : 0;
                cnq[g] = g_cnt[k * Cn + dq[g]];
                rawq[g] = simon_raw[c * Cn + dq[g]];
                inq[g] = v && cnq[g] > 0;
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) { lo = min(lo, inq[g] ? rawq[g] : 0x7fffffff); hi = max(hi, inq[g] ? rawq[g] : (int)0x80000000); }
            lo = wave_min_i32(lo); hi = wave_max_i32(hi);
            if (lane == 0) s_ext[k] = make_int2(lo, hi);
            const int range = hi >= lo ? hi - lo : 0;
            const double rr = range ? 1.0 / (double)range : 0.0;
#pragma unroll
            for (int g = 0; g < 4; ++g) tq[g] = (inq[g] && range) ? 2 * (int)__builtin_fma((double)(rawq[g] - lo) * 100.0, rr, 0.5 * rr) : 0;
            GPtr<const TableCold> cc = cold;
            asm volatile("" : "+s"(cc));                                  // rare path: its pointers are fetched here, and only when present
            if (sc.static_tables & 1) {
                int a[4], mx = 0;
#pragma unroll
                for (int g = 0; g < 4; ++g) a[g] = gp(cc->na_raw)[c * Cn + dq[g]];
#pragma unroll
                for (int g = 0; g < 4; ++g) mx = max(mx, inq[g] ? a[g] : 0);
                mx = wave_max_i32(mx);
                const double r = mx ? 1.0 / (double)mx : 0.0;
#pragma unroll
                for (int g = 0; g < 4; ++g) tq[g] += (inq[g] && mx) ? (int)__builtin_fma((double)a[g] * 100.0, r, 0.5 * r) : 0;
            }
            if (sc.static_tables & 2) {
                int a[4], mx = 0;
#pragma unroll
                for (int g = 0; g < 4; ++g) a[g] = gp(cc->tt_raw)[c * Cn + dq[g]];
#pragma unroll
                for (int g = 0; g < 4; ++g) mx = max(mx, inq[g] ? a[g] : 0);
                mx = wave_max_i32(mx);
                const double r = mx ? 1.0 / (double)mx : 0.0;
#pragma unroll
                for (int g = 0; g < 4; ++g) tq[g] += inq[g] ? (mx ? 100 - (int)__builtin_fma((double)a[g] * 100.0, r, 0.5 * r) : 100) : 0;
            }
            if (sc.static_tables & 4) {
#pragma unroll
                for (int g = 0; g < 4; ++g) tq[g] += inq[g] ? gp(cc->add_raw)[c * Cn + dq[g]] : 0;
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (g * 64 + lane < Cn) {
                    s_tmp[dq[g]] = tq[g] - (int)s_sn[k * Cn + dq[g]];
                    s_sn[k * Cn + dq[g]] = (unsigned short)tq[g];
                }
            }
        } else if (Cn > 64) {
            // 65 .. 128 classes: lane l evaluates classes l and 64 + l -- class_term's formulas with the extremes taken over both halves
            const bool v1 = 64 + lane < Cn;
            const int d0 = lane, d1 = v1 ? 64 + lane : 0;
            const int cn0 = !CNT_LDS ? g_cnt[k * Cn + d0] : s_cnt[k * Cn + d0], cn1 = v1 ? (!CNT_LDS ? g_cnt[k * Cn + d1] : s_cnt[k * Cn + d1]) : 0;
            const bool in0 = cn0 > 0, in1 = cn1 > 0;
            const int raw0 = simon_raw[c * Cn + d0], raw1 = simon_raw[c * Cn + d1];
            const int lo = min(wave_min_i32(in0 ? raw0 : 0x7fffffff), wave_min_i32(in1 ? raw1 : 0x7fffffff));
            const int hi = max(wave_max_i32(in0 ? raw0 : (int)0x80000000), wave_max_i32(in1 ? raw1 : (int)0x80000000));
            const int range = hi >= lo ? hi - lo : 0;
            const double rr = range ? 1.0 / (double)range : 0.0;
            int t0 = (in0 && range) ? 2 * (int)__builtin_fma((double)(raw0 - lo) * 100.0, rr, 0.5 * rr) : 0;
            int t1 = (in1 && range) ? 2 * (int)__builtin_fma((double)(raw1 - lo) * 100.0, rr, 0.5 * rr) : 0;
            GPtr<const TableCold> cc = cold;
            if (sc.static_tables & 1) {
                const int a0 = gp(cc->na_raw)[c * Cn + d0], a1 = gp(cc->na_raw)[c * Cn + d1];
                const int mx = max(wave_max_i32(in0 ? a0 : 0), wave_max_i32(in1 ? a1 : 0));
                const double r = mx ? 1.0 / (double)mx : 0.0;
                t0 += (in0 && mx) ? (int)__builtin_fma((double)a0 * 100.0, r, 0.5 * r) : 0;
                t1 += (in1 && mx) ? (int)__builtin_fma((double)a1 * 100.0, r, 0.5 * r) : 0;
            }
            if (sc.static_tables & 2) {
                const int a0 = gp(cc->tt_raw)[c * Cn + d0], a1 = gp(cc->tt_raw)[c * Cn + d1];
                const int mx = max(wave_max_i32(in0 ? a0 : 0), wave_max_i32(in1 ? a1 : 0));
                const double r = mx ? 1.0 / (double)mx : 0.0;
                t0 += in0 ? (mx ? 100 - (int)__builtin_fma((double)a0 * 100.0, r, 0.5 * r) : 100) : 0;
                t1 += in1 ? (mx ? 100 - (int)__builtin_fma((double)a1 * 100.0, r, 0.5 * r) : 100) : 0;
            }
            if (sc.static_tables & 4) { t0 += in0 ? gp(cc->add_raw)[c * Cn + d0] : 0; t1 += in1 ? gp(cc->add_raw)[c * Cn + d1] : 0; }
            s_tmp[d0] = t0 - (int)s_sn[k * Cn + d0];
            s_sn[k * Cn + d0] = (unsigned short)t0;
            if (v1) { s_tmp[d1] = t1 - (int)s_sn[k * Cn + d1]; s_sn[k * Cn + d1] = (unsigned short)t1; }
        } else {
        const int dd = lane < Cn ? lane : 0;
        const int cn = (lane < Cn) ? (!CNT_LDS ? g_cnt[k * Cn + dd] : s_cnt[k * Cn + dd]) : 0;
        const bool inb = cn > 0;
        const int rawc = simon_raw[c * Cn + dd];                      // global: this path runs a handful of times per signature
        const int sn = class_term(inb, rawc, c, dd);
        if (lane < Cn) {
            s_tmp[dd] = sn - (int)s_sn[k * Cn + dd];
            s_sn[k * Cn + dd] = (unsigned short)sn;
        }
        }
        lead_sync();
        unsigned short* srow = s_sum + k * nbp;
#pragma unroll
        for (int q = 0; q < NBQ; ++q) {
            const int b = q * 64 + lane;
            if (b < nun) {
                const int delta = s_tmp[bcls[q]];
                const unsigned m = srow[b];
                if (m) srow[b] = (unsigned short)((int)m + delta * UNIT);
                if constexpr (COARSE) {                               // the four per-16 entries behind this entry (same class)
                    uint2* fp = (uint2*)(g_fine + ((size_t)b * K + k) * 4);
                    uint2 f = *fp;
                    auto shift2 = [&](unsigned w) -> unsigned {         // both u16 halves: non-zero entries move by delta << 4
                        const unsigned lo = w & 0xFFFFu, hi = w >> 16;
                        return ((lo ? (unsigned)((int)lo + delta * 16) : 0u) & 0xFFFFu) | ((hi ? (unsigned)((int)hi + delta * 16) : 0u) << 16);
                    };
                    f.x = shift2(f.x); f.y = shift2(f.y);
                    *fp = f;
                }
            }
        }
        lead_sync();
    };

    TPROF_DECL
    // ---- REST path -------------------------------------------------------------------------------------------------------
    // A REST pod's descriptor (PodRowC::rest) = GPU request + 1 | extra-resource request + 1 << 6 | rows << 12 | offset << 18 into xrows: entry e < rows packs the
    // mask row the pod must find clear (low half) and the row it sets when it lands (high half).  Lane e holds entry e (`rowv`,
    // ONE vector load per pod, issued when the cycle starts); everything else about the pod's filters is register traffic.
    const uint2 my_gsig = REST ? ldg_u2(gp(cold->gsig) + (lane < G ? lane : 0)) : make_uint2(0, 0);    // lane g: GPU signature g
    unsigned my_xsig[5] = {0, 0, 0, 0, 0};                                              // lane 32 + x: extra-resource request x
    if (REST && X > 0) {
        const int x = (lane >= 32 && lane - 32 < X) ? lane - 32 : 0;
        const uint4 q0 = ldg_u4(gp(cold->xsig) + (size_t)x * 8);
        my_xsig[0] = q0.x; my_xsig[1] = q0.y; my_xsig[2] = q0.z; my_xsig[3] = q0.w; my_xsig[4] = gp(cold->xsig)[(size_t)x * 8 + 4];
    }
    // OR of the pod's filter rows for block b (lane-varying b < nblk), all loads independent
    // An entry with bit 31 is a required-affinity term: its row must be SET (a pod matching the term sits in the node's domain) --
    // unless the first-pod escape holds (`esc`: no pod anywhere matches any of the class's terms and the class matches them itself,
    // filtering.go:361-374), which switches those entries off.
    auto excluded = [&](int b, int nrows, int rowv, int gs, int xs, bool esc) -> unsigned {
        const unsigned short* xr = g_xm + b;                          // row r of block b: xr[r * nblk]
        unsigned bad = gs >= 0 ? (unsigned)xr[(size_t)gs * nblk] : 0u;
        if (xs >= 0) bad |= (unsigned)xr[(size_t)(G + xs) * nblk];
        for (int e = 0; e < nrows; ++e) {
            const unsigned ent = (unsigned)__builtin_amdgcn_readlane(rowv, e);
            const unsigned v = (unsigned)xr[(size_t)(ent & 0xFFFFu) * nblk];
            bad |= (AFF && (ent >> 31)) ? (esc ? 0u : (~v & 0xFFFFu)) : v;
        }
        return bad;
    };
    // the escape of a pod's required-affinity entries: lane e holds the total of entry e's row (`rtv`, loaded with `rowv`)
    auto aff_escape = [&](int nrows, int rowv, unsigned rtv) -> bool {
        if (!AFF) return false;
        const bool inv = lane < nrows && ((unsigned)rowv >> 31);
        const bool self = __ballot(inv && (((unsigned)rowv >> 27) & 1u)) != 0ull;
        return self && __ballot(inv && rtv != 0u) == 0ull;
    };
    // findNodesThatFitPod + prioritizeNodes + selectHost of a REST pod: returns the position (-1: no node), sets dstar / res.
    //
    // Summary first.  s_sum[k][u] already names the best table byte of unit u (64 positions) and WHERE it sits -- the first maximum in
    // position order.  If that position is not excluded by the pod's filter rows it is also the best of the unit's admitted positions
    // (an admitted earlier position with the same byte would have been the summary's choice), so a unit costs one LDS read and one
    // 64-bit word per filter row (rows are stored row-major: a lane's words of one row are contiguous -- one cache line per 16
    // units instead of one per block).  Only a unit whose best position IS excluded (and which is not excluded as a whole: a GPU
    // request on the units of a class without devices) reads its four table rows and looks for the best admitted byte.  The class
    // term folded into the summary entry is taken out again (s_sn: whatever is folded in right now, current or not): the term of a
    // REST pod is normalised over the classes that keep a node under ALL filters (simon.go:76-101), which is computed below.
    auto rest_select = [&](int k, int tc, int nrows, int rowv, int gs, int xs, bool esc, int& dstar, int& res) -> int {
        TPROF(11);                                                         // REST: pod row, descriptor
        const int dd = lane < Cn ? lane : 0;
        const int rawc = simon_raw[tc * Cn + dd];                         // needed after the scan: in flight meanwhile
        if (lane < Cn) s_tmp[lane] = 0;
        const bool v2 = CN2 && 64 + lane < Cn;                            // CN2: this lane's second class, 64 + lane
        const int dd2 = v2 ? 64 + lane : 0;
        int rawc2 = 0;
        if constexpr (CN2) {
            rawc2 = simon_raw[tc * Cn + dd2];
            if (v2) s_tmp[dd2] = 0;
        }
        __builtin_amdgcn_wave_barrier();
        const unsigned short* srow = s_sum + k * nbp;
        const uint2* xw = (const uint2*)g_xm;                             // [row][nun] words of 64 positions
        uint2 badw[NBQ];
        unsigned e64[NBQ], snu[NBQ];
        unsigned uu[NBQ];
#pragma unroll
        for (int q = 0; q < NBQ; ++q) {
            const int u = q * 64 + lane;
            uu[q] = (unsigned)(u < nun ? u : 0);
            badw[q] = gs >= 0 ? xw[(unsigned)gs * (unsigned)nun + uu[q]] : make_uint2(0u, 0u);
            if (xs >= 0) { const uint2 v = xw[(unsigned)(G + xs) * (unsigned)nun + uu[q]]; badw[q].x |= v.x; badw[q].y |= v.y; }
            e64[q] = u < nun ? (unsigned)srow[u] : 0u;
            snu[q] = (unsigned)s_sn[k * Cn + bcls[q]];
        }
        for (int e = 0; e < nrows; ++e) {
            const unsigned ent = (unsigned)__builtin_amdgcn_readlane(rowv, e);
            const unsigned row = (ent & 0xFFFFu) * (unsigned)nun;
            if (AFF && (ent >> 31)) {                                     // required affinity: the row must be set (see excluded)
                if (!esc) {
#pragma unroll
                    for (int q = 0; q < NBQ; ++q) { const uint2 v = xw[row + uu[q]]; badw[q].x |= ~v.x; badw[q].y |= ~v.y; }
                }
            } else {
#pragma unroll
                for (int q = 0; q < NBQ; ++q) { const uint2 v = xw[row + uu[q]]; badw[q].x |= v.x; badw[q].y |= v.y; }
            }
        }
        TPROF_WAIT_MEM; TPROF_WAIT_LDS; TPROF(12);                        // REST: filter words and summary entries arrived
        // candidate of a unit: byte << 6 | 63 - position inside the unit (0: nothing admitted), class term taken out
        unsigned cand[NBQ];
        bool need[NBQ];
        bool any_need = false;
#pragma unroll
        for (int q = 0; q < NBQ; ++q) {
            const unsigned long long bad = ((unsigned long long)badw[q].y << 32) | badw[q].x;
            const unsigned pos = 63u - (e64[q] & 63u);
            const bool hit = (bad >> pos) & 1ull;
            cand[q] = (e64[q] != 0u && !hit) ? e64[q] - (snu[q] << 6) : 0u;
            need[q] = e64[q] != 0u && hit && bad != ~0ull;
            any_need = any_need || need[q];
        }
        TPROF(13);                                                         // REST: candidates from the summaries
#ifdef SIMON_TABLE_PROFILE
        if (__ballot(any_need)) tp_acc[20] += 1;                           // how often table rows are needed
#endif
        if (__ballot(any_need)) {                                         // some unit's best position is excluded: its 4 x 16 table bytes
            const unsigned koff16 = (unsigned)k * KS;                       // (SPREAD keeps [unit][K][64]: tile_blk)
            auto keep = [&](unsigned w, unsigned nib) -> unsigned { return w & ~((((nib & 15u) * 0x00204081u) & 0x01010101u) * 0xFFu); };
#pragma unroll
            for (int q = 0; q < NBQ; ++q) {
                if (!need[q]) continue;
                uint4 R[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) R[j] = *(const uint4*)(g_tile + (tile_blk(uu[q] * 4u + (unsigned)j) + koff16));
                unsigned best = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned b16 = ((j & 2) ? badw[q].y : badw[q].x) >> ((j & 1) * 16);
                    R[j].x = keep(R[j].x, b16); R[j].y = keep(R[j].y, b16 >> 4); R[j].z = keep(R[j].z, b16 >> 8); R[j].w = keep(R[j].w, b16 >> 12);
                    const unsigned e16 = block_key16_t(R[j]);             // best byte << 4 | 15 - position inside the block
                    if (e16 >> 4) best = max(best, ((e16 >> 4) << 6) | ((3u - (unsigned)j) << 4) | (e16 & 15u));
                }
                cand[q] = best;
            }
        }
        TPROF_WAIT_MEM; TPROF(14);                                         // REST: table rows of the units whose best position is excluded
#pragma unroll
        for (int q = 0; q < NBQ; ++q)
            if (cand[q]) {                                                // best node of the class: highest base score, first position
                const unsigned key = ((cand[q] >> 6) << KB) | (PMASK - (uu[q] * 64u + 63u - (cand[q] & 63u)));
                atomicMax((unsigned*)&s_tmp[bcls[q]], key);
            }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const unsigned cbest = lane < Cn ? (unsigned)s_tmp[dd] : 0u;
        const bool present = cbest != 0u;
        TPROF_WAIT_LDS; TPROF(15);                                        // REST: per-class best (LDS max), read back
        if constexpr (CN2) {
            // 65 .. 128 node classes: the same per-class steps on both halves (lane l: classes l and 64 + l); the class term as renormalise forms it
            const unsigned cbest2 = v2 ? (unsigned)s_tmp[dd2] : 0u;
            const bool present2 = cbest2 != 0u;
            if (!__ballot(present || present2)) return -1;
            const int pos = (int)(PMASK - (cbest & PMASK)), pos2 = (int)(PMASK - (cbest2 & PMASK));
            const int idx = cls_off[dd] - s_seg[dd] + pos, idx2 = cls_off[dd2] - s_seg[dd2] + pos2;
            int sn, sn2;
            class_term2(present, present2, rawc, rawc2, tc, dd, dd2, sn, sn2);
            const unsigned total = present ? (cbest >> KB) + (unsigned)sn : 0u, total2 = present2 ? (cbest2 >> KB) + (unsigned)sn2 : 0u;
            const unsigned tmax = max(wave_max_u32(total), wave_max_u32(total2));
            const bool t0 = present && total == tmax, t1 = present2 && total2 == tmax;
            unsigned long long tied0 = __ballot(t0), tied1 = __ballot(t1);
            if (__builtin_expect(__popcll(tied0) + __popcll(tied1) > 1, 0)) {   // several classes reach the maximum: first in canonical order
                int canon0 = t0 ? cls_list[rk_off + (unsigned)idx] : (int)PMASK, canon1 = t1 ? cls_list[rk_off + (unsigned)idx2] : (int)PMASK;
                if (ranked && t0) canon0 = gp(cold->rk_rank)[(size_t)s * (size_t)cold->N + canon0];
                if (ranked && t1) canon1 = gp(cold->rk_rank)[(size_t)s * (size_t)cold->N + canon1];
                const unsigned cmin = max(wave_max_u32(t0 ? PMASK - (unsigned)canon0 : 0u), wave_max_u32(t1 ? PMASK - (unsigned)canon1 : 0u));
                tied0 = __ballot(t0 && PMASK - (unsigned)canon0 == cmin); tied1 = __ballot(t1 && PMASK - (unsigned)canon1 == cmin);
            }
            const bool half = tied0 == 0ull;
            const int wl = __builtin_ctzll(half ? tied1 : tied0);
            const int r0 = __builtin_amdgcn_readlane(idx, wl), r1 = __builtin_amdgcn_readlane(idx2, wl);
            const int p0 = __builtin_amdgcn_readlane(pos, wl), p1 = __builtin_amdgcn_readlane(pos2, wl);
            dstar = half ? 64 + wl : wl;
            res = half ? r1 : r0;
            return half ? p1 : p0;
        }
        if (!__ballot(present)) return -1;
        const int pos = (int)(PMASK - (cbest & PMASK));
        const int idx = cls_off[dd] - s_seg[dd] + pos;                    // index into the static per-class node lists
        // the class term over the classes that hold a feasible node (as renormalise does for the summaries)
        const int sn = class_term(present, rawc, tc, dd);
        TPROF(16);                                                         // REST: class term over the classes present
        const unsigned total = present ? (cbest >> KB) + (unsigned)sn : 0u;   // >= 1 when present (the byte is 1 + score)
        const unsigned tmax = wave_max_u32(total);
        unsigned long long tied = __ballot(present && total == tmax);
        int wl = __builtin_ctzll(tied);
        if (__builtin_expect((tied & (tied - 1)) != 0, 0)) {              // several classes reach the maximum: first in canonical order
            int canon;
            if constexpr (LDSX) canon = (present && total == tmax) ? (int)sx_canon[pos] : (int)PMASK;
            else {
            canon = (present && total == tmax) ? cls_list[rk_off + (unsigned)idx] : (int)PMASK;
            if (ranked && present && total == tmax) canon = gp(cold->rk_rank)[(size_t)s * (size_t)cold->N + canon];
            }
            const unsigned cmin = wave_max_u32((present && total == tmax) ? PMASK - (unsigned)canon : 0u);
            wl = __builtin_ctzll(__ballot(present && total == tmax && PMASK - (unsigned)canon == cmin));
        }
        TPROF(17);                                                         // REST: totals, maximum, tie
        dstar = wl;
        res = __builtin_amdgcn_readlane(idx, wl);
        return __builtin_amdgcn_readlane(pos, wl);
    };
    // What assume adds for a REST pod landing on position pstar: its term rows, the GPU commit and the GPU rows of that node.
    // Two halves: the loads go out with the table-row loads of the cycle, the updates follow the evaluation.
    struct RestLoads { unsigned xr_set, rt_set, xr_g, xr_x, tot, xu4, xa4; int gc; uint4 ua, ub, xu, xa; };
    auto rest_assume_load = [&](int pstar, int nrows, int rowv, int gs, int xs) -> RestLoads {
        RestLoads L{};
        const unsigned short* xr = g_xm + (pstar >> 4);               // row r of the touched block: xr[r * nblk]
        if (lane < nrows) { L.xr_set = xr[(size_t)(((unsigned)rowv >> 16) & 0x7FFu) * nblk]; if (AFF) L.rt_set = g_rowtot[((unsigned)rowv >> 16) & 0x7FFu]; }
        if (gs >= 0) {
            L.gc = g_gcnt[pstar];
            L.tot = g_gtot[pstar];
            L.ua = *(const uint4*)(g_gused + (size_t)pstar * 8);
            L.ub = *(const uint4*)(g_gused + (size_t)pstar * 8 + 4);
            if (lane < G) L.xr_g = xr[(size_t)lane * nblk];
        }
        if (xs >= 0) {
            L.xu = *(const uint4*)(g_xused + (size_t)pstar * 8); L.xu4 = g_xused[(size_t)pstar * 8 + 4];
            L.xa = *(const uint4*)(g_xalloc + (size_t)pstar * 8); L.xa4 = g_xalloc[(size_t)pstar * 8 + 4];
            if (lane >= 32 && lane - 32 < X) L.xr_x = xr[(size_t)(G + lane - 32) * nblk];
        }
        return L;
    };
    auto rest_assume_store = [&](const RestLoads& L, int pstar, int nrows, int rowv, int gs, int xs, int step) {
        const unsigned bit = 1u << (pstar & 15);
        unsigned short* xr = g_xm + (pstar >> 4);
        // node-level term: the pod's own position; a term on a zone-like key marks every position of the pod's domain (below)
        if (lane < nrows && (((unsigned)rowv >> 28) & 7u) == 0u) {
            xr[(size_t)(((unsigned)rowv >> 16) & 0x7FFu) * nblk] = (unsigned short)(L.xr_set | bit);
            if (AFF) g_rowtot[((unsigned)rowv >> 16) & 0x7FFu] = L.rt_set + 1u;   // term total (several entries never share a counted row)
        }
        if (NZ > 0 && __ballot(lane < nrows && (((unsigned)rowv >> 28) & 7u) != 0u)) {
            for (int e = 0; e < nrows; ++e) {
                const unsigned ent = (unsigned)__builtin_amdgcn_readlane(rowv, e);
                if (((ent >> 28) & 7u) == 0u) continue;
                const unsigned short* pd = g_pdom + (size_t)(((ent >> 28) & 7u) - 1u) * ni;
                const unsigned z = __builtin_amdgcn_readfirstlane((unsigned)pd[pstar]);
                if (z == 0xFFFFu) continue;                               // the node lacks the label: nothing is counted (:133-148)
                const unsigned srow = (ent >> 16) & 0x7FFu;
                if (AFF && lane == 0) g_rowtot[srow] += 1u;
                const unsigned zz = z * 0x00010001u;
                for (int b0 = 0; b0 < nblk; b0 += 64) {
                    const int b = b0 + lane;
                    if (b < nblk) {
                        const uint4 d0 = *(const uint4*)(pd + (size_t)b * 16), d1 = *(const uint4*)(pd + (size_t)b * 16 + 8);
                        const unsigned w[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
                        unsigned m = 0;
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const unsigned x = w[i] ^ zz;                 // a zero half = a position of domain z
                            m |= ((x & 0xFFFFu) == 0u ? 1u : 0u) << (2 * i) | ((x >> 16) == 0u ? 2u : 0u) << (2 * i);
                        }
                        if (m) { unsigned short* q = g_xm + (size_t)srow * nblk + b; *q = (unsigned short)(*q | m); }
                    }
                }
            }
        }
        if (gs >= 0) {
            const unsigned greq = (unsigned)__builtin_amdgcn_readlane((int)my_gsig.x, gs);
            const int gnum = __builtin_amdgcn_readlane((int)my_gsig.y, gs);
            unsigned u[8] = {L.ua.x, L.ua.y, L.ua.z, L.ua.w, L.ub.x, L.ub.y, L.ub.z, L.ub.w};
            const unsigned long long booked = gpu_commit_t(u, L.gc, L.tot, greq, gnum);   // Reserve (open-gpu-share.go:147-188), every lane alike
            if (__builtin_expect(sc.static_tables & 8, 0)) {              // the caller wants the devices (simon_batch_out.gpu_slices), by pod id
                const int pid = __builtin_amdgcn_readfirstlane(order[step]);
                if (lane == 0) gp(cold->gpu_slices)[(size_t)s * (size_t)P + (size_t)pid] = booked;
            }
            if (lane == 0) {
                *(uint4*)(g_gused + (size_t)pstar * 8) = make_uint4(u[0], u[1], u[2], u[3]);
                *(uint4*)(g_gused + (size_t)pstar * 8 + 4) = make_uint4(u[4], u[5], u[6], u[7]);
            }
            if (lane < G) {                                               // lane g: does GPU signature g still fit this node?
                const bool fits = gpu_fits_t(u, L.gc, L.tot, my_gsig.x, (int)my_gsig.y);
                xr[(size_t)lane * nblk] = (unsigned short)(fits ? (L.xr_g & ~bit) : (L.xr_g | bit));
            }
        }
        if (xs >= 0) {                                                    // Requested += the pod's ephemeral storage / extended resources
            unsigned us[5] = {L.xu.x, L.xu.y, L.xu.z, L.xu.w, L.xu4};
            const unsigned al[5] = {L.xa.x, L.xa.y, L.xa.z, L.xa.w, L.xa4};
#pragma unroll
            for (int r = 0; r < 5; ++r) us[r] += (unsigned)__builtin_amdgcn_readlane((int)my_xsig[r], 32 + xs);
            if (lane == 0) {
                *(uint4*)(g_xused + (size_t)pstar * 8) = make_uint4(us[0], us[1], us[2], us[3]);
                g_xused[(size_t)pstar * 8 + 4] = us[4];
            }
            if (lane >= 32 && lane - 32 < X) {                            // lane 32 + x: does request x still fit this node?
                const bool fits = xres_fits_t(my_xsig, us, al);
                xr[(size_t)(G + lane - 32) * nblk] = (unsigned short)(fits ? (L.xr_x & ~bit) : (L.xr_x | bit));
            }
        }
    };

    // ---- SPREAD path (generation 7) ---------------------------------------------------------------------------------------
    // PodTopologySpread, soft (ScheduleAnyway) constraints: PreScore + Score + NormalizeScore (podtopologyspread/scoring.go:60-256).
    // raw(node) = int64(sum_e count_e[domain_e(node)] * log(size_e + 2) + (maxSkew_e - 1)), score = 100 (max + min - raw) / max over the
    // feasible nodes that carry every constraint key (the others score 0), weight 2.  The count of a constraint on a hostname-like key
    // is a byte per position (g_hrow), on a zone-like key a counter per domain (g_zcnt) -- and the internal node classes are split by
    // zone domain (host), so a summary unit has ONE zone term.  size_e = scored nodes (hostname) or distinct zones among them: both
    // follow from the feasible-node counters per (signature, class).  The raw score changes on every node of a zone per placement, so
    // the summaries cannot carry it: a spread pod scans its signature's row, one position per lane, 64 positions per step, twice
    // (minimum / maximum of the raw scores, then totals), and takes the first maximum in canonical order (the static per-class node
    // lists give the canonical index of a position).  Everything else about the cycle -- assume, column refresh, summaries -- is the
    // table path's.  Descriptor (PodRowC::rest) = soft constraints | counted terms << 3 | preferred-term entries << 10 | hard constraints << 13 | offset << 15 into TableCold::sp_ent; lane e
    // holds entry e (`spv`).
    // (REST && SPREAD: f_* = the pod's REST descriptor -- GPU request row, extra-resource row, filter entries in the lanes of f_rowv; a pod that
    // carries any walks under them: a position a row excludes counts as infeasible -- in the sizes, the extremes and the totals)
    auto spread_select = [&](int k, int tc, int soft_n, int match_n, int ipa_n, int hard_n, int spv, int spt, int& dstar, int& res,
                             int f_nrows = 0, int f_rowv = 0, int f_gs = -1, int f_xs = -1, bool f_esc = false) -> int {
        // team mode: everything the leader stored in the cycles since the last spread pod -- table bytes, counters, class terms -- is
        // complete (the barrier's release waits for vmcnt / lgkmcnt) before a helper reads it; wave w walks units [ulo, uhi)
        if constexpr (NW > 1) __syncthreads();
        if constexpr (RS && AFF && NW > 1) {
            // a helper read the totals of its required-affinity rows when it reached this pod -- ahead of the leader's assumes: again, behind the barrier
            const unsigned rt2 = (lane < f_nrows && ((unsigned)f_rowv >> 31)) ? g_rowtot[(unsigned)f_rowv & 0xFFFFu] : 0u;
            f_esc = aff_escape(f_nrows, f_rowv, rt2);
        }
        int ulo = 0, uhi = nun;
        if constexpr (NW > 1) {
            const int per = (nun + NW - 1) / NW;
            ulo = min(wv * per, nun); uhi = min(ulo + per, nun);
        }
        const bool have = NW == 1 || ulo < uhi;                           // (a scenario with fewer units than waves leaves some without a share)
        auto team_extremes = [&](int& pmin, int& pmax, int& imin, int& imax) {   // wave-reduced extremes of pass 1, combined over the team
            if constexpr (NW > 1) {
                if (lane == 0) { s_xch[wv * 4 + 0] = pmin; s_xch[wv * 4 + 1] = pmax; s_xch[wv * 4 + 2] = imin; s_xch[wv * 4 + 3] = imax; }
                __syncthreads();
#pragma unroll
                for (int w = 0; w < NW; ++w) {
                    pmin = min(pmin, s_xch[w * 4 + 0]); pmax = max(pmax, s_xch[w * 4 + 1]);
                    imin = min(imin, s_xch[w * 4 + 2]); imax = max(imax, s_xch[w * 4 + 3]);
                }
                pmin = __builtin_amdgcn_readfirstlane(pmin); pmax = __builtin_amdgcn_readfirstlane(pmax);
                imin = __builtin_amdgcn_readfirstlane(imin); imax = __builtin_amdgcn_readfirstlane(imax);
            }
        };
        const int dd = lane < Cn ? lane : 0;
        const bool v2 = CN2 && 64 + lane < Cn;                            // CN2: this lane's second class, 64 + lane
        const int dd2 = v2 ? 64 + lane : 0;
        int kind[4], rowi[4], zsl[4], skew[4];
        bool dup[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int ent = __builtin_amdgcn_readlane(spv, e);
            const int ti = e < soft_n ? __builtin_amdgcn_readlane(spt, e) : 0;   // (the term's row travels with the entry: no dependent load)
            kind[e] = ti & 3; rowi[e] = (ti >> 2) & 0x3FFF; zsl[e] = (ti >> 16) & 7;
            skew[e] = (ent >> 16) & 0x3FFF; dup[e] = (ent >> 30) & 1;
        }
        TPROF(21);                                                         // spread: the pod's entries arrived (loop top -> here)
        int nh = 0, eh = 0;                                               // hostname-like constraints (each has its own counter row)
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (e < soft_n && kind[e] == 1) { ++nh; eh = e; }
        // InterPodAffinity preferred terms in self-referential form (simon_hip.hip: spread_supported): raw = Wh x count of the class's
        // hostname-like term (the row its spread constraint counts, if it has one) + sum of Wz x count of a zone-like term.  Entry i sits
        // in lane soft_n + match_n + i: the hostname-like term first.
        // The code lives in its own instantiations (kIpa = SPREAD && AFF: launched for problems with such terms), so that the others
        // keep their registers.
        constexpr bool kIpa = SPREAD && AFF;
        int wI[4] = {0, 0, 0, 0}, kI[4] = {0, 0, 0, 0}, rI[4] = {0, 0, 0, 0}, zI[4] = {0, 0, 0, 0};
        if constexpr (kIpa) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int l = (soft_n + match_n + i) & 63;
                const int w_ = __builtin_amdgcn_readlane(spv, l), t_ = __builtin_amdgcn_readlane(spt, l);
                wI[i] = i < ipa_n ? w_ : 0; kI[i] = i < ipa_n ? (t_ & 3) : 0; rI[i] = (t_ >> 2) & 0x3FFF; zI[i] = (t_ >> 16) & 7;
            }
        }
        // Hard (DoNotSchedule) constraints on zone-like keys (podtopologyspread/filtering.go:198-333): the node classes are split by zone, so
        // the filter is a per-CLASS verdict -- the class's zone must carry the label and count + self - min over the registered zones
        // must not exceed maxSkew -- and an excluded class simply has no feasible node for this pod.  Entry i sits behind the preferred
        // entries: skew | self << 14, the term's row.
        int hS[3] = {0, 0, 0}, hR[3] = {0, 0, 0}, hZ[3] = {0, 0, 0};
        if constexpr (kIpa) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int l = (soft_n + match_n + ipa_n + i) & 63;
                const int x_ = __builtin_amdgcn_readlane(spv, l), t_ = __builtin_amdgcn_readlane(spt, l);
                hS[i] = i < hard_n ? x_ : 0; hR[i] = (t_ >> 2) & 0x3FFF; hZ[i] = (t_ >> 16) & 7;
            }
        }
        const bool ipa_h = kIpa && ipa_n > 0 && kI[0] == 1;
        const int Wh = ipa_h ? wI[0] : 0;
        const bool has_hrow = nh == 1 || (nh == 0 && ipa_h);             // ONE per-node counter (host: the same term when both exist)
        const int hrow_i = nh == 1 ? rowi[eh] : rI[0];
        // SIMPLE: at most ONE per-node term, and it comes first -- [hostname, zone] (the system defaults, plugin.go:39-50), [hostname],
        // or zone-like constraints alone (their sum is a per-class value): raw = int64((count * w + c) + zone_term(class)).
        const bool simple = nh == 0 || (nh == 1 && eh == 0 && soft_n <= 2);
        const unsigned toff = (unsigned)k * KS + (unsigned)lane;         // this lane's byte of a unit's row of signature k ([unit][K][64])
        // (team mode: no register budget to keep -- a wave's share of the units in as few batches of loads as its size suggests)
        // (team mode keeps the batch sizes: 10 / 8 units per batch measured SLOWER than 6 -- a wave's share is 6 .. 20 units and the
        // slots beyond it repeat the last unit; profiles/r04/r04b_team_ab_*.txt)
        constexpr int SB = kSpreadBatch1, SC = kSpreadBatch2, SG = kSpreadBatch4;
        // the first loads of the walk go out before the class bookkeeping below waits for its own (memory is served in order)
        const bool filt = RS && (f_gs >= 0 || f_xs >= 0 || f_nrows > 0);
        const uint2* const fxw = (const uint2*)g_xm;                      // RS: [row][nun] words of 64 positions, bit = excluded
        auto bad_of = [&](int u) -> unsigned long long {                  // the pod's filter rows ORed for unit u (uniform address: one line per row)
            uint2 b = f_gs >= 0 ? fxw[(unsigned)f_gs * (unsigned)nun + (unsigned)u] : make_uint2(0u, 0u);
            if (f_xs >= 0) { const uint2 v = fxw[(unsigned)(G + f_xs) * (unsigned)nun + (unsigned)u]; b.x |= v.x; b.y |= v.y; }
            for (int e = 0; e < f_nrows; ++e) {
                const unsigned ent = (unsigned)__builtin_amdgcn_readlane(f_rowv, e);
                const uint2 v = fxw[(ent & 0xFFFFu) * (unsigned)nun + (unsigned)u];
                if (AFF && (ent >> 31)) { if (!f_esc) { b.x |= ~v.x; b.y |= ~v.y; } }   // required affinity: the row must be SET, unless the first-pod escape holds (excluded)
                else { b.x |= v.x; b.y |= v.y; }
            }
            return ((unsigned long long)b.y << 32) | b.x;
        };
        int cntd_f = 0;
        if constexpr (RS) {
            if (filt) {
                // the feasible-node counters per (signature, class) know nothing of the rows: count the admitted positions of every class first
                // (F, the zone sizes and the Simon class term are taken over the nodes that pass ALL filters, generic_scheduler.go:131-209)
                constexpr int PB = 8;
                for (int u0 = ulo; u0 < uhi; u0 += PB) {
                    unsigned b0[PB];
                    unsigned long long x0[PB];
#pragma unroll
                    for (int j = 0; j < PB; ++j) {
                        const int u = min(u0 + j, uhi - 1);
                        b0[j] = g_tile[(unsigned)u * (Krow * 4u) + toff];
                        x0[j] = bad_of(u);
                    }
#pragma unroll
                    for (int j = 0; j < PB; ++j) {
                        const int n_ok = __popcll(__ballot(b0[j] != 0u && !((x0[j] >> lane) & 1ull)));
                        const int c = winner_info(min(u0 + j, uhi - 1)) >> 16;
                        cntd_f += (u0 + j < uhi && lane == c) ? n_ok : 0;
                    }
                }
            }
        }
        if constexpr (RS && NW > 1) {
            if (filt) {                                                   // the team's counts: every wave's share through the (still unused) score table
                s_tab[wv * 64 + lane] = cntd_f;
                __syncthreads();
                cntd_f = 0;
#pragma unroll
                for (int w = 0; w < NW; ++w) cntd_f += s_tab[w * 64 + lane];
                __syncthreads();
            }
        }
        const int cntd = filt ? cntd_f : (lane < Cn ? (CNT_LDS ? s_cnt : g_cnt)[k * Cn + dd] : 0);   // feasible nodes of class d for signature k
        unsigned czv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            czv[e] = 0u;
            if (lane < Cn && e < soft_n && kind[e] == 2) {
                const int zd = s_zdom[zsl[e] * Cn + dd];
                czv[e] = zd >= 0 ? g_zcnt[rowi[e] * 16 + zd] : 0u;
            }
        }
        int cntd2 = 0;
        unsigned czv2[4] = {0u, 0u, 0u, 0u};
        if constexpr (CN2) {
            cntd2 = v2 ? g_cnt[k * Cn + dd2] : 0;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (v2 && e < soft_n && kind[e] == 2) {
                    const int zd = s_zdom[zsl[e] * Cn + dd2];
                    czv2[e] = zd >= 0 ? g_zcnt[rowi[e] * 16 + zd] : 0u;
                }
        }
        int zipa = 0;                                                     // per class: the zone-like part of the InterPodAffinity raw score
        if constexpr (kIpa) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (lane < Cn && kI[i] == 2) {
                    const int zd = s_zdom[zI[i] * Cn + dd];
                    zipa += zd >= 0 ? wI[i] * (int)g_zcnt[rI[i] * 16 + zd] : 0;
                }
        }
        unsigned hcz[3] = {0u, 0u, 0u};                                   // per class: matching pods in the class's zone, per hard constraint
        int hzd[3] = {0, 0, 0};
        if constexpr (kIpa) {
#pragma unroll
            for (int i = 0; i < 3; ++i)
                if (i < hard_n) {
                    hzd[i] = lane < Cn ? (int)s_zdom[hZ[i] * Cn + dd] : -1;
                    hcz[i] = hzd[i] >= 0 ? g_zcnt[hR[i] * 16 + hzd[i]] : 0u;
                }
        }
        int zipa2 = 0;                                                    // (CN2: the same for class 64 + lane)
        unsigned hcz2[3] = {0u, 0u, 0u};
        int hzd2[3] = {0, 0, 0};
        if constexpr (kIpa && CN2) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (v2 && kI[i] == 2) {
                    const int zd = s_zdom[zI[i] * Cn + dd2];
                    zipa2 += zd >= 0 ? wI[i] * (int)g_zcnt[rI[i] * 16 + zd] : 0;
                }
#pragma unroll
            for (int i = 0; i < 3; ++i)
                if (i < hard_n) {
                    hzd2[i] = v2 ? (int)s_zdom[hZ[i] * Cn + dd2] : -1;
                    hcz2[i] = hzd2[i] >= 0 ? g_zcnt[hR[i] * 16 + hzd2[i]] : 0u;
                }
        }
        const unsigned char* hb1 = has_hrow ? g_hrow + (size_t)hrow_i * ni : (const unsigned char*)g_tile;   // (no hostname term: value unused)
        const int hmx_v = (simple && has_hrow) ? (int)g_hmax[hrow_i] : 0;   // largest counter of the row (uniform address)
        unsigned byte1[SB], h1[SB];
        unsigned long long bad1[RS ? SB : 1];
        auto load1 = [&](int u0) {
#pragma unroll
            for (int j = 0; j < SB; ++j) {
                const int u = min(u0 + j, uhi - 1);                       // a slot beyond the last unit repeats it (changes nothing)
                byte1[j] = g_tile[(unsigned)u * (Krow * 4u) + toff];
                h1[j] = hb1[(unsigned)(u * 64 + lane)];
                if constexpr (RS) bad1[j] = filt ? bad_of(u) : 0ull;
            }
        };
        if (simple && have) load1(ulo);
        TPROF(12);                                                         // spread: descriptor, first loads issued
        bool ign = false;                                                 // IgnoredNodes (:80-85): a constraint key is missing
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (e < soft_n && kind[e] == 2) ign = ign || s_zdom[zsl[e] * Cn + dd] < 0;
        bool ign2 = false;
        if constexpr (CN2) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (e < soft_n && kind[e] == 2) ign2 = ign2 || s_zdom[zsl[e] * Cn + dd2] < 0;
        }
        bool excl = false;                                                // this class fails a hard constraint of the pod
        bool excl2 = false;
        if constexpr (kIpa) {
#pragma unroll
            for (int i = 0; i < 3; ++i)
                if (i < hard_n) {
                    // registered zones (:236-251): those of the scenario's labelled nodes (the host admits eligibility sets that leave out
                    // unlabelled nodes only); criticalPaths minimum over them (:272-278), math.MaxInt32 when there is none
                    const bool reg = lane < Cn && cnt_d > 0 && hzd[i] >= 0;
                    int mn = wave_min_i32(reg ? (int)hcz[i] : 0x7fffffff);
                    if constexpr (CN2) {
                        const bool reg2 = v2 && cnt_d2 > 0 && hzd2[i] >= 0;
                        mn = min(mn, wave_min_i32(reg2 ? (int)hcz2[i] : 0x7fffffff));
                        const long long skew2 = (long long)hcz2[i] + ((hS[i] >> 14) & 1) - (long long)mn;
                        excl2 = excl2 || hzd2[i] < 0 || skew2 > (long long)(hS[i] & 0x3FFF);
                    }
                    const long long skew = (long long)hcz[i] + ((hS[i] >> 14) & 1) - (long long)mn;
                    excl = excl || hzd[i] < 0 || skew > (long long)(hS[i] & 0x3FFF);
                }
        }
        const bool scored = lane < Cn && cntd > 0 && !ign && !excl;
        const bool scored2 = CN2 && v2 && cntd2 > 0 && !ign2 && !excl2;
        const int F = __builtin_amdgcn_readfirstlane(wave_sum_i32_t((scored ? cntd : 0) + (scored2 ? cntd2 : 0)));   // len(filteredNodes) - len(IgnoredNodes)
        int sz[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {                                     // TopologyNormalizingWeight (:98-106, :279-281): its size argument
            sz[e] = 0;
            if (e < soft_n && !dup[e]) {
                if (kind[e] == 1) sz[e] = F;
                else sz[e] = __popc((unsigned)__builtin_amdgcn_readfirstlane((int)wave_or_u32_t((scored ? 1u << (s_zdom[zsl[e] * Cn + dd] & 31) : 0u) |
                                                                                                   (scored2 ? 1u << (s_zdom[zsl[e] * Cn + dd2] & 31) : 0u))));
            }
        }
        TPROF(13);                                                         // spread: feasible-node counters arrived, sizes
        // (the doubles are built where they are used: the two branches below keep different ones alive)
        auto w_of = [&](int e) -> double { return e < soft_n ? gp(cold->spread_log)[sz[e]] : 0.0; };
        auto cst_of = [&](int e) -> double { return (double)(skew[e] - 1); };
        // per class, held by lane = class: the zone term of constraint e (scoreForCount, :287-289, of the class's zone)
        auto az_of = [&](int e) -> double { return (lane < Cn && e < soft_n && kind[e] == 2) ? (double)czv[e] * w_of(e) + cst_of(e) : 0.0; };
        auto az2_of = [&](int e) -> double { return (v2 && e < soft_n && kind[e] == 2) ? (double)czv2[e] * w_of(e) + cst_of(e) : 0.0; };   // (CN2: class 64 + lane)
        // per class as well: "ignored" and the class term of the signature's row
        // (bit 30: excluded -- its nodes count as infeasible; the Simon normalisation then runs over the classes that are left, simon.go:76-101)
        int ctermv = (int)s_sn[k * Cn + dd];
        if constexpr (RS) {                                               // under filter rows: the Simon term over the classes that keep an admitted node (simon.go:76-101)
            if (filt) ctermv = class_term(lane < Cn && cntd > 0, simon_raw[tc * Cn + dd], tc, dd);
        }
        int ctermv2 = CN2 ? (int)s_sn[k * Cn + dd2] : 0;
        if constexpr (kIpa && CN2) {
            if (hard_n > 0 && __ballot((cntd > 0 && excl) || (v2 && cntd2 > 0 && excl2)) != 0ull)
                class_term2(cntd > 0 && !excl, v2 && cntd2 > 0 && !excl2, simon_raw[tc * Cn + dd], simon_raw[tc * Cn + dd2], tc, dd, dd2, ctermv, ctermv2);
        } else if constexpr (kIpa) {
            if (hard_n > 0 && __ballot(lane < Cn && cntd > 0 && excl) != 0ull)
                ctermv = class_term(lane < Cn && cntd > 0 && !excl, simon_raw[tc * Cn + dd], tc, dd);
        }
        const int clsw = (int)((unsigned)ctermv | (ign ? 0x80000000u : 0u) | (excl ? 0x40000000u : 0u));
        const int clsw2 = CN2 ? (int)((unsigned)ctermv2 | (ign2 ? 0x80000000u : 0u) | (excl2 ? 0x40000000u : 0u)) : 0;
        auto cls_word = [&](int c) -> int {                                // the word of class c (uniform), from the lane that holds it
            if constexpr (CN2) {
                const int a = __builtin_amdgcn_readlane(clsw, c & 63), b = __builtin_amdgcn_readlane(clsw2, c & 63);
                return c < 64 ? a : b;
            } else return __builtin_amdgcn_readlane(clsw, c);
        };
        auto zipa_of = [&](int c) -> int {                                 // the zone-like part of the InterPodAffinity raw score of class c (uniform)
            if constexpr (CN2) {
                const int a = __builtin_amdgcn_readlane(zipa, c & 63), b = __builtin_amdgcn_readlane(zipa2, c & 63);
                return c < 64 ? a : b;
            } else return __builtin_amdgcn_readlane(zipa, c);
        };
        auto lane_f64 = [&](double v, int l) -> double {
            const unsigned long long b = (unsigned long long)__double_as_longlong(v);
            const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, l), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), l);
            return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
        };
        int pmin = 0x7fffffff, pmax = 0;
        unsigned bkey = 0;
        int bpos = 0;
        // Both passes (minimum / maximum of the raw scores, then totals) walk the row a batch of units at a time: the loads of a batch
        // are issued together, so a pass costs ceil(units / batch) memory round trips instead of one per unit (a scenario's workspace
        // lives in HBM / MALL: ~2 000 cycles each).  A slot beyond the last unit repeats it: duplicates change neither the extremes
        // nor the first maximum.
        // SIMPLE and small enough: raw score and total depend on (class, count) only, so both are tabulated -- one entry per lane, all
        // the float arithmetic of the pod in a few instructions -- and the walks look them up: integer work and LDS reads per position.
        const int hmx = __builtin_amdgcn_readfirstlane(hmx_v);
        const int lg = 32 - __clz(hmx);                                  // counts 0 .. (1 << lg) - 1
        const int E = Cn << lg;
        if (simple && E <= TABMAX) {
            const double ws = nh == 1 ? w_of(0) : 0.0, cs = nh == 1 ? cst_of(0) : 0.0;
            const bool ipa_pod = kIpa && ipa_n > 0;
            // the per-class part: the zone-like constraints in list order (float addition is not associative; x + 0.0 == x)
            double pcl = 0.0;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (e < soft_n && kind[e] == 2) pcl = pcl + az_of(e);
            const unsigned long long pclb = (unsigned long long)__double_as_longlong(pcl);
            unsigned long long pclb2 = 0ull;
            if constexpr (CN2) {
                double pcl2 = 0.0;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (e < soft_n && kind[e] == 2) pcl2 = pcl2 + az2_of(e);
                pclb2 = (unsigned long long)__double_as_longlong(pcl2);
            }
            const int hmask = (1 << lg) - 1;
            for (int i0 = 0; i0 < E; i0 += 64) {                          // raw(class, count) = int64((count * w + c) + zone terms of the class)
                const int i = i0 + lane, ci = min(i >> lg, Cn - 1), c4 = (CN2 ? ci & 63 : ci) * 4;
                unsigned lo = (unsigned)__builtin_amdgcn_ds_bpermute(c4, (int)(unsigned)pclb), hi = (unsigned)__builtin_amdgcn_ds_bpermute(c4, (int)(unsigned)(pclb >> 32));
                if constexpr (CN2) {
                    const unsigned lo2 = (unsigned)__builtin_amdgcn_ds_bpermute(c4, (int)(unsigned)pclb2), hi2 = (unsigned)__builtin_amdgcn_ds_bpermute(c4, (int)(unsigned)(pclb2 >> 32));
                    lo = ci < 64 ? lo : lo2; hi = ci < 64 ? hi : hi2;
                }
                const double pc = __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
                const int raw = (int)(((double)(i & hmask) * ws + cs) + pc);
                if (i < E) s_tab[i] = raw;
                if constexpr (kIpa) {                                     // InterPodAffinity raw score of (class, count): integer (scoring.go:211-236)
                    int zc = __builtin_amdgcn_ds_bpermute(c4, zipa);          // (every lane: the source lane of a permute must be active)
                    if constexpr (CN2) { const int zc2 = __builtin_amdgcn_ds_bpermute(c4, zipa2); zc = ci < 64 ? zc : zc2; }
                    if (ipa_pod && i < E) s_tabi[i] = Wh * (i & hmask) + zc;
                }
            }
            TPROF_WAIT_LDS; TPROF(14);                                     // spread: zone counters, Log table, raw score table
            int imin = 0, imax = 0;                                       // InterPodAffinity: extremes over ALL feasible nodes, from 0 (:247-256)
            auto pass1 = [&](auto ipa_tag) {                              // pass 1; leaves (count, table byte) of every position in LDS
                constexpr bool IPA = decltype(ipa_tag)::value;
                if (!have) return;
                for (int u0 = ulo;;) {
#pragma unroll
                    for (int j = 0; j < SB; ++j) {
                        const int u = min(u0 + j, uhi - 1);
                        const int c = winner_info(u) >> 16;
                        const int idx = (c << lg) + ((int)h1[j] & hmask);     // (padding positions and the no-hostname-term case read arbitrary bytes)
                        const int raw = s_tab[idx];
                        const int cwv = cls_word(c);
                        unsigned beff = byte1[j];
                        if constexpr (RS) beff = ((bad1[j] >> lane) & 1ull) ? 0u : beff;   // a position the pod's filter rows exclude
                        if constexpr (kIpa) beff = (cwv & 0x40000000) ? 0u : beff;   // a class a hard constraint excludes: no feasible node
                        const bool ok = (beff != 0u) & (cwv >= 0);            // (no short circuit: no branch)
                        pmin = min(pmin, ok ? raw : 0x7fffffff);
                        pmax = max(pmax, ok ? raw : 0);
                        if constexpr (IPA) {
                            const int ir = s_tabi[idx];
                            imin = min(imin, beff != 0u ? ir : 0);
                            imax = max(imax, beff != 0u ? ir : 0);
                        }
                        (s_stash + u * 64)[lane] = (unsigned short)(h1[j] | (beff << 8));   // (uniform base + lane: one address add)
                    }
                    u0 += SB;
                    if (u0 >= uhi) break;
                    load1(u0);
                }
            };
            if constexpr (kIpa) { if (ipa_pod) pass1(std::true_type{}); else pass1(std::false_type{}); }
            else pass1(std::false_type{});
            TPROF(15);                                                     // spread: pass 1
            pmin = wave_min_i32(pmin);
            pmax = wave_max_i32(pmax);
            if constexpr (kIpa) {
                if (ipa_pod) { imin = wave_min_i32(imin); imax = wave_max_i32(imax); }
            }
            team_extremes(pmin, pmax, imin, imax);
            const int idiff = imax - imin;
            const double rinv = pmax > 0 ? 1.0 / (double)pmax : 0.0, hrinv = 0.5 * rinv;
            const int pmm = pmax + pmin;
            for (int i0 = 0; i0 < E; i0 += 64) {                          // class term + 2 x NormalizeScore (:217-256), in place
                const int i = i0 + lane;
                const int raw = s_tab[min(i, E - 1)];
                const int ci = min(i >> lg, Cn - 1);
                int cw = __builtin_amdgcn_ds_bpermute((CN2 ? ci & 63 : ci) * 4, clsw);
                if constexpr (CN2) { const int cw2 = __builtin_amdgcn_ds_bpermute((ci & 63) * 4, clsw2); cw = ci < 64 ? cw : cw2; }
                // (an entry no feasible scored node has may lie outside [min, max]: its value is never looked up)
                int v = pmax == 0 ? 100 : (int)__builtin_fma((double)(100 * (pmm - raw)), rinv, hrinv);
                v = cw >= 0 ? v : 0;                                      // ignored nodes score 0
                int iv = 0;                                               // InterPodAffinity NormalizeScore (:258-271): float64, weight 1
                if constexpr (kIpa) {
                    if (ipa_pod && idiff > 0) iv = (int)(100.0 * ((double)(s_tabi[min(i, E - 1)] - imin) / (double)idiff));
                }
                if (i < E) s_tot[i] = (cw & 0x3fffffff) + 2 * v + iv;       // (team mode: every wave writes the whole table -- identical values)
            }
            TPROF_WAIT_LDS; TPROF(16);                                     // spread: extremes, table of totals
            for (int u0 = ulo; u0 < uhi; u0 += SC) {                     // pass 2: totals, first maximum in canonical order
                int canon[SC], cbase[SC];
                unsigned stj[SC];
#pragma unroll
                for (int j = 0; j < SC; ++j) {
                    const int u = min(u0 + j, uhi - 1);
                    const int info = winner_info(u);                      // (offset into cls_list + 8192) | class << 16 of the unit
                    // uniform base + lane (scalar address arithmetic); the padding positions of the last class read past the lists, into
                    // the 64 entries of slack behind them (their table byte is 0)
                    canon[j] = RANKED ? (int)(g_canon + u * 64)[lane] : (cls_list + (rk_off + (unsigned)((info & 0xFFFF) - 8192 + u * 64)))[lane];
                    stj[j] = (s_stash + u * 64)[lane];
                    cbase[j] = (info >> 16) << lg;
                }
                int t2[SC];
#pragma unroll
                for (int j = 0; j < SC; ++j) t2[j] = s_tot[cbase[j] + ((int)stj[j] & hmask)];
#pragma unroll
                for (int j = 0; j < SC; ++j) {
                    const int u = min(u0 + j, uhi - 1);
                    const unsigned byte = stj[j] >> 8;                   // total + 1 = (byte - 1) + (class term + 2 x score) + 1
                    const unsigned key = byte != 0u ? ((byte + (unsigned)t2[j]) << 13) | (8191u - (unsigned)canon[j]) : 0u;
                    if (key > bkey) { bkey = key; bpos = u * 64 + lane; }
                }
            }
        } else {
            // several per-node terms, or class terms before the per-node one: every constraint in list order, counters re-read in pass 2
            const unsigned char* hb[4];
            double w[4], cst[4], azv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {                                 // a slot without a hostname-like constraint reads the table (value unused)
                hb[e] = (e < soft_n && kind[e] == 1) ? g_hrow + (size_t)rowi[e] * ni : (const unsigned char*)g_tile;
                w[e] = w_of(e); cst[e] = cst_of(e); azv[e] = az_of(e);
            }
            double azv2[4] = {0.0, 0.0, 0.0, 0.0};
            if constexpr (CN2) {
#pragma unroll
                for (int e = 0; e < 4; ++e) azv2[e] = az2_of(e);
            }
            // (InterPodAffinity raw score per position here: the count of its hostname-like term + the class's zone part)
            const unsigned char* hbI = ipa_h ? g_hrow + (size_t)rI[0] * ni : (const unsigned char*)g_tile;
            const bool ipa_pod = kIpa && ipa_n > 0;
            int imin = 0, imax = 0;
            auto raw_of = [&](const unsigned (&h)[4], int c) -> int {
                double scv = 0.0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    double az = lane_f64(azv[e], CN2 ? c & 63 : c);
                    if constexpr (CN2) { const double az2 = lane_f64(azv2[e], c & 63); az = c < 64 ? az : az2; }
                    const double ah = (double)h[e] * w[e] + cst[e];
                    const double A = e < soft_n ? (kind[e] == 1 ? ah : az) : 0.0;
                    scv = scv + A;
                }
                return (int)scv;
            };
            for (int u0 = ulo; u0 < uhi; u0 += SG) {                      // pass 1
                unsigned byte[SG], h[SG][4], hI[SG];
#pragma unroll
                for (int j = 0; j < SG; ++j) {
                    const int u = min(u0 + j, uhi - 1);
                    byte[j] = g_tile[(unsigned)u * (Krow * 4u) + toff];
                    if constexpr (RS) { if (filt) byte[j] = ((bad_of(u) >> lane) & 1ull) ? 0u : byte[j]; }
#pragma unroll
                    for (int e = 0; e < 4; ++e) h[j][e] = hb[e][(unsigned)(u * 64 + lane)];
                    hI[j] = kIpa ? hbI[(unsigned)(u * 64 + lane)] : 0u;
                }
#pragma unroll
                for (int j = 0; j < SG; ++j) {
                    const int c = winner_info(min(u0 + j, uhi - 1)) >> 16;
                    const int raw = raw_of(h[j], c);
                    const int cwv = cls_word(c);
                    unsigned beff = byte[j];
                    if constexpr (kIpa) beff = (cwv & 0x40000000) ? 0u : beff;
                    const bool ok = (beff != 0u) & (cwv >= 0);
                    pmin = min(pmin, ok ? raw : 0x7fffffff);
                    pmax = max(pmax, ok ? raw : 0);
                    if constexpr (kIpa) {
                        const int ir = (ipa_h ? Wh * (int)hI[j] : 0) + zipa_of(c);
                        imin = min(imin, (ipa_pod && beff != 0u) ? ir : 0);
                        imax = max(imax, (ipa_pod && beff != 0u) ? ir : 0);
                    }
                }
            }
            pmin = wave_min_i32(pmin);
            pmax = wave_max_i32(pmax);
            if constexpr (kIpa) { imin = wave_min_i32(imin); imax = wave_max_i32(imax); }
            team_extremes(pmin, pmax, imin, imax);
            const int idiff = imax - imin;
            const double rinv = pmax > 0 ? 1.0 / (double)pmax : 0.0, hrinv = 0.5 * rinv;
            for (int u0 = ulo; u0 < uhi; u0 += SG) {                      // pass 2
                unsigned byte[SG], h[SG][4], hI[SG];
                int canon[SG], infoj[SG];
#pragma unroll
                for (int j = 0; j < SG; ++j) {
                    const int u = min(u0 + j, uhi - 1);
                    byte[j] = g_tile[(unsigned)u * (Krow * 4u) + toff];
                    if constexpr (RS) { if (filt) byte[j] = ((bad_of(u) >> lane) & 1ull) ? 0u : byte[j]; }
#pragma unroll
                    for (int e = 0; e < 4; ++e) h[j][e] = hb[e][(unsigned)(u * 64 + lane)];
                    hI[j] = kIpa ? hbI[(unsigned)(u * 64 + lane)] : 0u;
                    const int info = infoj[j] = winner_info(u);
                    canon[j] = RANKED ? (int)(g_canon + u * 64)[lane] : (cls_list + (rk_off + (unsigned)((info & 0xFFFF) - 8192 + u * 64)))[lane];
                }
#pragma unroll
                for (int j = 0; j < SG; ++j) {
                    const int u = min(u0 + j, uhi - 1);
                    const int c = infoj[j] >> 16;
                    const int raw = raw_of(h[j], c);
                    const int cw = cls_word(c);
                    int v = 0;
                    if (cw >= 0) v = pmax == 0 ? 100 : (int)__builtin_fma((double)(100 * (pmax + pmin - raw)), rinv, hrinv);
                    int iv = 0;                                           // InterPodAffinity NormalizeScore (:258-271)
                    if constexpr (kIpa) {
                        if (ipa_pod && idiff > 0) iv = (int)(100.0 * ((double)((ipa_h ? Wh * (int)hI[j] : 0) + zipa_of(c) - imin) / (double)idiff));
                    }
                    unsigned beff = byte[j];
                    if constexpr (kIpa) beff = (cw & 0x40000000) ? 0u : beff;
                    const int total = (int)beff - 1 + (cw & 0x3fffffff) + 2 * v + iv;
                    const unsigned key = beff != 0u ? ((unsigned)(total + 1) << 13) | (8191u - (unsigned)canon[j]) : 0u;
                    if (key > bkey) { bkey = key; bpos = u * 64 + lane; }
                }
            }
        }
        TPROF(17);                                                         // spread: pass 2
        unsigned best = wave_max_u32(bkey);
        int pstar;
        if constexpr (NW == 1) {
            if (best == 0u) return -1;
            const int wl = __builtin_ctzll(__ballot(bkey == best));
            pstar = __builtin_amdgcn_readlane(bpos, wl);
        } else {                                                          // the team's best: keys carry the canonical index, so the maximum is one node
            const int wl = best != 0u ? __builtin_ctzll(__ballot(bkey == best)) : 0;
            const int pw = __builtin_amdgcn_readlane(bpos, wl);
            if (lane == 0) { s_xch[4 * NW + 2 * wv] = (int)best; s_xch[4 * NW + 2 * wv + 1] = pw; }
            __syncthreads();
            best = 0u; pstar = -1;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const unsigned kw = (unsigned)s_xch[4 * NW + 2 * w];
                const int pq = s_xch[4 * NW + 2 * w + 1];
                if (kw > best) { best = kw; pstar = pq; }
            }
            pstar = __builtin_amdgcn_readfirstlane(pstar);
            if (best == 0u) return -1;
        }
        const int info = winner_info(pstar >> 6);
        dstar = info >> 16;
        res = (info & 0xFFFF) - 8192 + pstar;
        return pstar;
    };
    // AddPod's side of the counters (podtopologyspread/scoring.go:129-160 counts the pods ON the nodes; here they are counted as
    // they land): lane soft_n + e holds counted entry e = term slot | multiplicity << 16 (terms are distinct per class: plain
    // read-modify-writes of distinct bytes / words).  A term counts only on the nodes of its node set (scoring.go:137-141).
    // The loads go out with the assume's (right after the select), the stores at the end of the cycle: nothing else touches these
    // bytes / words in between, and the wave does not wait a memory round trip of its own for them.
    struct SpreadLoads { unsigned old, mx; bool on; };
    auto spread_count_load = [&](int pstar, int dstar, int res, int spv, int spt, int soft_n, int match_n) -> SpreadLoads {
        SpreadLoads L{0u, 0u, false};
        const int e = lane - soft_n;
        if (e < 0 || e >= match_n) return L;
        const int ti = spt;
        const int kindt = ti & 3, rowt = (ti >> 2) & 0x3FFF, zst = (ti >> 16) & 7, set = ((ti >> 19) & 0xFFF) - 1;
#ifdef SIMON_SPREAD_DEBUG
        printf("SPC s=%d lane=%d e=%d spv=%x ti=%x kind=%d row=%d set=%d pstar=%d dstar=%d res=%d\n", s, lane, e, spv, ti, kindt, rowt, set, pstar, dstar, res);
#endif
        L.on = true;
        if (set >= 0) {
            const int jn = cls_list[rk_off + (unsigned)res];              // the node itself (canonical pool index)
            L.on = (gp(cold->node_sets)[(size_t)set * cold->set_words + (jn >> 6)] >> (jn & 63)) & 1ull;
        }
        if (kindt == 1) {
            L.old = g_hrow[(size_t)rowt * ni + (unsigned)pstar];
            L.mx = g_hmax[rowt];
        } else if (kindt == 2) {
            const int zd = s_zdom[zst * Cn + dstar];
            L.on = L.on && zd >= 0;
            L.old = g_zcnt[rowt * 16 + (zd >= 0 ? zd : 0)];
        } else {
            L.on = false;
        }
        return L;
    };
    auto spread_count_store = [&](const SpreadLoads& L, int pstar, int dstar, int spv, int spt, int soft_n, int match_n) {
        const int e = lane - soft_n;
        if (e < 0 || e >= match_n || !L.on) return;
        const unsigned mult = ((unsigned)spv >> 16) & 0xFFu;
        const int kindt = spt & 3, rowt = (spt >> 2) & 0x3FFF, zst = (spt >> 16) & 7;
        const unsigned nv = L.old + mult;
        if (kindt == 1) {
            g_hrow[(size_t)rowt * ni + (unsigned)pstar] = (unsigned char)nv;
            if (nv > L.mx) g_hmax[rowt] = (unsigned char)nv;              // (the row's largest counter bounds spread_select's score table)
        } else {
            g_zcnt[rowt * 16 + s_zdom[zst * Cn + dstar]] = nv;
        }
    };

    int unsched = 0;
    int32_t* __restrict__ place = place_step ? place_step + (size_t)s * P : nullptr;

    // pod stream: 64 rows per vector load (lane l holds step i0 + l), one chunk ahead.  x packs what the common path needs into ONE
    // readlane: signature | pod class << 10, sign bit = "special" (gated out of this scenario, preset or pinned): y = preset, z = gate.
    auto load_chunk = [&](int i0) -> int4 {
        const int idx = i0 + lane;
        if (idx >= P) return make_int4((int)0x80000000, -1, 0x7fffffff, 0);
        const PodRowC r = pods[order[idx]];
        const bool special = r.gate >= n || r.preset != -1;
        return make_int4(r.sigcls | (special ? (int)0x80000000 : 0), r.preset, r.gate, r.rest);
    };
    int4 nxt = load_chunk(0);
    auto load_spw = [&](int i0) -> int {                                   // REST && SPREAD: the SPREAD descriptor, by pod id (TableCold::sp_word)
        if constexpr (RS) { const int idx = i0 + lane; return idx < P ? gp(cold->sp_word)[order[idx]] : 0; }
        else return 0;
    };
    int nxt_sw = load_spw(0);

    for (int i0 = 0; i0 < P; i0 += 64) {
        const int4 cur = nxt;
        nxt = load_chunk(i0 + 64);
        const int cur_sw = nxt_sw;
        if constexpr (RS) nxt_sw = load_spw(i0 + 64);
        const int steps = P - i0 < 64 ? P - i0 : 64;
        int plreg = -2;
        for (int il = 0; il < steps; ++il) {
        TPROF(0);                                                      // loop control, placement flush, pod chunk
        const int pk = __builtin_amdgcn_readlane(cur.x, il);
        const int r_sig = pk & 0x3FF, r_cls = (pk >> 10) & 0x1FFFFF;
        const int rw = (REST || SPREAD) ? __builtin_amdgcn_readlane(cur.w, il) : 0;    // REST / SPREAD descriptor (0: the score table alone decides the pod)
        const int sw = RS ? __builtin_amdgcn_readlane(cur_sw, il) : rw;    // the SPREAD descriptor (REST && SPREAD: a word of its own)
        const int sp_soft = SPREAD ? (sw & 7) : 0, sp_match = SPREAD ? ((sw >> 3) & 127) : 0, sp_ipa = SPREAD ? ((sw >> 10) & 7) : 0, sp_hard = SPREAD ? ((sw >> 13) & 3) : 0;
        int spv = 0;                                                       // SPREAD: lane e holds entry e of the pod's constraint / counted-term list
        int spt = 0;                                                       // ... and its term's row (TableCold::sp_ent holds both)
        if (SPREAD && __builtin_expect(sw != 0, 0) && lane < sp_soft + sp_match + sp_ipa + sp_hard) {
            const int2 spe = ldg_i2(gp(cold->sp_ent) + (((unsigned)sw >> 15) + lane));
            spv = spe.x; spt = spe.y;
        }
        const int r_gs = REST ? (rw & 63) - 1 : -1, r_xs = REST ? ((rw >> 6) & 63) - 1 : -1, r_nrows = REST ? (rw >> 12) & 63 : 0;
        int rowv = 0;                                                      // lane e: filter row | row to set << 16 of entry e
        if (REST && lane < r_nrows) rowv = gp(cold->xrows)[((unsigned)rw >> 18) + lane];
        unsigned rtv = 0;                                                  // lane e: placed pods counted on entry e's row (required affinity)
        if (REST && AFF && lane < r_nrows && ((unsigned)rowv >> 31)) rtv = g_rowtot[(unsigned)rowv & 0xFFFFu];

        // res: what the placement row records for this step: >= 0 an index into cls_list (turned into the canonical node index
        // 64 steps at a time, off the critical path), -1 unschedulable, -2 not part of the scenario
        int res, pstar = -1, dstar = 0;
        unsigned top = 0, m16q[NBQ];
        bool scanned = false, bound = false;                           // bound: a preset pod (no Reserve: the scheduler never saw it)
        // a pod with soft spread constraints / preferred pod (anti-)affinity / hard zone constraints: every node's score moves -- the one
        // select the whole team takes part in (a single call site below: helpers skip everything else of the cycle)
        const bool spread_pod = SPREAD && pk >= 0 && (sp_soft | sp_ipa | sp_hard) != 0;
        if (!lead) {
            res = -2;
        } else if (__builtin_expect(pk < 0, 0)) {
            const int r_preset = __builtin_amdgcn_readlane(cur.y, il), r_gate = __builtin_amdgcn_readlane(cur.z, il);
            GPtr<const TableCold> cc = cold;
            asm volatile("" : "+s"(cc));                               // rare paths: fetch their pointers here, not in loop-long SGPRs
            if (r_gate >= n) {
                res = -2;                                              // pod not part of this scenario
            } else if (r_preset >= 0) {                                // addPodToCache path (V/eventhandlers.go:223-236)
                bound = true;
                dstar = __builtin_amdgcn_readfirstlane(gp(cc->ncls)[r_preset]);
                const int rk = __builtin_amdgcn_readfirstlane(ranked ? gp(cc->rk_pos)[(size_t)s * (size_t)cc->N + r_preset] : gp(cc->rank)[r_preset]);
                pstar = __builtin_amdgcn_readfirstlane(s_seg[dstar]) + rk;
                res = __builtin_amdgcn_readfirstlane(gp(cc->cls_off)[dstar]) + rk;
            } else {                                                   // pinned pod (simon_pods_soa.pin_node, stored as -2 - node):
                const int pin = -2 - r_preset;                         // its node affinity admits ONE node; the table byte of
                res = -1;                                              // (signature, node) holds static filters + fit
                if (HAS_PIN && pin < n) {
                    const int dp = __builtin_amdgcn_readfirstlane(gp(cc->ncls)[pin]);
                    const int rk = __builtin_amdgcn_readfirstlane(ranked ? gp(cc->rk_pos)[(size_t)s * (size_t)cc->N + pin] : gp(cc->rank)[pin]);
                    const int pp = __builtin_amdgcn_readfirstlane(s_seg[dp]) + rk;
                    const unsigned char byte = g_tile[tile_blk((unsigned)(pp >> 4)) + (unsigned)r_sig * KS + (unsigned)(pp & 15)];
                    bool clear = true;
                    if (REST && rw != 0) clear = !((__builtin_amdgcn_readfirstlane(excluded(pp >> 4, r_nrows, rowv, r_gs, r_xs, aff_escape(r_nrows, rowv, rtv))) >> (pp & 15)) & 1u);
                    if (byte != 0 && clear) { res = __builtin_amdgcn_readfirstlane(gp(cc->cls_off)[dp]) + rk; pstar = pp; dstar = dp; }
                }
                if (res < 0) ++unsched;
            }
        } else if (REST && __builtin_expect(rw != 0, 0) && !(RS && spread_pod)) {   // (cold for the register allocator: spills belong here)
            pstar = rest_select(r_sig, r_cls, r_nrows, rowv, r_gs, r_xs, aff_escape(r_nrows, rowv, rtv), dstar, res);
            TPROF(10);                                                 // REST pods: the whole select
            if (pstar < 0) { ++unsched; res = -1; }
        } else if (spread_pod) {
            const int k = r_sig;
            const unsigned dq = (unsigned)__builtin_amdgcn_readlane((int)my_dirty, k & 63);
            if ((dq >> (k >> 6)) & 1u) {                               // the class terms of row k (s_sn) must be current
                renormalise(k, r_cls);
                if (lane == (k & 63)) my_dirty &= ~(1u << (k >> 6));
            }
            if constexpr (NW == 1) {                                   // (one wave: the select sits where it always sat -- the team's call site is below)
                if constexpr (RS) pstar = spread_select(k, r_cls, sp_soft, sp_match, sp_ipa, sp_hard, spv, spt, dstar, res, r_nrows, rowv, r_gs, r_xs, aff_escape(r_nrows, rowv, rtv));
                else pstar = spread_select(k, r_cls, sp_soft, sp_match, sp_ipa, sp_hard, spv, spt, dstar, res);
                TPROF(18);                                             // spread: winner
                if (pstar < 0) { ++unsched; res = -1; }
            }
        } else {
            const int k = r_sig;
            const unsigned dq = (unsigned)__builtin_amdgcn_readlane((int)my_dirty, k & 63);
            if (__builtin_expect((dq >> (k >> 6)) & 1u, 0)) {          // a node class lost its last feasible node for k (or first use)
                renormalise(k, r_cls);
                if (lane == (k & 63)) my_dirty &= ~(1u << (k >> 6));
            }
            // -------- summary scan: one u16 per 16 (COARSE: 64) positions ------------------------------
            const unsigned short* srow = s_sum + k * nbp;
            unsigned key = 0;
#pragma unroll
            for (int q = 0; q < NBQ; ++q) {
                const int b = q * 64 + lane;
                // a lane beyond the last entry re-reads entry 0: the duplicate ties on the total and loses on the position
                m16q[q] = srow[b < nun ? b : 0];
            }
#pragma unroll
            for (int q = 0; q < NBQ; ++q) {
                // (total + 1) << KB | (last - b) << UB | UNIT - 1 - pos  ==  (total + 1) << KB | PMASK - position; an entry of 0 (no
                // feasible node behind it) stays below 1 << KB
                const unsigned m16 = m16q[q];
                key = max(key, (((m16 << (KB - UB)) & ~PMASK) | (unsigned)cb[q]) | (m16 & UMASK));
            }
            TPROF_WAIT_LDS; TPROF(1);                                  // pod row, dirty check, summary row arrived
            key = wave_max_u32(key);
            TPROF(2);                                                  // key build + wave max
            if (__builtin_expect(key <= PMASK, 0)) {                   // FitError: pod deleted, state unchanged
                ++unsched;
                res = -1;
            } else {
                pstar = (int)(PMASK - (key & PMASK));                  // first maximum in POSITION order (speculative: tie check below)
                top = key >> KB;
                const int info = winner_info(pstar >> UB);
                dstar = info >> 16;
                res = (info & 0xFFFF) - 8192 + pstar;                  // index into cls_list
                scanned = true;
            }
        }
        if constexpr (NW > 1) {
            if (spread_pod) {                                          // every wave of the team
                if constexpr (RS) pstar = spread_select(r_sig, r_cls, sp_soft, sp_match, sp_ipa, sp_hard, spv, spt, dstar, res, r_nrows, rowv, r_gs, r_xs, aff_escape(r_nrows, rowv, rtv));
                else pstar = spread_select(r_sig, r_cls, sp_soft, sp_match, sp_ipa, sp_hard, spv, spt, dstar, res);
                TPROF(18);                                             // spread: winner
                if (pstar < 0) { ++unsched; res = -1; }
            }
        }
        // -------- assume: NodeInfo.AddPod (V/framework/types.go:482-508) + table column + summary ----
        if (lead && pstar >= 0) {
            TPROF(3);                                                  // winner info
            // Position order is canonical order inside a class only: do entries of ANOTHER class reach the same total?  Then the
            // first maximum in CANONICAL order decides (static per-class node lists, L2-hot).
            auto tie_with_other_class = [&]() -> bool {
                bool other = false;
#pragma unroll
                for (int q = 0; q < NBQ; ++q) other = other || ((m16q[q] >> UB) == top && bcls[q] != dstar);   // duplicates carry entry 0's class
                return __ballot(other) != 0;
            };
            auto canonical_first = [&]() -> int {
#ifdef SIMON_TABLE_PROFILE
                tp_acc[7] += 1;                                        // how often the canonical tie-break runs
#endif
                unsigned key2 = 0;
                int canon[NBQ];
#pragma unroll
                for (int q = 0; q < NBQ; ++q) {
                    const int pq = (q * 64 + lane) * UNIT + (UNIT - 1) - (int)(m16q[q] & UMASK);
                    const bool tied = (m16q[q] >> UB) == top && q * 64 + lane < nun;
                    if constexpr (LDSX) canon[q] = tied ? (int)sx_canon[pq] : (int)PMASK;
                    else {
                    canon[q] = tied ? cls_list[rk_off + (unsigned)((binfo[q] & 0xFFFF) - 8192 + pq)] : (int)PMASK;
                    if (ranked && tied) canon[q] = gp(cold->rk_rank)[(size_t)s * (size_t)cold->N + canon[q]];
                    }
                }
#pragma unroll
                for (int q = 0; q < NBQ; ++q) {
                    const int pq = (q * 64 + lane) * UNIT + (UNIT - 1) - (int)(m16q[q] & UMASK);
                    if ((m16q[q] >> UB) == top && q * 64 + lane < nun) key2 = max(key2, ((PMASK - (unsigned)canon[q]) << KB) | (unsigned)pq);
                }
                key2 = wave_max_u32(key2);
                return (int)(key2 & PMASK);
            };
            // Generations 4 / 5 speculate: the loads of the first maximum in POSITION order go out first and the tie check runs while
            // they are in flight (it fires on 0.2 % of the cycles of config 3).  The REST instantiation resolves the tie BEFORE it
            // loads: its problems split node classes into nodes with / without devices -- twin classes of one shape that tie on
            // every other cycle (config 5: 49 %), and a lost speculation costs a second round trip to the scenario's workspace.
            if (TIE_FIRST && scanned && __builtin_expect(tie_with_other_class(), 0)) {
                const int p2 = canonical_first();
                if (p2 != pstar) {
                    pstar = p2;
                    const int info = winner_info(pstar >> UB);
                    dstar = info >> 16;
                    res = (info & 0xFFFF) - 8192 + pstar;
                }
            }
            NodeState st = g_state[pstar];
            unsigned char* rowp[KQ];
            uint4 T[KQ];
            uint2 F[KQ];                                               // COARSE: the four per-16 entries of (signature, touched 64 positions)
            unsigned oldq[KQ];
#pragma unroll
            for (int q = 0; q < KQ; ++q) {
                rowp[q] = g_tile + (tile_blk((unsigned)(pstar >> 4)) + koff[q]);   // uniform table base + 32-bit byte offset
                T[q] = *(const uint4*)rowp[q];
                oldq[q] = rowp[q][pstar & 15];                         // this signature's byte before the cycle (same cache line as the row)
                if (COARSE) F[q] = *(const uint2*)(g_fine + ((unsigned)(pstar >> 6) * (unsigned)K + (unsigned)kk[q]) * 4u);
            }
            // Fold (TableScalars::static_tables & 32): required anti-affinity / host ports on node-level keys.  The landing pod's signature
            // names the signatures that may not use this node any more (TableCold::foldx, one bit per signature): their bytes go to 0 with
            // the refresh below and stay there -- the table's monotone infeasibility; summaries and counters follow as for a full node.
            // Only the two-level instantiations without the REST rows that know pinned pods carry the code (the host picks them for such
            // problems): the kernels of the benchmark configurations stay as they are.
            constexpr bool kFoldable = COARSE && !REST && HAS_PIN;   // (launch_table sends a problem with the fold to the HAS_PIN instantiations)
            const bool fold = kFoldable && (sc.static_tables & 32);
            const unsigned KW = ((unsigned)K + 31u) >> 5;
            unsigned xfold[KQ];
            if constexpr (kFoldable) {
                if (__builtin_expect(fold, 0)) {                      // (a uniform branch)
#pragma unroll
                    for (int q = 0; q < KQ; ++q) xfold[q] = gp(cold->foldx)[(unsigned)r_sig * KW + ((unsigned)kk[q] >> 5)];
                }
            }
            // MANY: the rows of the signatures beyond the register-resident ones, same round trip (uniform group conditions)
            unsigned xfg[NG ? NG : 1][KQ];
            unsigned char* rowg[NG ? NG : 1][KQ];
            uint4 Tg[NG ? NG : 1][KQ];
            uint2 Fg[NG ? NG : 1][KQ];
            unsigned oldg[NG ? NG : 1][KQ];
            int kg[NG ? NG : 1][KQ];
            if constexpr (MANY) {
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    if (128 * (g + 1) < K) {
#pragma unroll
                        for (int q = 0; q < KQ; ++q) {
                            const int k = 128 * (g + 1) + 64 * q + lane;
                            kg[g][q] = k < K ? k : 0;
                            rowg[g][q] = g_tile + (tile_blk((unsigned)(pstar >> 4)) + (unsigned)kg[g][q] * KS);
                            Tg[g][q] = *(const uint4*)rowg[g][q];
                            oldg[g][q] = rowg[g][q][pstar & 15];
                            Fg[g][q] = *(const uint2*)(g_fine + ((unsigned)(pstar >> 6) * (unsigned)K + (unsigned)kg[g][q]) * 4u);
                            if (__builtin_expect(fold, 0)) xfg[g][q] = gp(cold->foldx)[(unsigned)r_sig * KW + ((unsigned)kg[g][q] >> 5)];
                        }
                    }
                }
            }
            uint2 z = make_uint2(0, 0);
            if (!NZEQ) z = g_nz[pstar];
            if (!TIE_FIRST && scanned && __builtin_expect(tie_with_other_class(), 0)) {   // rare: first maximum in CANONICAL order
                const int p2 = canonical_first();
                if (p2 != pstar) {                                     // the speculated winner loses the tie: load the real one
                    pstar = p2;
                    const int info = winner_info(pstar >> UB);
                    dstar = info >> 16;
                    res = (info & 0xFFFF) - 8192 + pstar;
                    st = g_state[pstar];
#pragma unroll
                    for (int q = 0; q < KQ; ++q) {
                        rowp[q] = g_tile + (tile_blk((unsigned)(pstar >> 4)) + koff[q]);
                        T[q] = *(const uint4*)rowp[q];
                        oldq[q] = rowp[q][pstar & 15];
                        if (COARSE) F[q] = *(const uint2*)(g_fine + ((unsigned)(pstar >> 6) * (unsigned)K + (unsigned)kk[q]) * 4u);
                    }
                    if (!NZEQ) z = g_nz[pstar];
                }
            }
            const int blk = pstar >> 4, pos = pstar & 15;
            RestLoads RL{};
            if (REST && __builtin_expect(rw != 0, 0)) RL = rest_assume_load(pstar, r_nrows, rowv, bound ? -1 : r_gs, r_xs);
            // GPU fold: the landing position's devices travel with the assume's loads (uniform addresses)
            uint4 gfa = make_uint4(0, 0, 0, 0), gfb = make_uint4(0, 0, 0, 0);
            unsigned gft = 0;
            int gfc = 0;
            if constexpr (kGpuFoldable) {
                if (__builtin_expect(gfold, 0)) {
                    gfc = g_fc[pstar]; gft = g_ft[pstar];
                    gfa = *(const uint4*)(g_fu + (size_t)pstar * 8); gfb = *(const uint4*)(g_fu + (size_t)pstar * 8 + 4);
                }
            }
            SpreadLoads SPL{0u, 0u, false};
            if (SPREAD && sp_match != 0) SPL = spread_count_load(pstar, dstar, res, spv, spt, sp_soft, sp_match);
            const ShapeRow sh = shape_of(dstar);
            unsigned snq[KQ];
#pragma unroll
            for (int q = 0; q < KQ; ++q) snq[q] = s_sn[kk[q] * Cn + dstar];
            // byte -> u16 expansion selectors of the touched block row with the new byte already in place (kSel, above)
            const uint4 selA = ((const uint4*)kSel)[pos * 2], selB = ((const uint4*)kSel)[pos * 2 + 1];
            TPROF_WAIT_LDS; TPROF(4);                                  // loads issued; shape row and class term arrived (LDS)
            TPROF_WAIT_MEM; TPROF(5);                                  // node state and table row arrived (L2 / HBM)
            const int sl = r_sig & 63;
            const bool hiq = KQ > 1 && (r_sig >> 6);
            unsigned add_c, add_m, addz_c = 0, addz_m = 0;
            if (MANY && __builtin_expect(r_sig >= 64 * KQ, 0)) {       // a signature beyond the register-resident ones (K > 128): its row
                const SigRow rs = sigs[r_sig];                         // (uniform index: scalar loads)
                add_c = (unsigned)rs.req_c; add_m = (unsigned)rs.req_m; addz_c = (unsigned)rs.nz_c; addz_m = (unsigned)rs.nz_m;
            } else {
                add_c = (unsigned)(hiq ? __builtin_amdgcn_readlane((int)my_add_c[KQ - 1], sl) : __builtin_amdgcn_readlane((int)my_add_c[0], sl));
                add_m = (unsigned)(hiq ? __builtin_amdgcn_readlane((int)my_add_m[KQ - 1], sl) : __builtin_amdgcn_readlane((int)my_add_m[0], sl));
                if (!NZEQ) {
                    addz_c = (unsigned)(hiq ? __builtin_amdgcn_readlane((int)my_addz_c[KQ - 1], sl) : __builtin_amdgcn_readlane((int)my_addz_c[0], sl));
                    addz_m = (unsigned)(hiq ? __builtin_amdgcn_readlane((int)my_addz_m[KQ - 1], sl) : __builtin_amdgcn_readlane((int)my_addz_m[0], sl));
                }
            }
            st.rq_c += add_c;
            st.rq_m += add_m;
            st.freep -= 1u;
            double nzc = 0.0, nzm = 0.0;
            if (!NZEQ) {
                z.x += addz_c;
                z.y += addz_m;
                if (lane == 0) g_nz[pstar] = z;
                nzc = (double)z.x; nzm = (double)z.y;
            }
            if (lane == 0) g_state[pstar] = st;
            const double rq_c = (double)st.rq_c, rq_m = (double)st.rq_m;
            TPROF(8);                                                  // state update (readlanes of the signature's request), state store
            // Signature k's byte of the touched node, its block key, summary entries and feasible-node counter (lane-local k).
            auto refresh_sig = [&](int k, bool valid, unsigned nb_raw,
                                   unsigned char* rowk, const uint4 Tk, const uint2 Fk, unsigned old, unsigned snk, int dirty_bit) {
                const unsigned nb = old ? nb_raw : 0u;                    // static mask / monotone infeasibility
                // One wave: its vector memory accesses are served in order, so the next cycle's loads of this row / state
                // observe these stores; no cache maintenance, no wait.
                if (valid && nb != old) {
                    rowk[pos] = (unsigned char)nb;
                    const unsigned m = block_key16_patched(Tk, nb, selA, selB);
                    const unsigned e16 = (m >> 4) ? m + (snk << 4) : 0u;
                    if constexpr (COARSE) {
                        // per-16 entry to the workspace; the entry of the 64 positions = max over its four per-16 entries, each
                        // re-keyed to total << 6 | 63 - position (a feasible entry is >= 64, the constants alone stay below)
                        const int j = blk & 3;                            // uniform
                        g_fine[((unsigned)(pstar >> 6) * (unsigned)K + (unsigned)k) * 4u + (unsigned)j] = (unsigned short)e16;
                        const unsigned keep = (j & 1) ? 0x0000FFFFu : 0xFFFF0000u, ins = e16 << ((j & 1) * 16);
                        const unsigned fx = (j & 2) ? Fk.x : ((Fk.x & keep) | ins);
                        const unsigned fy = (j & 2) ? ((Fk.y & keep) | ins) : Fk.y;
                        const unsigned cx = (((fx & 0xFFF0FFF0u) << 2) | (fx & 0x000F000Fu)) | 0x00200030u;
                        const unsigned cy = (((fy & 0xFFF0FFF0u) << 2) | (fy & 0x000F000Fu)) | 0x00000010u;
                        const unsigned mm = pkmax_t(cx, cy);
                        const unsigned c64 = max(mm & 0xFFFFu, mm >> 16);
                        s_sum[k * nbp + (pstar >> 6)] = (unsigned short)(c64 >= 64u ? c64 : 0u);
                    } else {
                        s_sum[k * nbp + blk] = (unsigned short)e16;
                    }
                    if (!nb) {                                            // the node stopped being feasible for this signature
                        const int cidx = k * Cn + dstar;
                        int left;
                        // plain read-modify-write (lane-local: a (signature, class) counter belongs to the lane that owns the signature):
                        // an atomic is performed in L2 and would leave the wave's later plain loads of the counter to a stale L1 line
                        if constexpr (!CNT_LDS) { left = g_cnt[cidx] - 1; g_cnt[cidx] = left; }
                        else { left = s_cnt[cidx] - 1; s_cnt[cidx] = left; }
                        // The class term of row k changes: re-base before its next use -- unless (kCls4: hundreds of small classes, each leaving the
                        // feasible set of every signature at some point) the class that left sat strictly INSIDE the extremes of the last re-base:
                        // lo and hi are then attained by classes that stay, the min-max normalisation (simon.go:76-101) of every other class is
                        // what it was, and the leaver's own entries are all 0 and stay 0 (bytes never come back).  While a row is dirty the stored
                        // extremes may be stale -- it is re-based before its next use whatever this test says.  Static score tables have maxima
                        // of their own: with them every leave re-bases.
                        bool rebase = left == 0;
                        if constexpr (kCls4) {
                            if (rebase && !(sc.static_tables & 7)) {
                                const int raw = simon_raw[((KQ > 1 && dirty_bit == KQ - 1) ? my_tc[KQ - 1] : my_tc[0]) * Cn + dstar];
                                const int2 e = s_ext[k];
                                rebase = !(e.x < raw && raw < e.y);
                            }
                        }
                        if (rebase) my_dirty |= 1u << dirty_bit;
#ifdef SIMON_TABLE_DEBUG
                        printf("DBG s=%d step=%d CNT k=%d class=%d left=%d\n", s, i0 + il, k, dstar, left);
#endif
                    }
                }
            };
            // NodeResourcesFit + LeastAllocated + BalancedAllocation of the touched node per signature of this lane.  With 65 .. 128
            // signatures the host puts a signature's twin -- same request, another table class (static mask / Simon row) -- 64 slots up
            // whenever every upper signature has one (TableScalars::static_tables & 16): the upper byte is the lower lane's, unmasked.
            unsigned nbq[KQ];
            nbq[0] = eval_node(my_req_c[0], my_req_m[0], my_nz_c[0], my_nz_m[0], my_zero[0], rq_c, rq_m, nzc, nzm, (int)st.freep, sh);
            if constexpr (KQ > 1) {
                if (sc.static_tables & 16) nbq[KQ - 1] = nbq[0];
                else nbq[KQ - 1] = eval_node(my_req_c[KQ - 1], my_req_m[KQ - 1], my_nz_c[KQ - 1], my_nz_m[KQ - 1], my_zero[KQ - 1], rq_c, rq_m, nzc, nzm, (int)st.freep, sh);
            }
            if constexpr (kFoldable) {
                if (__builtin_expect(fold, 0)) {                      // folded exclusions: the signatures this landing rules out get byte 0
#pragma unroll
                    for (int q = 0; q < KQ; ++q) nbq[q] = ((xfold[q] >> (kk[q] & 31)) & 1u) ? 0u : nbq[q];
                }
            }
            unsigned gu[8] = {0, 0, 0, 0, 0, 0, 0, 0};                   // GPU fold: the landing node's devices after Reserve (MANY: the further groups read them)
            bool gbooked = false;
            if constexpr (kGpuFoldable) {
                if (__builtin_expect(gfold, 0)) {
                    // Open-Gpu-Share folded into the table: a GPU pod the scheduler placed books its devices (Reserve,
                    // open-gpu-share.go:147-188: every lane alike), and the GPU signatures whose request stopped fitting the node get
                    // byte 0 -- for good (device memory is never released).  Pods bound by Spec.NodeName never reach Reserve.
                    unsigned greq = (unsigned)(hiq ? __builtin_amdgcn_readlane((int)my_greq[KQ - 1], sl) : __builtin_amdgcn_readlane((int)my_greq[0], sl));
                    int gnum = hiq ? __builtin_amdgcn_readlane(my_gnum[KQ - 1], sl) : __builtin_amdgcn_readlane(my_gnum[0], sl);
                    if (MANY && __builtin_expect(r_sig >= 64 * KQ, 0)) {   // a signature beyond the register-resident ones: its row (uniform index)
                        const SigRow rs = sigs[r_sig];
                        greq = (unsigned)rs.pad[0]; gnum = rs.pad[1];
                    }
                    if (gnum > 0 && !bound) {
                        unsigned (&u)[8] = gu;
                        u[0] = gfa.x; u[1] = gfa.y; u[2] = gfa.z; u[3] = gfa.w; u[4] = gfb.x; u[5] = gfb.y; u[6] = gfb.z; u[7] = gfb.w;
                        gbooked = true;
                        const unsigned long long booked = gpu_commit_t(u, gfc, gft, greq, gnum);
                        if (sc.static_tables & 8) {                       // the caller wants the devices (simon_batch_out.gpu_slices), by pod id
                            const int pid = __builtin_amdgcn_readfirstlane(order[i0 + il]);
                            if (lane == 0) gp(cold->gpu_slices)[(size_t)s * (size_t)P + (size_t)pid] = booked;
                        }
                        if (lane == 0) {
                            *(uint4*)(g_fu + (size_t)pstar * 8) = make_uint4(u[0], u[1], u[2], u[3]);
                            *(uint4*)(g_fu + (size_t)pstar * 8 + 4) = make_uint4(u[4], u[5], u[6], u[7]);
                        }
#pragma unroll
                        for (int q = 0; q < KQ; ++q)
                            if (my_gnum[q] != 0 && !gpu_fits_t(u, gfc, gft, my_greq[q], my_gnum[q])) nbq[q] = 0u;
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < KQ; ++q)
                refresh_sig(kk[q], kvalid[q], nbq[q], rowp[q], T[q], COARSE ? F[q] : make_uint2(0u, 0u),
                            oldq[q], snq[q], q);
            // MANY: the further groups (their table rows arrived with group 0's; the signature rows of TableCold::sigs are L2-hot)
            if constexpr (MANY) {
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    if (128 * (g + 1) < K) {
                        SigRow rg[KQ];
                        unsigned sng[KQ];
#pragma unroll
                        for (int q = 0; q < KQ; ++q) { rg[q] = sigs[kg[g][q]]; sng[q] = s_sn[kg[g][q] * Cn + dstar]; }
#pragma unroll
                        for (int q = 0; q < KQ; ++q)
                            refresh_sig(kg[g][q], 128 * (g + 1) + 64 * q + lane < K,
                                        ((fold && ((xfg[g][q] >> (kg[g][q] & 31)) & 1u)) ||
                                         (kGpuFoldable && gbooked && rg[q].pad[1] != 0 && !gpu_fits_t(gu, gfc, gft, (unsigned)rg[q].pad[0], rg[q].pad[1]))) ? 0u :
                                        eval_node(rg[q].req_c, rg[q].req_m, rg[q].nz_c, rg[q].nz_m, rg[q].flags & 1u, rq_c, rq_m, nzc, nzm, (int)st.freep, sh),
                                        rowg[g][q], Tg[g][q], Fg[g][q], oldg[g][q], sng[q], 2 * (g + 1) + q);
                    }
                }
                // 385 .. 1 023 signatures (round 4): the groups beyond the two whose rows travel with group 0's take a round trip of their
                // own each -- a problem of that many signatures that needs generation 7's walks or a fold has the all-feature kernel as
                // its alternative, not a faster table.  (uniform trip count; a lane beyond K evaluates signature 0 and stores nothing)
                for (int g = NG; 128 * (g + 1) < K; ++g) {
#pragma unroll
                    for (int q = 0; q < KQ; ++q) {
                        const int kq = 128 * (g + 1) + 64 * q + lane;
                        const int kx = kq < K ? kq : 0;
                        unsigned char* rowx = g_tile + (tile_blk((unsigned)(pstar >> 4)) + (unsigned)kx * KS);
                        const uint4 Tx = *(const uint4*)rowx;
                        const unsigned oldx = rowx[pstar & 15];
                        const uint2 Fx = *(const uint2*)(g_fine + ((unsigned)(pstar >> 6) * (unsigned)K + (unsigned)kx) * 4u);
                        const unsigned xfx = fold ? gp(cold->foldx)[(unsigned)r_sig * KW + ((unsigned)kx >> 5)] : 0u;
                        const SigRow rx = sigs[kx];
                        const unsigned snx = s_sn[kx * Cn + dstar];
                        refresh_sig(kx, kq < K,
                                    ((fold && ((xfx >> (kx & 31)) & 1u)) ||
                                     (kGpuFoldable && gbooked && rx.pad[1] != 0 && !gpu_fits_t(gu, gfc, gft, (unsigned)rx.pad[0], rx.pad[1]))) ? 0u :
                                    eval_node(rx.req_c, rx.req_m, rx.nz_c, rx.nz_m, rx.flags & 1u, rq_c, rq_m, nzc, nzm, (int)st.freep, sh),
                                    rowx, Tx, Fx, oldx, snx, 2 * (g + 1) + q);
                    }
                }
            }
            TPROF(9);                                                  // evaluation, patch, block key, summary / table stores
            if (REST && __builtin_expect(rw != 0, 0)) rest_assume_store(RL, pstar, r_nrows, rowv, bound ? -1 : r_gs, r_xs, i0 + il);
            if (SPREAD && sp_match != 0) spread_count_store(SPL, pstar, dstar, spv, spt, sp_soft, sp_match);
            TPROF(19);                                                 // spread: counter stores
            __builtin_amdgcn_wave_barrier();
            TPROF(6);                                                  // REST: term rows, GPU commit and GPU rows
        }
#ifdef SIMON_TABLE_DEBUG
        if (lane == 0) printf("DBG s=%d step=%d sig=%d cls=%d pstar=%d dstar=%d res=%d top=%u ni=%d nblk=%d\n", s, i0 + il, r_sig, r_cls, pstar, dstar, res, top, ni, nblk);
#endif
        // -------- placement, recorded by STEP (coalesced); simon_hip.hip permutes to pod ids ---
        plreg = (il == lane) ? res : plreg;
        }
        if (lead && place && lane < steps) place[i0 + lane] = plreg >= 0 ? cls_list[rk_off + (unsigned)plreg] : plreg;   // 64 canonical indices per gather
    }

    // ---- epilogue: sum of Requested over the scenario's nodes (padding rows hold 0) -----------
    if (!lead) return;
    long long uc = 0, um = 0;
    for (int p = lane; p < ni; p += 64) { const NodeState st = g_state[p]; uc += st.rq_c; um += st.rq_m; }
    uc = wave_sum_i64(uc);
    um = wave_sum_i64(um);
#ifdef SIMON_TABLE_PROFILE
    if (lane == 0 && cold->prof)
        for (int q = 0; q < 24; ++q) gp(cold->prof)[(size_t)s * 24 + q] = tp_acc[q];
#endif
    if (lane == 0) {
        gp(cold->unscheduled)[s] = unsched;
        gp(cold->used_cpu)[s] = uc * (long long)sc.g_cpu;
        gp(cold->used_mem)[s] = um * (long long)sc.g_mem;
    }
}

// launch of one instantiation of the single-wave kernel (shared by this unit and simon_table_rest.hip: a template costs nothing where it is not used)
template <bool M, bool Z, bool PIN, int KQ, int NBQ, bool COARSE, bool REST, bool RANKED, bool AFF = false, bool MANY = false, bool SPREAD = false, bool CN2 = false>
static hipError_t launch_t7(const TableLaunch& a, int n_blocks, size_t lds, hipStream_t st) {
    if constexpr (REST && !AFF) {
        if (a.aff) return launch_t7<M, Z, PIN, KQ, NBQ, COARSE, REST, RANKED, true, false, false, CN2>(a, n_blocks, lds, st);
    }
    if (REST && !CN2 && a.sc.Cn > 64) return hipErrorInvalidValue;    // (65 .. 128 node classes under the REST select: simon_table_rest2.hip)
    if constexpr (KQ == 2 && COARSE && !MANY && !SPREAD && !CN2) {    // more than 128 signatures: the instantiation with further groups (round 6: also with the REST rows)
        if (a.sc.K > 128) return launch_t7<M, Z, PIN, KQ, NBQ, COARSE, REST, RANKED, AFF, true>(a, n_blocks, lds, st);
    }
    if (!MANY && a.sc.K > 64 * KQ) return hipErrorInvalidValue;       // simon_hip.hip keeps such batches away (two-level, no REST)
    auto kern = table_kernel<M, Z, PIN, KQ, NBQ, COARSE, REST, RANKED, AFF, MANY, SPREAD, 1, false, false, CN2>;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(n_blocks), dim3(64), lds, st, a.cold, a.cls_list, a.pods, a.orders, a.perm, a.ws_off, a.place_step, a.ws, a.sc);
    return hipGetLastError();
}
template <bool M, bool Z, bool PIN, int KQ, int NBQ, bool COARSE, bool REST = false, bool CN2 = false>
static hipError_t launch_t6(const TableLaunch& a, int n_blocks, size_t lds, hipStream_t st) {
    return a.sc.rk_stride != 0 ? launch_t7<M, Z, PIN, KQ, NBQ, COARSE, REST, true, false, false, false, CN2>(a, n_blocks, lds, st)
                               : launch_t7<M, Z, PIN, KQ, NBQ, COARSE, REST, false, false, false, false, CN2>(a, n_blocks, lds, st);
}
#ifdef SIMON_TABLE_TEAM_TU
// ---- this translation unit (simon_table_team<N>.hip) holds the team-mode instantiations only: NW = SIMON_TABLE_TEAM_TU waves per scenario ----
constexpr int kTuWaves = SIMON_TABLE_TEAM_TU;
#define SIMON_TEAM_CAT2(a, b) a##b
#define SIMON_TEAM_CAT(a, b) SIMON_TEAM_CAT2(a, b)
template <bool M, bool Z, int KQ, int NBQ, bool RANKED, bool AFF, bool CN2 = false, bool REST = false, bool MANY = false>
static hipError_t launch_team6(const TableLaunch& a, int n_blocks, size_t lds, hipStream_t st) {
    if constexpr (!CN2 && !REST) {                                    // the walks over the position-mask rows (REST && SPREAD): <= 64 classes
        if (a.rest) {
            if (a.sc.Cn > 64) return hipErrorInvalidValue;
            if constexpr (!AFF) { if (a.aff) return launch_team6<M, Z, KQ, NBQ, RANKED, true, false, true>(a, n_blocks, lds, st); }
            return launch_team6<M, Z, KQ, NBQ, RANKED, AFF, false, true>(a, n_blocks, lds, st);
        }
    }
    if (a.rest && !REST) return hipErrorInvalidValue;
    if constexpr (!CN2 && !REST) {                                    // 65 .. 128 node classes: two per lane in the walks
        if (a.sc.Cn > 64) return launch_team6<M, Z, KQ, NBQ, RANKED, AFF, true>(a, n_blocks, lds, st);
    }
    if constexpr (KQ == 2 && NBQ == 2 && !CN2 && !MANY) {             // 129 .. 1 023 signatures (round 6: the leader's refresh takes the further groups along as the single wave does)
        if (a.sc.K > 128) return launch_team6<M, Z, KQ, NBQ, RANKED, AFF, false, REST, true>(a, n_blocks, lds, st);
    }
    if ((!MANY && a.sc.K > 64 * KQ) || (!CN2 && a.sc.Cn > 64)) return hipErrorInvalidValue;
    auto kern = table_kernel<M, Z, true, KQ, NBQ, true, REST, RANKED, AFF, MANY, true, kTuWaves, false, false, CN2>;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(n_blocks), dim3(64 * kTuWaves), lds, st, a.cold, a.cls_list, a.pods, a.orders, a.perm, a.ws_off, a.place_step, a.ws, a.sc);
    return hipGetLastError();
}
template <bool M, bool Z, int KQ, int NBQ>
static hipError_t launch_team4(const TableLaunch& a, int n_blocks, size_t lds, hipStream_t st) {
    const bool ranked = a.sc.rk_stride != 0, ipa = (a.sc.static_tables & 64) != 0;     // (& 64: preferred pod (anti-)affinity / hard zone constraints: SPREAD && AFF)
    if (ranked) return ipa ? launch_team6<M, Z, KQ, NBQ, true, true>(a, n_blocks, lds, st) : launch_team6<M, Z, KQ, NBQ, true, false>(a, n_blocks, lds, st);
    return ipa ? launch_team6<M, Z, KQ, NBQ, false, true>(a, n_blocks, lds, st) : launch_team6<M, Z, KQ, NBQ, false, false>(a, n_blocks, lds, st);
}
template <bool M, bool Z>
static hipError_t launch_team2(const TableLaunch& a, int n_blocks, size_t lds, hipStream_t st) {
    const bool one = a.sc.ni_max / 64 <= 64 && a.sc.K <= 128;        // (more than 128 signatures: one shape, two entries per lane)
    if (a.sc.K > 64) return one ? launch_team4<M, Z, 2, 1>(a, n_blocks, lds, st) : launch_team4<M, Z, 2, 2>(a, n_blocks, lds, st);
    return one ? launch_team4<M, Z, 1, 1>(a, n_blocks, lds, st) : launch_team4<M, Z, 1, 2>(a, n_blocks, lds, st);
}
// The unit's instantiations in two halves by NonZero == request (Z), each a hipcc process of its own (the unit was the build's longest pole by far:
// 4 m 37 s next to 16 others on 8 cores): simon_table_team<N>z.hip defines SIMON_TABLE_NZEQ_HALF and holds Z = true, simon_table_team<N>.hip the rest + the entry.
#ifdef SIMON_TABLE_NZEQ_HALF
hipError_t SIMON_TEAM_CAT(launch_table_team_nzeq, SIMON_TABLE_TEAM_TU)(const TableLaunch& a, int n_blocks, size_t lds_bytes, hipStream_t st) {
    return launch_team2<true, true>(a, n_blocks, lds_bytes, st);
}
#else
hipError_t SIMON_TEAM_CAT(launch_table_team_nzeq, SIMON_TABLE_TEAM_TU)(const TableLaunch& a, int n_blocks, size_t lds_bytes, hipStream_t st);
hipError_t SIMON_TEAM_CAT(launch_table_team, SIMON_TABLE_TEAM_TU)(const TableLaunch& a, int n_blocks, bool has_mask, bool nzeq, size_t lds_bytes, hipStream_t st) {
    if (!a.spread || !a.coarse || a.team != kTuWaves) return hipErrorInvalidValue;
    (void)has_mask;                                                   // (a run-time test of the prologue: TableCold::static_mask is null without one)
    return nzeq ? SIMON_TEAM_CAT(launch_table_team_nzeq, SIMON_TABLE_TEAM_TU)(a, n_blocks, lds_bytes, st) : launch_team2<true, false>(a, n_blocks, lds_bytes, st);
}
#endif
#elif defined(SIMON_TABLE_SPREAD_TU)
// ---- this translation unit (simon_table_spread.hip) holds generation 7: the single-wave SPREAD instantiations (80 of the largest kernels of
// the library -- next to simon_table.hip instead of inside it, build() runs one hipcc process per unit) ----
template <bool M, bool Z, int KQ, int NBQ, bool RANKED, bool AFF, bool MANY>
static hipError_t launch_sp7(const TableLaunch& a, int n_blocks, size_t lds, hipStream_t st) {
    if (!MANY && a.sc.K > 64 * KQ) return hipErrorInvalidValue;
    auto kern = table_kernel<M, Z, true, KQ, NBQ, true, false, RANKED, AFF, MANY, true>;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(n_blocks), dim3(64), lds, st, a.cold, a.cls_list, a.pods, a.orders, a.perm, a.ws_off, a.place_step, a.ws, a.sc);
    return hipGetLastError();
}
template <bool M, bool Z, int KQ>
static hipError_t launch_sp4(const TableLaunch& a, int n_blocks, size_t lds, hipStream_t st) {
    const bool ranked = a.sc.rk_stride != 0, ipa = (a.sc.static_tables & 64) != 0;     // (& 64: preferred pod (anti-)affinity / hard zone constraints in spread_select: SPREAD && AFF)
    if constexpr (KQ == 2) {
        // 129 .. 1 023 signatures behind Services (round 4): the signature groups of MANY under generation 7's walks.  The walk reads its
        // pod's row of the table whatever the signature's number; the refresh takes the further groups along as it does without the
        // walks.  One shape (two entries per lane) serves every cluster size: the regime is rare.
        if (a.sc.K > 128) {
            if (ipa) return ranked ? launch_sp7<M, Z, KQ, 2, true, true, true>(a, n_blocks, lds, st) : launch_sp7<M, Z, KQ, 2, false, true, true>(a, n_blocks, lds, st);
            return ranked ? launch_sp7<M, Z, KQ, 2, true, false, true>(a, n_blocks, lds, st) : launch_sp7<M, Z, KQ, 2, false, false, true>(a, n_blocks, lds, st);
        }
    }
    const bool one = a.sc.ni_max / 64 <= 64;
    if (ipa) {
        if (ranked) return one ? launch_sp7<M, Z, KQ, 1, true, true, false>(a, n_blocks, lds, st) : launch_sp7<M, Z, KQ, 2, true, true, false>(a, n_blocks, lds, st);
        return one ? launch_sp7<M, Z, KQ, 1, false, true, false>(a, n_blocks, lds, st) : launch_sp7<M, Z, KQ, 2, false, true, false>(a, n_blocks, lds, st);
    }
    if (ranked) return one ? launch_sp7<M, Z, KQ, 1, true, false, false>(a, n_blocks, lds, st) : launch_sp7<M, Z, KQ, 2, true, false, false>(a, n_blocks, lds, st);
    return one ? launch_sp7<M, Z, KQ, 1, false, false, false>(a, n_blocks, lds, st) : launch_sp7<M, Z, KQ, 2, false, false, false>(a, n_blocks, lds, st);
}
template <bool M, bool Z>
static hipError_t launch_sp2(const TableLaunch& a, int n_blocks, size_t lds, hipStream_t st) {
    return a.sc.K > 64 ? launch_sp4<M, Z, 2>(a, n_blocks, lds, st) : launch_sp4<M, Z, 1>(a, n_blocks, lds, st);
}
hipError_t launch_table_spread(const TableLaunch& a, int n_blocks, bool has_mask, bool nzeq, size_t lds_bytes, hipStream_t st) {
    if (!a.spread || !a.coarse || a.rest || a.team > 1) return hipErrorInvalidValue;
    (void)has_mask;
    return nzeq ? launch_sp2<true, true>(a, n_blocks, lds_bytes, st) : launch_sp2<true, false>(a, n_blocks, lds_bytes, st);
}
#elif defined(SIMON_TABLE_SPREAD2_TU)
// ---- this translation unit (simon_table_spread2.hip) holds generation 7 for 65 .. 128 internal node classes (CN2): <= 128 signatures,
// one wave per scenario ----
template <bool Z, int KQ, int NBQ, bool RANKED = false, bool AFF = false>
static hipError_t launch_sc3(const TableLaunch& a, int n_blocks, size_t lds, hipStream_t st) {
    if constexpr (!RANKED) {
        if (a.sc.rk_stride != 0) return launch_sc3<Z, KQ, NBQ, true, AFF>(a, n_blocks, lds, st);
    }
    if constexpr (!AFF) {                                             // (& 64: preferred pod (anti-)affinity / hard zone constraints in spread_select: SPREAD && AFF)
        if (a.sc.static_tables & 64) return launch_sc3<Z, KQ, NBQ, RANKED, true>(a, n_blocks, lds, st);
    }
    if (a.sc.K > 64 * KQ) return hipErrorInvalidValue;
    auto kern = table_kernel<true, Z, true, KQ, NBQ, true, false, RANKED, AFF, false, true, 1, false, false, true>;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(n_blocks), dim3(64), lds, st, a.cold, a.cls_list, a.pods, a.orders, a.perm, a.ws_off, a.place_step, a.ws, a.sc);
    return hipGetLastError();
}
template <bool Z, int KQ>
static hipError_t launch_sc2(const TableLaunch& a, int n_blocks, size_t lds, hipStream_t st) {
    return a.sc.ni_max / 64 <= 64 ? launch_sc3<Z, KQ, 1>(a, n_blocks, lds, st) : launch_sc3<Z, KQ, 2>(a, n_blocks, lds, st);
}
hipError_t launch_table_spread2(const TableLaunch& a, int n_blocks, bool nzeq, size_t lds_bytes, hipStream_t st) {
    if (!a.spread || !a.coarse || a.rest || a.team > 1 || a.sc.Cn <= 64 || a.sc.K > 128) return hipErrorInvalidValue;
    if (a.sc.K > 64) return nzeq ? launch_sc2<true, 2>(a, n_blocks, lds_bytes, st) : launch_sc2<false, 2>(a, n_blocks, lds_bytes, st);
    return nzeq ? launch_sc2<true, 1>(a, n_blocks, lds_bytes, st) : launch_sc2<false, 1>(a, n_blocks, lds_bytes, st);
}
#elif defined(SIMON_TABLE_LDS_TU)
// ---- this translation unit (simon_table_lds.hip) holds generation 4 with the scenario's workspace in LDS (LDSWS): small batches of small problems ----
template <bool Z, int KQ, int NBQ, bool RANKED = false>
static hipError_t launch_lds3(const TableLaunch& a, int n_blocks, size_t lds, hipStream_t st) {
    if constexpr (!RANKED) {                                          // per-scenario node order (a sweep over several zones): its own instantiation
        if (a.sc.rk_stride != 0) return launch_lds3<Z, KQ, NBQ, true>(a, n_blocks, lds, st);
    }
    if (a.sc.K > 64 * KQ) return hipErrorInvalidValue;
    auto kern = table_kernel<true, Z, true, KQ, NBQ, false, false, RANKED, false, false, false, 1, true>;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(n_blocks), dim3(64), lds, st, a.cold, a.cls_list, a.pods, a.orders, a.perm, a.ws_off, a.place_step, a.ws, a.sc);
    return hipGetLastError();
}
template <bool Z, int KQ>
static hipError_t launch_lds2(const TableLaunch& a, int n_blocks, size_t lds, hipStream_t st) {
    const int nblk = a.sc.ni_max / 16;
    return nblk <= 64 ? launch_lds3<Z, KQ, 1>(a, n_blocks, lds, st) : nblk <= 128 ? launch_lds3<Z, KQ, 2>(a, n_blocks, lds, st) : launch_lds3<Z, KQ, 4>(a, n_blocks, lds, st);
}
hipError_t launch_table_lds(const TableLaunch& a, int n_blocks, bool nzeq, size_t lds_bytes, hipStream_t st) {
    if (!a.lds_ws || a.coarse || a.rest || a.spread || a.team > 1 || a.sc.K > 128 || (a.sc.static_tables & (32 | 128))) return hipErrorInvalidValue;
    if (a.sc.K > 64) return nzeq ? launch_lds2<true, 2>(a, n_blocks, lds_bytes, st) : launch_lds2<false, 2>(a, n_blocks, lds_bytes, st);
    return nzeq ? launch_lds2<true, 1>(a, n_blocks, lds_bytes, st) : launch_lds2<false, 1>(a, n_blocks, lds_bytes, st);
}
#elif defined(SIMON_TABLE_RESTLDS_TU)
// ---- this translation unit (simon_table_restlds.hip) holds generation 6 with its mask rows, row totals and canonical indices in LDS (LDSX) ----
template <bool Z, int KQ, int NBQ, bool AFF, bool RANKED = false>
static hipError_t launch_rl4(const TableLaunch& a, int n_blocks, size_t lds, hipStream_t st) {
    if constexpr (!RANKED) {
        if (a.sc.rk_stride != 0) return launch_rl4<Z, KQ, NBQ, AFF, true>(a, n_blocks, lds, st);
    }
    if (a.sc.K > 64 * KQ) return hipErrorInvalidValue;
    auto kern = table_kernel<true, Z, true, KQ, NBQ, true, true, RANKED, AFF, false, false, 1, false, true>;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(n_blocks), dim3(64), lds, st, a.cold, a.cls_list, a.pods, a.orders, a.perm, a.ws_off, a.place_step, a.ws, a.sc);
    return hipGetLastError();
}
template <bool Z, int KQ>
static hipError_t launch_rl2(const TableLaunch& a, int n_blocks, size_t lds, hipStream_t st) {
    const bool one = a.sc.ni_max / 64 <= 64;
    if (a.aff) return one ? launch_rl4<Z, KQ, 1, true>(a, n_blocks, lds, st) : launch_rl4<Z, KQ, 2, true>(a, n_blocks, lds, st);
    return one ? launch_rl4<Z, KQ, 1, false>(a, n_blocks, lds, st) : launch_rl4<Z, KQ, 2, false>(a, n_blocks, lds, st);
}
hipError_t launch_table_rest_lds(const TableLaunch& a, int n_blocks, bool nzeq, size_t lds_bytes, hipStream_t st) {
    if (!a.lds_x || !a.rest || !a.coarse || a.spread || a.team > 1) return hipErrorInvalidValue;
    if (a.sc.K > 64) return nzeq ? launch_rl2<true, 2>(a, n_blocks, lds_bytes, st) : launch_rl2<false, 2>(a, n_blocks, lds_bytes, st);
    return nzeq ? launch_rl2<true, 1>(a, n_blocks, lds_bytes, st) : launch_rl2<false, 1>(a, n_blocks, lds_bytes, st);
}
#elif defined(SIMON_TABLE_RS_TU)
// ---- this translation unit (simon_table_rs.hip) holds generation 7's walks over generation 6's position-mask rows (REST && SPREAD): one wave per
// scenario (or the team unit's four), <= 64 node classes ----
template <bool Z, int KQ, int NBQ, bool RANKED = false, bool AFF = false>
static hipError_t launch_rs3(const TableLaunch& a, int n_blocks, size_t lds, hipStream_t st) {
    if constexpr (!RANKED) {
        if (a.sc.rk_stride != 0) return launch_rs3<Z, KQ, NBQ, true, AFF>(a, n_blocks, lds, st);
    }
    if constexpr (!AFF) {                                             // (& 64: preferred pod (anti-)affinity / hard zone constraints in the walk; a.aff: required-affinity entries)
        if ((a.sc.static_tables & 64) || a.aff) return launch_rs3<Z, KQ, NBQ, RANKED, true>(a, n_blocks, lds, st);
    }
    if constexpr (KQ == 2 && NBQ == 2) {                              // 129 .. 1 023 signatures: the signature groups of MANY (one shape, as for generation 7 alone)
        if (a.sc.K > 128) {
            auto kern = table_kernel<true, Z, true, 2, 2, true, true, RANKED, AFF, true, true>;
            hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(kern, dim3(n_blocks), dim3(64), lds, st, a.cold, a.cls_list, a.pods, a.orders, a.perm, a.ws_off, a.place_step, a.ws, a.sc);
            return hipGetLastError();
        }
    }
    if (a.sc.K > 64 * KQ) return hipErrorInvalidValue;
    auto kern = table_kernel<true, Z, true, KQ, NBQ, true, true, RANKED, AFF, false, true>;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(n_blocks), dim3(64), lds, st, a.cold, a.cls_list, a.pods, a.orders, a.perm, a.ws_off, a.place_step, a.ws, a.sc);
    return hipGetLastError();
}
template <bool Z, int KQ>
static hipError_t launch_rs2(const TableLaunch& a, int n_blocks, size_t lds, hipStream_t st) {
    if constexpr (KQ == 2) { if (a.sc.K > 128) return launch_rs3<Z, KQ, 2>(a, n_blocks, lds, st); }
    return a.sc.ni_max / 64 <= 64 ? launch_rs3<Z, KQ, 1>(a, n_blocks, lds, st) : launch_rs3<Z, KQ, 2>(a, n_blocks, lds, st);
}
// (two halves by NonZero == request, as the team unit: simon_table_rsz.hip holds Z = true, simon_table_rs.hip the rest + the entry)
#ifdef SIMON_TABLE_NZEQ_HALF
hipError_t launch_table_rs_nzeq(const TableLaunch& a, int n_blocks, size_t lds_bytes, hipStream_t st) {
    return a.sc.K > 64 ? launch_rs2<true, 2>(a, n_blocks, lds_bytes, st) : launch_rs2<true, 1>(a, n_blocks, lds_bytes, st);
}
#else
hipError_t launch_table_rs_nzeq(const TableLaunch& a, int n_blocks, size_t lds_bytes, hipStream_t st);
hipError_t launch_table_rs(const TableLaunch& a, int n_blocks, bool nzeq, size_t lds_bytes, hipStream_t st) {
    if (!a.spread || !a.rest || !a.coarse || a.team > 1 || a.lds_x || a.sc.Cn > 64 || (a.sc.static_tables & (32 | 128))) return hipErrorInvalidValue;
    if (nzeq) return launch_table_rs_nzeq(a, n_blocks, lds_bytes, st);
    return a.sc.K > 64 ? launch_rs2<false, 2>(a, n_blocks, lds_bytes, st) : launch_rs2<false, 1>(a, n_blocks, lds_bytes, st);
}
#endif
#elif defined(SIMON_TABLE_CLS4_TU)
// ---- this translation unit (simon_table_cls4.hip) holds generations 4 for 129 .. 256 internal node classes: one-level layout, > 2 048 padded positions (a class
// segment is padded to 16), so NBQ = 4; the instantiation that knows pinned pods serves every problem ----
template <bool Z>
static hipError_t launch_cls4b(const TableLaunch& a, int n_blocks, size_t lds, hipStream_t st) {
    return a.sc.K > 64 ? launch_t6<true, Z, true, 2, 4, false, false, true>(a, n_blocks, lds, st) : launch_t6<true, Z, true, 1, 4, false, false, true>(a, n_blocks, lds, st);
}
hipError_t launch_table_cls4(const TableLaunch& a, int n_blocks, bool nzeq, size_t lds_bytes, hipStream_t st) {
    if (a.rest || a.spread || a.coarse || a.team > 1 || a.lds_ws || a.sc.Cn <= 128 || a.sc.Cn > kTableMaxClassesPlain || a.sc.K > 128 || (a.sc.static_tables & (32 | 128))) return hipErrorInvalidValue;
    return nzeq ? launch_cls4b<true>(a, n_blocks, lds_bytes, st) : launch_cls4b<false>(a, n_blocks, lds_bytes, st);
}
#elif defined(SIMON_TABLE_REST2_TU)
// ---- this translation unit (simon_table_rest2.hip) holds generation 6 for 65 .. 128 internal node classes (CN2 in rest_select; rows in HBM) ----
template <bool Z, int KQ>
static hipError_t launch_rest2c(const TableLaunch& a, int n_blocks, size_t lds, hipStream_t st) {
    return a.sc.ni_max / 64 <= 64 ? launch_t6<true, Z, true, KQ, 1, true, true, true>(a, n_blocks, lds, st)
                                  : launch_t6<true, Z, true, KQ, 2, true, true, true>(a, n_blocks, lds, st);
}
hipError_t launch_table_rest2(const TableLaunch& a, int n_blocks, bool nzeq, size_t lds_bytes, hipStream_t st) {
    if (!a.rest || !a.coarse || a.spread || a.team > 1 || a.lds_x || a.sc.Cn <= 64) return hipErrorInvalidValue;
    if (a.sc.K > 64) return nzeq ? launch_rest2c<true, 2>(a, n_blocks, lds_bytes, st) : launch_rest2c<false, 2>(a, n_blocks, lds_bytes, st);
    return nzeq ? launch_rest2c<true, 1>(a, n_blocks, lds_bytes, st) : launch_rest2c<false, 1>(a, n_blocks, lds_bytes, st);
}
#elif defined(SIMON_TABLE_REST_TU)
// ---- this translation unit (simon_table_rest.hip) holds generation 6: the REST instantiations (position masks: Open-Gpu-Share devices,
// required (anti-)affinity, host ports, ephemeral storage / extended resources) -- the 32 largest kernels of the single-wave family ----
template <bool Z, int KQ>
static hipError_t launch_rest2(const TableLaunch& a, int n_blocks, size_t lds, hipStream_t st) {
    return a.sc.ni_max / 64 <= 64 ? launch_t6<true, Z, true, KQ, 1, true, true>(a, n_blocks, lds, st)
                                  : launch_t6<true, Z, true, KQ, 2, true, true>(a, n_blocks, lds, st);
}
hipError_t launch_table_rest(const TableLaunch& a, int n_blocks, bool nzeq, size_t lds_bytes, hipStream_t st) {
    if (!a.rest || !a.coarse || a.spread || a.team > 1) return hipErrorInvalidValue;
    if (a.sc.K > 64) return nzeq ? launch_rest2<true, 2>(a, n_blocks, lds_bytes, st) : launch_rest2<false, 2>(a, n_blocks, lds_bytes, st);
    return nzeq ? launch_rest2<true, 1>(a, n_blocks, lds_bytes, st) : launch_rest2<false, 1>(a, n_blocks, lds_bytes, st);
}
#else
// placement[s][pod] = place_step[s][inv_order[order_id(s)][pod]]: gather (scattered reads hit L2, stores coalesced)
__global__ __launch_bounds__(256) void unpermute_kernel(const int32_t* __restrict__ place_step, const int32_t* __restrict__ inv_orders,
                                                        const ScenarioDesc* __restrict__ scen, int P, int32_t* __restrict__ placement) {
    const int s = blockIdx.y;
    const int32_t* __restrict__ inv = inv_orders + (size_t)scen[s].order_id * P;
    const int32_t* __restrict__ src = place_step + (size_t)s * P;
    int32_t* __restrict__ dst = placement + (size_t)s * P;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) dst[p] = src[inv[p]];
}

// The same through LDS, one workgroup per scenario, for rows of at most 64 KB (P <= 16 384; config 3: 40 KB, four workgroups per CU): the
// row arrives coalesced and the gather runs in LDS -- 64 scattered dwords of a 40 KB row cost the texture path one cache line each
// (config 3's 4 096 rows: 154 us per batch, 1.1 % of the step; the row is read once and written once either way).
__global__ __launch_bounds__(512) void unpermute_lds_kernel(const int32_t* __restrict__ place_step, const int32_t* __restrict__ inv_orders,
                                                            const ScenarioDesc* __restrict__ scen, int P, int32_t* __restrict__ placement) {
    extern __shared__ int32_t s_unp_row[];
    const int s = blockIdx.x;
    const int32_t* __restrict__ inv = inv_orders + (size_t)scen[s].order_id * P;
    const int32_t* __restrict__ src = place_step + (size_t)s * P;
    int32_t* __restrict__ dst = placement + (size_t)s * P;
    for (int i = threadIdx.x; i < P; i += 512) s_unp_row[i] = src[i];
    __syncthreads();
    for (int p = threadIdx.x; p < P; p += 512) dst[p] = s_unp_row[inv[p]];
}

hipError_t launch_unpermute(const int32_t* place_step, const int32_t* inv_orders, const ScenarioDesc* scen, int S, int P,
                            int32_t* placement, hipStream_t st) {
    if (S <= 0 || P <= 0) return hipSuccess;
    if (P >= 2048 && P <= 16384) {
        hipLaunchKernelGGL(unpermute_lds_kernel, dim3(S), dim3(512), (size_t)P * 4, st, place_step, inv_orders, scen, P, placement);
        return hipGetLastError();
    }
    const int bx = std::min(64, (P + 255) / 256);
    for (int s0 = 0; s0 < S; s0 += 65535) {   // gridDim.y limit
        const int ns = std::min(65535, S - s0);
        hipLaunchKernelGGL(unpermute_kernel, dim3(bx, ns), dim3(256), 0, st, place_step + (size_t)s0 * P, inv_orders, scen + s0, P,
                           placement + (size_t)s0 * P);
    }
    return hipGetLastError();
}

template <bool M, bool Z, bool PIN, int KQ>
static hipError_t launch_t4(const TableLaunch& a, int n_blocks, size_t lds, hipStream_t st) {
    if constexpr (PIN) {                                              // REST rides on the instantiation that knows pinned pods: simon_table_rest.hip
        if (a.rest && a.sc.Cn > 64) return launch_table_rest2(a, n_blocks, Z, lds, st);   // two node classes per lane in the REST select: simon_table_rest2.hip
        if (a.rest) return a.lds_x ? launch_table_rest_lds(a, n_blocks, Z, lds, st) : launch_table_rest(a, n_blocks, Z, lds, st);
    }
    if (a.spread) return hipErrorInvalidValue;                        // (generation 7 lives in simon_table_spread.hip: launch_table_spread)
    if (a.coarse) {                                                   // entries of 64 positions: <= 8192 padded positions
        return a.sc.ni_max / 64 <= 64 ? launch_t6<M, Z, PIN, KQ, 1, true>(a, n_blocks, lds, st) : launch_t6<M, Z, PIN, KQ, 2, true>(a, n_blocks, lds, st);
    }
    if (a.sc.Cn > 128) return launch_table_cls4(a, n_blocks, Z, lds, st);   // 129 .. 256 node classes: simon_table_cls4.hip
    const int nblk = a.sc.ni_max / 16;
    return nblk <= 64 ? launch_t6<M, Z, PIN, KQ, 1, false>(a, n_blocks, lds, st)
           : nblk <= 128 ? launch_t6<M, Z, PIN, KQ, 2, false>(a, n_blocks, lds, st) : launch_t6<M, Z, PIN, KQ, 4, false>(a, n_blocks, lds, st);
}

size_t table_lds_bytes(int K, int ni_max, int Cn, bool coarse, bool rest, int nzk) { return (size_t)tcarve(K, ni_max, Cn, coarse, rest, nzk, !coarse && Cn > 128).total; }
size_t table_ws_bytes(int K, int ni, bool nzeq, bool coarse, int Cn, int M, int NZ, int TH, int TZ) { return table_ws_of(K, ni, nzeq, coarse, Cn, M, NZ, TH, TZ, !coarse && Cn > 128); }

template <bool M, bool Z, bool PIN>
static hipError_t launch_t3(const TableLaunch& a, int n_blocks, size_t lds, hipStream_t st) {
    return a.sc.K > 64 ? launch_t4<M, Z, PIN, 2>(a, n_blocks, lds, st) : launch_t4<M, Z, PIN, 1>(a, n_blocks, lds, st);
}
template <bool M, bool Z>
static hipError_t launch_t2(const TableLaunch& a, int n_blocks, bool has_pin, size_t lds, hipStream_t st) {
    return has_pin ? launch_t3<M, Z, true>(a, n_blocks, lds, st) : launch_t3<M, Z, false>(a, n_blocks, lds, st);
}

hipError_t launch_table(const TableLaunch& a, int n_blocks, bool has_mask, bool nzeq, bool has_pin, size_t lds_bytes, hipStream_t st) {
    if (a.team > 1)                                                   // several waves per scenario: simon_table_team<N>.hip
        return a.team == kTeamWaves ? launch_table_team4(a, n_blocks, has_mask, nzeq, lds_bytes, st) : hipErrorInvalidValue;
    if (a.spread && a.rest) return launch_table_rs(a, n_blocks, nzeq, lds_bytes, st);             // generation 7 over the position-mask rows: simon_table_rs.hip
    if (a.spread && a.sc.Cn > 64) return launch_table_spread2(a, n_blocks, nzeq, lds_bytes, st);   // generation 7, two node classes per lane: simon_table_spread2.hip
    if (a.spread) return launch_table_spread(a, n_blocks, has_mask, nzeq, lds_bytes, st);   // generation 7: simon_table_spread.hip
    if (a.lds_ws) return launch_table_lds(a, n_blocks, nzeq, lds_bytes, st);                // generation 4, workspace in LDS: simon_table_lds.hip
    has_pin = has_pin || a.rest || (a.sc.static_tables & (32 | 128));   // (& 32, & 128: the folds, carried by COARSE && !REST && HAS_PIN)
    (void)has_mask;                                                   // (HAS_MASK is a prologue-only, run-time test since round 5: one instantiation serves both)
    return nzeq ? launch_t2<true, true>(a, n_blocks, has_pin, lds_bytes, st) : launch_t2<true, false>(a, n_blocks, has_pin, lds_bytes, st);
}
#endif  // SIMON_TABLE_TEAM_TU

}  // namespace simon
