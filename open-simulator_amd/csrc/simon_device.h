// simon_device.h -- shared device-side definitions for the gfx950 scenario kernels.
//
// Arithmetic contract (reference lines cited at each use in the kernels):
//   * integer scores use truncating int64 semantics of Go; the NARROW kernel evaluates them in
//     fp64 with a half-unit bias that is provably exact for operands < 2^31 (see la_term);
//   * BalancedAllocation is IEEE binary64 with NO fma contraction (compile with
//     -ffp-contract=off); explicit fma() calls are part of the proofs, not contractions.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace simon {

constexpr int kWave = 64;

// One pod of the stream, NARROW layout (32 B, fetched with one s_load_dwordx8).
// Quantities are gcd-normalised (cpu / g_cpu, mem / g_mem) and < 2^31.
struct PodRowN {
    uint32_t req_cpu, req_mem;  // computePodResourceRequest, V/framework/plugins/noderesources/fit.go:148-165
    uint32_t nz_cpu, nz_mem;    // calculateResource non-zero request, V/framework/types.go:601-636
    int32_t cls;                // pod class: row of static_mask / simon_raw
    int32_t preset;             // Spec.NodeName preset (>= 0) or -1
    int32_t gate;               // pod exists only when n_nodes > gate
    uint32_t flags;             // bit0: request is all-zero (fit.go:244-249 early return)
};
static_assert(sizeof(PodRowN) == 32, "PodRowN must be 32 bytes");

struct ScenarioDesc {
    int32_t n_nodes;
    int32_t order_id;
};

struct NarrowArgs {
    // node pool, static (shared by every scenario)
    const uint32_t* __restrict__ a_cpu;    // [N] Allocatable.MilliCPU / g_cpu
    const uint32_t* __restrict__ a_mem;    // [N] Allocatable.Memory / g_mem
    const int32_t* __restrict__ a_pods;    // [N] AllowedPodNumber
    const int32_t* __restrict__ ncls;      // [N] node class (simon_raw column)
    // initial dynamic state (pods bound before the stream)
    const uint32_t* __restrict__ i_rq_cpu; // [N]
    const uint32_t* __restrict__ i_rq_mem;
    const uint32_t* __restrict__ i_nz_cpu;
    const uint32_t* __restrict__ i_nz_mem;
    const int32_t* __restrict__ i_npods;
    // pod stream
    const PodRowN* __restrict__ pods;      // [P]
    const int32_t* __restrict__ orders;    // [n_orders][P]
    const ScenarioDesc* __restrict__ scen; // [S]
    const int32_t* __restrict__ perm;      // [S] launch order (largest scenarios first)
    const uint64_t* __restrict__ static_mask; // [Cp][mask_words] or nullptr
    const int32_t* __restrict__ simon_raw; // [Cp][Cn]
    int32_t mask_words, Cn, Cp, P, S;
    uint64_t g_cpu, g_mem;                 // de-normalisation factors for used_cpu / used_mem
    // outputs
    int32_t* __restrict__ unscheduled;     // [S]
    int64_t* __restrict__ used_cpu;        // [S]
    int64_t* __restrict__ used_mem;        // [S]
    int32_t* __restrict__ placement;       // [S][P] by pod id, or nullptr
};

// ---- wave64 reductions on DPP (gfx9 row/bank permutes; result is wave-uniform) -----------------
// quad_perm [1,0,3,2] / [2,3,0,1], row_half_mirror, row_mirror make every 16-lane row uniform;
// row_bcast15 (rows 1,3) and row_bcast31 (rows 2,3) fold the rows into lane 63.
#define SIMON_DPP(old, v, ctrl, rowmask) __builtin_amdgcn_update_dpp((old), (v), (ctrl), (rowmask), 0xF, false)

__device__ __forceinline__ int wave_min_i32(int v) {
    const int id = 0x7fffffff;
    v = min(v, SIMON_DPP(id, v, 0xB1, 0xF));
    v = min(v, SIMON_DPP(id, v, 0x4E, 0xF));
    v = min(v, SIMON_DPP(id, v, 0x141, 0xF));
    v = min(v, SIMON_DPP(id, v, 0x140, 0xF));
    v = min(v, SIMON_DPP(id, v, 0x142, 0xA));
    v = min(v, SIMON_DPP(id, v, 0x143, 0xC));
    return __builtin_amdgcn_readlane(v, 63);
}

__device__ __forceinline__ int wave_max_i32(int v) {
    const int id = (int)0x80000000;
    v = max(v, SIMON_DPP(id, v, 0xB1, 0xF));
    v = max(v, SIMON_DPP(id, v, 0x4E, 0xF));
    v = max(v, SIMON_DPP(id, v, 0x141, 0xF));
    v = max(v, SIMON_DPP(id, v, 0x140, 0xF));
    v = max(v, SIMON_DPP(id, v, 0x142, 0xA));
    v = max(v, SIMON_DPP(id, v, 0x143, 0xC));
    return __builtin_amdgcn_readlane(v, 63);
}

__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
    const int id = 0;
    v = max(v, (unsigned)SIMON_DPP(id, (int)v, 0xB1, 0xF));
    v = max(v, (unsigned)SIMON_DPP(id, (int)v, 0x4E, 0xF));
    v = max(v, (unsigned)SIMON_DPP(id, (int)v, 0x141, 0xF));
    v = max(v, (unsigned)SIMON_DPP(id, (int)v, 0x140, 0xF));
    v = max(v, (unsigned)SIMON_DPP(id, (int)v, 0x142, 0xA));
    v = max(v, (unsigned)SIMON_DPP(id, (int)v, 0x143, 0xC));
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

__device__ __forceinline__ long long wave_sum_i64(long long v) {
    // used once per scenario (epilogue): plain shuffles are fine here
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// ---- exact integer score terms in fp64 ---------------------------------------------------------
// leastRequestedScore (V/framework/plugins/noderesources/least_allocated.go:108-117):
//   capacity == 0 || requested > capacity ? 0 : ((capacity - requested) * 100) / capacity
// NARROW operands: cap, req < 2^31.  t = (cap-req)*100 < 2^38 is exact in fp64.  With
// rc = RN(1/cap) and v = fma(t, rc, 0.5*rc) = ((t + 0.5)/cap)(1+e), |e| <= 2^-52 + 2^-53, the
// absolute error is < 101 * 2^-51 < 5e-14, while (t+0.5)/cap is at least 0.5/cap > 2.3e-10 away
// from every integer (t/cap has a fractional part in [0, 1-1/cap]).  Hence trunc(v) == t / cap.
__device__ __forceinline__ int la_term(uint32_t cap, uint32_t req, double rc, double half_rc) {
    const double t = (double)(cap - req) * 100.0;
    const int q = (int)__builtin_fma(t, rc, half_rc);
    return (cap == 0u || req > cap) ? 0 : q;
}

// fractionOfCapacity + balancedResourceScorer (balanced_allocation.go:82-119), IEEE binary64:
//   cf = cap==0 ? 1 : float64(req)/float64(cap);  (cf>=1 || mf>=1) ? 0 : int64((1-|cf-mf|)*100)
__device__ __forceinline__ int ba_term(uint32_t cap_c, uint32_t req_c, uint32_t cap_m, uint32_t req_m) {
    const double cf = cap_c == 0u ? 1.0 : (double)req_c / (double)cap_c;
    const double mf = cap_m == 0u ? 1.0 : (double)req_m / (double)cap_m;
    const double diff = __builtin_fabs(cf - mf);
    const int s = (int)((1.0 - diff) * 100.0);
    return (cf >= 1.0 || mf >= 1.0) ? 0 : s;
}

// Correctly rounded req/cap from a precomputed rc = RN(1/cap): two Markstein corrections.
// q0 = RN(req*rc) is within 1.5 ulp; one residual step makes it faithful, the second one
// (with rc strictly within half an ulp of 1/cap, true for every integer cap) rounds correctly
// (Markstein 1990; Muller et al., Handbook of FP Arithmetic, thm. 4.7).  All operands are
// integers < 2^31, far from under/overflow.  Verified against '/' on-device in tests.
__device__ __forceinline__ double div_by_rcp(double num, double den, double rc) {
    double q = num * rc;
    double e = __builtin_fma(-q, den, num);
    q = __builtin_fma(e, rc, q);
    e = __builtin_fma(-q, den, num);
    q = __builtin_fma(e, rc, q);
    return q;
}

// The same quotient with ONE correction, exact on the score-table kernel's domain: den = b an integer in [1, 2^31), num = a
// an integer in [0, 2^32), rc = RN(1/b).  q0 = RN(a rc) = (a/b)(1+d1)(1+d2), |d| <= u = 2^-53; the residual e = a - q0 b is
// exact (fma); q0 + e rc = a/b + (a/b - q0) d1 differs from a/b by at most 2.01 u^2 (a/b), i.e. 2^-105 relative.  A midpoint m
// between two doubles near a/b is M 2^e with M an odd 54-bit integer and 2^-e an integer (a/b < 2^32): |a/b - m| =
// |a 2^-e - b M| 2^e / b is a non-zero multiple of 2^e / b (a/b = m would need a >= M > 2^53), so a/b stays at least
// 2^-84 (relative) away from every midpoint -- the 2^-105 perturbation cannot change the rounding: q1 = RN(a/b).
// Checked against '/' on 9.3e8 operand pairs of that domain by oracle/div_by_rcp_check.c (0 mismatches).
__device__ __forceinline__ double div_by_rcp1(double num, double den, double rc) {
    const double q = num * rc;
    const double e = __builtin_fma(-q, den, num);
    return __builtin_fma(e, rc, q);
}

__device__ __forceinline__ int ba_term_rcp(uint32_t cap_c, uint32_t req_c, double rc_c, uint32_t cap_m,
                                           uint32_t req_m, double rc_m) {
    const double cf = cap_c == 0u ? 1.0 : div_by_rcp((double)req_c, (double)cap_c, rc_c);
    const double mf = cap_m == 0u ? 1.0 : div_by_rcp((double)req_m, (double)cap_m, rc_m);
    const double diff = __builtin_fabs(cf - mf);
    const int s = (int)((1.0 - diff) * 100.0);
    return (cf >= 1.0 || mf >= 1.0) ? 0 : s;
}

}  // namespace simon
