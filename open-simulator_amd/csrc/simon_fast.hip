// simon_fast.hip -- NARROW kernel, second generation (the one bench.py measures).
//
// Same mapping as simon_narrow.hip (one workgroup = one scenario, node j in slot j/T of lane
// j%T, node state in registers) with the instruction stream cut down:
//   * every wave-uniform value (pod row, reductions, branch conditions) lives in SGPRs: the pod
//     stream comes through scalar loads (restrict-qualified kernel parameters) and cross-wave
//     results are re-uniformised with readfirstlane, so control flow is scalar branches;
//   * node quantities are kept as exact fp64 integers (< 2^31 after gcd normalisation), so the
//     filter is two v_add_f64 + two v_cmp_f64 and the scores need no int->fp conversions;
//   * phase A reduces ONE 32-bit mask "node classes that have a feasible node" (DPP v_or) instead
//     of (lo, hi); SimonPlugin.NormalizeScore then becomes a per-(pod class, node class) table
//     lookup: precomputed in LDS for the common case "every class present is feasible",
//     recomputed by lanes < Cn otherwise;
//   * consecutive pods with an identical request signature (replicas of one workload) reuse the
//     per-node base score LeastAllocated+BalancedAllocation: only the node touched by the
//     previous assume is re-evaluated (exactly the same arithmetic, so still bit-identical).
// Requires Cn <= 32 node classes and no zero-capacity node; otherwise the host dispatches
// simon_narrow.hip.
#include "simon_device.h"

#include <type_traits>

namespace simon {

template <int K, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (K < N) {
        f(std::integral_constant<int, K>{});
        static_for<K + 1, N>(f);
    }
}

// 48-byte pod row: gcd-normalised quantities as exact doubles.
struct PodRowF {
    double req_c, req_m;   // computePodResourceRequest (fit.go:148-165)
    double nz_c, nz_m;     // non-zero request (V/framework/types.go:601-636)
    int32_t cls, preset, gate;
    uint32_t flags;        // bit0: all-zero request (fit.go:244-249)
};
static_assert(sizeof(PodRowF) == 48, "PodRowF must be 48 bytes");

__device__ __forceinline__ unsigned wave_or_u32(unsigned v) {
    v |= (unsigned)SIMON_DPP(0, (int)v, 0xB1, 0xF);
    v |= (unsigned)SIMON_DPP(0, (int)v, 0x4E, 0xF);
    v |= (unsigned)SIMON_DPP(0, (int)v, 0x141, 0xF);
    v |= (unsigned)SIMON_DPP(0, (int)v, 0x140, 0xF);
    v |= (unsigned)SIMON_DPP(0, (int)v, 0x142, 0xA);
    v |= (unsigned)SIMON_DPP(0, (int)v, 0x143, 0xC);
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// leastRequestedScore term for r < cap (least_allocated.go:108-117):
//   ((cap - r) * 100) / cap = floor(100 - 100 r / cap).
// v = fma(-r, rc100, 100 + 2^-33) with rc100 = RN(100 * RN(1/cap)):  |v - (x + 2^-33)| <= 3e-14
// where x = 100 (cap - r)/cap has a fractional part in [0, 1 - 1/cap]; 3e-14 < 2^-33 and
// 2^-33 + 3e-14 < 1/cap for cap <= 2^31, so trunc(v) == floor(x) exactly.
__device__ __forceinline__ int la_term_f(double r, double rc100) {
    const double C = 100.0 + 0x1.0p-33;
    return (int)__builtin_fma(-r, rc100, C);
}

struct FastScalars {
    int32_t mask_words, Cn, Cp, P, S;
    uint64_t g_cpu, g_mem;
};

template <int T, int SLOTS, bool HAS_MASK, bool NZEQ>
__global__ __launch_bounds__(T) void fast_kernel(
    const uint32_t* __restrict__ a_cpu, const uint32_t* __restrict__ a_mem, const int32_t* __restrict__ a_pods,
    const int32_t* __restrict__ ncls, const uint32_t* __restrict__ i_rq_cpu, const uint32_t* __restrict__ i_rq_mem,
    const uint32_t* __restrict__ i_nz_cpu, const uint32_t* __restrict__ i_nz_mem, const int32_t* __restrict__ i_npods,
    const PodRowF* __restrict__ pods, const int32_t* __restrict__ orders, const ScenarioDesc* __restrict__ scen,
    const int32_t* __restrict__ perm, const uint64_t* __restrict__ static_mask, const int32_t* __restrict__ simon_raw,
    int32_t* __restrict__ unscheduled, int64_t* __restrict__ used_cpu, int64_t* __restrict__ used_mem,
    int32_t* __restrict__ placement, const FastScalars sc) {
    constexpr int NW = T / kWave;
    __shared__ unsigned mbA[2][NW];   // phase A mailboxes: feasible-class mask per wave
    __shared__ unsigned mbB[2][NW];   // phase B mailboxes: best key per wave
    __shared__ int s_tmp[NW][32];     // per-wave SN row when some class has no feasible node
    __shared__ long long red[2][NW];
    extern __shared__ __attribute__((aligned(16))) int s_dyn[];  // [Cp*Cn] raw | [Cp*Cn] sn2_full
    const int Cn = sc.Cn, Cp = sc.Cp, P = sc.P;
    int* s_raw = s_dyn;
    int* s_sn2 = s_dyn + Cp * Cn;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int lane = tid & (kWave - 1);
    const int s = perm[blockIdx.x];
    const int n = scen[s].n_nodes;
    const int32_t* __restrict__ order = orders + (size_t)scen[s].order_id * P;
    const int nslots = (n + T - 1) / T;

    for (int i = tid; i < Cp * Cn; i += T) s_raw[i] = simon_raw[i];

    // ---- prologue: node rows -> registers (exact fp64 integers) -----------------------------
    double cap_c[SLOTS], cap_m[SLOTS], rc_c[SLOTS], rc_m[SLOTS], rc100_c[SLOTS], rc100_m[SLOTS];
    double rq_c[SLOTS], rq_m[SLOTS], nzs_c[NZEQ ? 1 : SLOTS], nzs_m[NZEQ ? 1 : SLOTS];
    int freep[SLOTS], ncl4[SLOTS], base[SLOTS];
    unsigned clsbit[SLOTS];
    unsigned bits_all = 0;
#pragma unroll
    for (int k = 0; k < SLOTS; ++k) {
        const int j = k * T + tid;
        const bool in = j < n;
        const uint32_t ac = in ? a_cpu[j] : 1u, am = in ? a_mem[j] : 1u;
        cap_c[k] = (double)ac;
        cap_m[k] = (double)am;
        rc_c[k] = 1.0 / cap_c[k];           // correctly rounded, once per scenario
        rc_m[k] = 1.0 / cap_m[k];
        rc100_c[k] = 100.0 * rc_c[k];
        rc100_m[k] = 100.0 * rc_m[k];
        rq_c[k] = in ? (double)i_rq_cpu[j] : 0.0;
        rq_m[k] = in ? (double)i_rq_mem[j] : 0.0;
        if (!NZEQ) {
            nzs_c[k] = in ? (double)i_nz_cpu[j] : 0.0;
            nzs_m[k] = in ? (double)i_nz_mem[j] : 0.0;
        }
        // free pod slots; 0 for lanes beyond n, so they are never feasible (fit.go:233-242)
        freep[k] = in ? a_pods[j] - i_npods[j] : 0;
        const int c = in ? ncls[j] : 0;
        ncl4[k] = c * 4;
        clsbit[k] = in ? (1u << c) : 0u;
        bits_all |= clsbit[k];
        base[k] = 0;
    }
    const unsigned idxb = 0xFFFFFu - (unsigned)tid;   // key low bits of slot k: idxb - k*T
    bits_all = wave_or_u32(bits_all);
    if (NW > 1) {
        if (lane == 0) mbA[0][wave] = bits_all;
        __syncthreads();
#pragma unroll
        for (int w = 0; w < NW; ++w) bits_all |= mbA[0][w];
        bits_all = __builtin_amdgcn_readfirstlane(bits_all);
    }
    __syncthreads();  // s_raw complete; mbA[0] free again
    // sn2_full[c][d] = 2 * NormalizeScore(raw[c][d]) over the node classes present in this
    // scenario (pkg/simulator/plugin/simon.go:76-101), exact int64 arithmetic, once.
    for (int i = tid; i < Cp * Cn; i += T) {
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

    int unsched = 0, bufA = 1, bufB = 0;
    int32_t* __restrict__ place = placement ? placement + (size_t)s * P : nullptr;

    // request-signature cache: base[] / feas hold scores for (sig_*); one slot may be dirty
    bool cache_valid = false;
    double sig_rc = -1.0, sig_rm = -1.0, sig_nc = -1.0, sig_nm = -1.0;
    int sig_cls = -1, dirty_wave = -1, dirty_k = -1;
    unsigned feas = 0;

    int pid_next = P > 0 ? order[0] : 0;
    PodRowF row_next = pods[pid_next];
    int pid_next2 = P > 1 ? order[1] : 0;

    for (int i = 0; i < P; ++i) {
        const int pid = pid_next;
        const PodRowF row = row_next;
        pid_next = pid_next2;
        row_next = pods[pid_next];
        pid_next2 = (i + 2 < P) ? order[i + 2] : 0;

        if (row.gate >= n) {
            if (place && tid == 0) place[pid] = -2;
            continue;
        }
        int jstar;
        if (row.preset >= 0) {   // addPodToCache path (V/eventhandlers.go:223-236)
            jstar = row.preset;
            cache_valid = false;
        } else {
            const bool zero_req = row.flags & 1u;
            const uint64_t* __restrict__ mrow = HAS_MASK ? static_mask + (size_t)row.cls * sc.mask_words : nullptr;
            const bool same = cache_valid && row.req_c == sig_rc && row.req_m == sig_rm && row.nz_c == sig_nc &&
                              row.nz_m == sig_nm && row.cls == sig_cls;
            // -------- evaluate slots: filter (fit.go:230-302) + LeastAllocated + BalancedAllocation
            auto eval_slot = [&](auto kc) {
                constexpr int k = decltype(kc)::value;
                const double t_c = rq_c[k] + row.req_c, t_m = rq_m[k] + row.req_m;
                bool ok = freep[k] >= 1;
                const bool res_ok = (cap_c[k] >= t_c) && (cap_m[k] >= t_m);
                ok = ok && (zero_req || res_ok);
                if (HAS_MASK) {
                    const int j = k * T + tid;
                    ok = ok && ((mrow[(j < n ? j : 0) >> 6] >> (j & 63)) & 1ull);
                }
                feas = ok ? (feas | (1u << k)) : (feas & ~(1u << k));
                // resource_allocation.go:91-98: requested = NonZeroRequested + pod non-zero request
                const double r_c = NZEQ ? t_c : nzs_c[NZEQ ? 0 : k] + row.nz_c;
                const double r_m = NZEQ ? t_m : nzs_m[NZEQ ? 0 : k] + row.nz_m;
                const bool ge_c = r_c >= cap_c[k], ge_m = r_m >= cap_m[k];
                const int la_c = ge_c ? 0 : la_term_f(r_c, rc100_c[k]);
                const int la_m = ge_m ? 0 : la_term_f(r_m, rc100_m[k]);
                // balanced_allocation.go:82-119; fraction >= 1  <=>  r >= cap (cap < 2^31)
                const double cf = div_by_rcp(r_c, cap_c[k], rc_c[k]);
                const double mf = div_by_rcp(r_m, cap_m[k], rc_m[k]);
                const int bs = (int)((1.0 - __builtin_fabs(cf - mf)) * 100.0);
                base[k] = ((la_c + la_m) >> 1) + ((ge_c || ge_m) ? 0 : bs);
            };
            if (!same) {
                static_for<0, SLOTS>([&](auto kc) { if (decltype(kc)::value < nslots) eval_slot(kc); });
                sig_rc = row.req_c; sig_rm = row.req_m; sig_nc = row.nz_c; sig_nm = row.nz_m; sig_cls = row.cls;
                cache_valid = true;
            } else if (dirty_wave == wave) {   // replica of the previous pod: one node changed
                static_for<0, SLOTS>([&](auto kc) { if (decltype(kc)::value == dirty_k) eval_slot(kc); });
            }
            dirty_wave = -1; dirty_k = -1;

            // -------- phase A: which node classes still have a feasible node ----------------
            unsigned bits = 0;
#pragma unroll
            for (int k = 0; k < SLOTS; ++k)
                if (k < nslots) bits |= ((feas >> k) & 1u) ? clsbit[k] : 0u;
            bits = wave_or_u32(bits);
            if (NW > 1) {
                if (lane == 0) mbA[bufA][wave] = bits;
                __syncthreads();
#pragma unroll
                for (int w = 0; w < NW; ++w) bits |= mbA[bufA][w];
                bits = __builtin_amdgcn_readfirstlane(bits);
                bufA ^= 1;
            }
            if (bits == 0u) {   // FitError: pod deleted, state (and cache) unchanged
                ++unsched;
                if (place && tid == 0) place[pid] = -1;
                continue;
            }
            // -------- SimonPlugin/GpuSharePlugin NormalizeScore row (x2: both plugins) -------
            const int* snrow;
            if (bits == bits_all) {
                snrow = s_sn2 + row.cls * Cn;
            } else {
                const int c = lane < Cn ? lane : 0;
                const int rawc = s_raw[row.cls * Cn + c];
                const bool inb = (lane < Cn) && ((bits >> c) & 1u);
                const int lo = wave_min_i32(inb ? rawc : 0x7fffffff);
                const int hi = wave_max_i32(inb ? rawc : (int)0x80000000);
                const int range = hi - lo;
                const double rr = range ? 1.0 / (double)range : 0.0;
                // (raw-lo)*100/range with the half-unit bias of la_term (simon_device.h); range < 2^31
                const int sn = range ? (int)__builtin_fma((double)(rawc - lo) * 100.0, rr, 0.5 * rr) : 0;
                if (lane < 32) s_tmp[wave][lane] = inb ? 2 * sn : 0;
                snrow = s_tmp[wave];
            }
            // -------- phase B: keys + arg-max ----------------------------------------------
            unsigned key = 0;
#pragma unroll
            for (int k = 0; k < SLOTS; ++k) {
                if (k < nslots) {
                    const int sn2 = *(const int*)((const char*)snrow + ncl4[k]);
                    const unsigned total = (unsigned)(base[k] + sn2);       // BA + LA + Simon + GpuShare
                    const unsigned kk = (total << 20) | (idxb - (unsigned)(k * T));
                    key = max(key, ((feas >> k) & 1u) ? kk : 0u);
                }
            }
            key = wave_max_u32(key);
            if (NW > 1) {
                if (lane == 0) mbB[bufB][wave] = key;
                __syncthreads();
#pragma unroll
                for (int w = 0; w < NW; ++w) key = max(key, mbB[bufB][w]);
                key = __builtin_amdgcn_readfirstlane(key);
                bufB ^= 1;
            }
            jstar = (int)(0xFFFFFu - (key & 0xFFFFFu));   // first maximum in canonical order
        }
        // -------- assume: NodeInfo.AddPod (V/framework/types.go:482-508) ---------------------
        const int owner = jstar % T, kstar = jstar / T;
        dirty_wave = owner / kWave;
        dirty_k = kstar;
        if (tid == owner) {
#pragma unroll
            for (int k = 0; k < SLOTS; ++k) {
                if (k == kstar) {
                    rq_c[k] += row.req_c;
                    rq_m[k] += row.req_m;
                    if (!NZEQ) { nzs_c[NZEQ ? 0 : k] += row.nz_c; nzs_m[NZEQ ? 0 : k] += row.nz_m; }
                    freep[k] -= 1;
                }
            }
            if (place) place[pid] = jstar;
        }
    }

    // ---- epilogue: sum of Requested over the scenario's nodes -------------------------------
    long long uc = 0, um = 0;
#pragma unroll
    for (int k = 0; k < SLOTS; ++k)
        if (k * T + tid < n) { uc += (long long)rq_c[k]; um += (long long)rq_m[k]; }
    uc = wave_sum_i64(uc);
    um = wave_sum_i64(um);
    if (NW > 1) {
        __syncthreads();
        if (lane == 0) { red[0][wave] = uc; red[1][wave] = um; }
        __syncthreads();
        uc = 0; um = 0;
        for (int w = 0; w < NW; ++w) { uc += red[0][w]; um += red[1][w]; }
    }
    if (tid == 0) {
        unscheduled[s] = unsched;
        used_cpu[s] = uc * (long long)sc.g_cpu;
        used_mem[s] = um * (long long)sc.g_mem;
    }
}

struct FastLaunch {
    const uint32_t *a_cpu, *a_mem; const int32_t *a_pods, *ncls; const uint32_t *i_rq_cpu, *i_rq_mem, *i_nz_cpu, *i_nz_mem;
    const int32_t* i_npods; const PodRowF* pods; const int32_t* orders; const ScenarioDesc* scen; const int32_t* perm;
    const uint64_t* static_mask; const int32_t* simon_raw; int32_t* unscheduled; int64_t *used_cpu, *used_mem;
    int32_t* placement; FastScalars sc;
};

template <int T, int SLOTS, bool M, bool Z>
static void launch_one(const FastLaunch& a, size_t lds, hipStream_t st) {
    hipLaunchKernelGGL((fast_kernel<T, SLOTS, M, Z>), dim3(a.sc.S), dim3(T), lds, st, a.a_cpu, a.a_mem, a.a_pods, a.ncls,
                       a.i_rq_cpu, a.i_rq_mem, a.i_nz_cpu, a.i_nz_mem, a.i_npods, a.pods, a.orders, a.scen, a.perm,
                       a.static_mask, a.simon_raw, a.unscheduled, a.used_cpu, a.used_mem, a.placement, a.sc);
}

template <int T, int SLOTS>
static hipError_t launch_ts(const FastLaunch& a, bool has_mask, bool nzeq, size_t lds, hipStream_t st) {
    if (has_mask) { if (nzeq) launch_one<T, SLOTS, true, true>(a, lds, st); else launch_one<T, SLOTS, true, false>(a, lds, st); }
    else { if (nzeq) launch_one<T, SLOTS, false, true>(a, lds, st); else launch_one<T, SLOTS, false, false>(a, lds, st); }
    return hipGetLastError();
}

template <int T>
static hipError_t launch_t(const FastLaunch& a, int slots, bool has_mask, bool nzeq, size_t lds, hipStream_t st) {
    switch (slots) {
        case 1: return launch_ts<T, 1>(a, has_mask, nzeq, lds, st);
        case 2: return launch_ts<T, 2>(a, has_mask, nzeq, lds, st);
        case 3: return launch_ts<T, 3>(a, has_mask, nzeq, lds, st);
        case 4: return launch_ts<T, 4>(a, has_mask, nzeq, lds, st);
        case 5: case 6: return launch_ts<T, 6>(a, has_mask, nzeq, lds, st);
        case 7: case 8: return launch_ts<T, 8>(a, has_mask, nzeq, lds, st);
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_fast(const FastLaunch& a, int T, int slots, bool has_mask, bool nzeq, size_t lds_bytes, hipStream_t st) {
    switch (T) {
        case 128: return launch_t<128>(a, slots, has_mask, nzeq, lds_bytes, st);
        case 256: return launch_t<256>(a, slots, has_mask, nzeq, lds_bytes, st);
        case 512: return launch_t<512>(a, slots, has_mask, nzeq, lds_bytes, st);
        case 1024: return launch_t<1024>(a, slots, has_mask, nzeq, lds_bytes, st);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace simon
