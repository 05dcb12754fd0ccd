// simon_cache.h -- host/device interface of simon_cache.hip (the LDS score-table scenario kernel).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "simon_device.h"

namespace simon {

struct SigRow {      // one pod request signature (48 B)
    double req_c, req_m;   // computePodResourceRequest (fit.go:148-165), gcd-normalised, exact
    double nz_c, nz_m;     // non-zero request (V/framework/types.go:601-636)
    int32_t cls;           // pod class: row of static_mask / simon_raw
    uint32_t flags;        // bit0: all-zero request (fit.go:244-249)
    int32_t pad[2];
};
static_assert(sizeof(SigRow) == 48, "SigRow must be 48 bytes");

struct ShapeRow {    // one distinct (Allocatable.MilliCPU, Allocatable.Memory) pair (48 B)
    double cap_c, cap_m;
    double rc_c, rc_m;         // RN(1/cap)
    double rc100_c, rc100_m;   // RN(100 * rc)
};
static_assert(sizeof(ShapeRow) == 48, "ShapeRow must be 48 bytes");

struct PodRowC { int32_t sig, preset, gate, cls; };   // 16 B, one s_load_dwordx4

struct CacheScalars {
    int32_t mask_words, Cn, Cp, P, S, K, n_shapes;
    int32_t ni_max;      // padded scenario size bound of this launch (multiple of 16, <= 2032)
    int32_t ablate;      // timing experiments only (env SIMON_CACHE_ABLATE): results are wrong when non-zero
    uint64_t g_cpu, g_mem;
};

struct CacheLaunch {
    const int32_t *ncls, *rank, *shape_of, *a_pods; const uint32_t *i_rq_cpu, *i_rq_mem, *i_nz_cpu, *i_nz_mem;
    const int32_t *i_npods, *clsprefix; const SigRow* sigs; const ShapeRow* shapes; const PodRowC* pods;
    const int32_t* orders; const ScenarioDesc* scen; const int32_t* perm; const uint64_t* static_mask;
    const int32_t* simon_raw; int32_t* unscheduled; int64_t *used_cpu, *used_mem; int32_t* place_step;
    unsigned char* ws;   // HBM workspace [n_blocks][cache_ws_bytes]: tiles + node state
    CacheScalars sc;
};

constexpr int kCacheMaxNodes = 2047;    // canonical index and padded position: 11 bits each in the key
constexpr int kCacheMaxPadded = 2032;
constexpr int kCacheMaxSigs = 64;       // one lane per signature
constexpr int kCacheMaxShapes = 256;
constexpr size_t kLdsPerCU = 160 * 1024;

// summary row pitch in u16 entries: >= nblk and == 2 (mod 4), so the K column entries of one block
// (pitch * 2 bytes apart) fall into distinct LDS banks
__host__ __device__ inline int cache_nbp(int nblk) { return nblk + ((2 - (nblk & 3)) & 3); }
size_t cache_lds_bytes(int K, int ni_max, int Cn, int Cp, int n_shapes, bool nzeq);
size_t cache_ws_bytes(int K, int ni_max, int Cn, int Cp, int n_shapes, bool nzeq);
// launches n_blocks scenarios (one 64-thread workgroup each), scenario of block b = a.perm[b]
hipError_t launch_cache(const CacheLaunch& a, int n_blocks, bool has_mask, bool nzeq, bool has_pin, size_t lds_bytes, hipStream_t st);
hipError_t launch_unpermute(const int32_t* place_step, const int32_t* inv_orders, const ScenarioDesc* scen, int S, int P,
                            int32_t* placement, hipStream_t st);

}  // namespace simon
