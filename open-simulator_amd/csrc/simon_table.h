// simon_table.h -- host/device interface of simon_table.hip (generation 4 of the cpu+memory scenario kernel:
// one wave per scenario, pre-keyed (signature, node) score table in canonical node order).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "simon_cache.h"   // SigRow, ShapeRow, PodRowC, cache_nbp

namespace simon {

struct TableScalars {
    int32_t mask_words, Cn, Cp, P, S, K, n_shapes;
    int32_t ni_max;      // padded scenario size bound of this launch (multiple of 16, <= 4096)
    uint64_t g_cpu, g_mem;
};

struct TableLaunch {
    const int32_t *ncls, *shape_of, *a_pods; const uint32_t *i_rq_cpu, *i_rq_mem, *i_nz_cpu, *i_nz_mem;
    const int32_t* i_npods; const SigRow* sigs; const ShapeRow* shapes; const PodRowC* pods;
    const int32_t* orders; const ScenarioDesc* scen; const int32_t* perm; const uint64_t* static_mask;
    const int32_t* simon_raw; int32_t* unscheduled; int64_t *used_cpu, *used_mem; int32_t* place_step;
    unsigned char* ws;   // HBM workspace [n_blocks][table_ws_bytes]: pre-keyed table + node state
    TableScalars sc;
};

constexpr int kTableMaxNodes = 4095;    // position (= canonical index) is a 12-bit field of the arg-max key
constexpr int kTableMaxSigs = 128;      // two signatures per lane
constexpr int kTableMaxShapes = 256;    // u8 shape id per node
constexpr int kTableMaxClasses = 64;    // one lane per node class in the renormalisation; u8 class id per node

size_t table_lds_bytes(int K, int ni_max, int Cn, int Cp, int n_shapes, bool nzeq);
size_t table_ws_bytes(int K, int ni_max, int Cn, int Cp, int n_shapes, bool nzeq);
// launches n_blocks scenarios (one 64-thread workgroup each), scenario of block b = a.perm[b]
hipError_t launch_table(const TableLaunch& a, int n_blocks, bool has_mask, bool nzeq, bool has_pin, size_t lds_bytes, hipStream_t st);

}  // namespace simon
