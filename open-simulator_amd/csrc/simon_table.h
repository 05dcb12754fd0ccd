// simon_table.h -- host/device interface of simon_table.hip (generation 4 of the cpu+memory scenario kernel:
// one wave per scenario over a (signature, node) byte table, class term folded into the block summaries).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "simon_cache.h"   // SigRow, ShapeRow, PodRowC, cache_nbp

namespace simon {

struct TableScalars {
    int32_t mask_words, Cn, Cp, P, S, K, n_shapes;
    int32_t ni_max;      // padded (class-major) scenario size bound of this launch (multiple of 16, <= 4096)
    uint64_t g_cpu, g_mem;
};

// pointers only the prologue, the epilogue and the rare paths (preset / pinned pods) use: one device-resident struct
struct TableCold {
    const int32_t *ncls, *rank, *shape_of, *cls_off, *clsprefix, *a_pods; const uint32_t *i_rq_cpu, *i_rq_mem, *i_nz_cpu, *i_nz_mem;
    const int32_t* i_npods; const SigRow* sigs; const ShapeRow* shapes; const ScenarioDesc* scen; const uint64_t* static_mask;
    const int32_t* simon_raw; int32_t* unscheduled; int64_t *used_cpu, *used_mem;
};

struct TableLaunch {
    const TableCold* cold;      // device pointer
    const int32_t* cls_list; const PodRowC* pods; const int32_t* orders; const int32_t* perm; int32_t* place_step;
    unsigned char* ws;   // HBM workspace [n_blocks][table_ws_bytes]: byte table + node state
    TableScalars sc;
};

constexpr int kTableMaxNodes = 4095;    // canonical index and padded position are 12-bit fields of the arg-max keys
constexpr int kTableMaxPadded = 4096;   // class-major padded positions of one scenario
constexpr int kTableMaxSigs = 128;      // two signatures per lane
constexpr int kTableMaxShapes = 256;    // u8 shape id per node
constexpr int kTableMaxClasses = 64;    // one lane per node class in the renormalisation; u8 class id per node

size_t table_lds_bytes(int K, int ni_max, int Cn, int Cp, int n_shapes, bool nzeq);
size_t table_ws_bytes(int K, int ni_max, int Cn, int Cp, int n_shapes, bool nzeq);
// launches n_blocks scenarios (one 64-thread workgroup each), scenario of block b = a.perm[b]
hipError_t launch_table(const TableLaunch& a, int n_blocks, bool has_mask, bool nzeq, bool has_pin, size_t lds_bytes, hipStream_t st);

}  // namespace simon
