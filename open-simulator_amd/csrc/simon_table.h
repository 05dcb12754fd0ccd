// simon_table.h -- host/device interface of simon_table.hip (generations 4 to 6 of the cpu+memory scenario kernel:
// one wave per scenario over a (signature, node) byte table, class term folded into the block summaries; 5: two-level
// summary; 6: + Open-Gpu-Share and node-level anti-affinity as per-block position masks).
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
    int32_t pad[2];        // GPU fold (TableScalars::static_tables & 128): gpu-mem per device in gcd units, devices requested (0: not a GPU signature; -1: a GPU request for no device)
};
static_assert(sizeof(SigRow) == 48, "SigRow must be 48 bytes");

struct ShapeRow {    // capacity of one internal node class = one distinct (Allocatable.MilliCPU, Allocatable.Memory) pair (48 B)
    double cap_c, cap_m;
    double rc_c, rc_m;         // RN(1/cap)
    double rc100_c, rc100_m;   // RN(100 * rc)
};
static_assert(sizeof(ShapeRow) == 48, "ShapeRow must be 48 bytes");

// 16 B: signature | table class << 10, preset (>= 0) or pinned (<= -2: -2 - node) node, gate, and (REST) GPU request + 1 | extra-resource
// request + 1 << 6 | entries << 12 | offset << 18 of the pod's entries in TableCold::xrows (0 = the score table alone decides the pod)
struct PodRowC { int32_t sigcls, preset, gate, rest; };

struct TableScalars {
    int32_t mask_words, Cn, Cp, P, S, K;
    int32_t rk_stride;   // 0: cls_list = the pool's per-class node lists; N: per-scenario lists in rank order (simon_set_node_ranks)
    int32_t static_tables;   // bit 0 / 1 / 2: TableCold::na_raw / tt_raw / add_raw present; bit 3: record TableCold::gpu_slices; bit 4: signature k + 64 is a twin of k (same request); bit 5: TableCold::foldx present; bit 6: SPREAD && AFF instantiations; bit 7: Open-Gpu-Share folded into the table (SigRow::pad = GPU request, devices per position behind the workspace)
    int32_t NZ;          // REST: topology keys that are NOT node-level (a term on one marks every position of the pod's domain)
    int32_t M, G, X;     // REST: rows of the per-block position masks (G GPU requests + X extra-resource requests + 2 x terms)
    int32_t TH, TZ, NZK; // SPREAD: hostname-key term rows, zone-key term rows, zone-like topology keys (class split)
    int32_t ni_max;      // padded (class-major) scenario size bound of this launch (multiple of 16, <= 4096; coarse: of 64, <= 8192)
    uint64_t g_cpu, g_mem;
};

// pointers only the prologue, the epilogue and the rare paths (preset / pinned pods) use: one device-resident struct
struct TableCold {
    const int32_t *ncls, *rank, *cls_off, *clsprefix, *a_pods; const uint32_t *i_rq_cpu, *i_rq_mem, *i_nz_cpu, *i_nz_mem;
    const int32_t* i_npods; const SigRow* sigs; const ShapeRow* shapes; const ScenarioDesc* scen; const uint64_t* static_mask;
    const int32_t* simon_raw; int32_t* unscheduled; int64_t *used_cpu, *used_mem;
    // static score tables by (table class, internal node class), null when the plugin scores every node alike:
    // NodeAffinity preferred terms, TaintToleration PreferNoSchedule, already weighted additions (include/simon_hip.h, ABI v2)
    const int32_t *na_raw, *tt_raw, *add_raw;
    unsigned long long* prof;   // [S][12] phase ticks (builds with -DSIMON_TABLE_PROFILE and env SIMON_TABLE_PROF), else null
    // per-scenario node order (simon_set_node_ranks), [S][N] each: a node's index inside its class in rank order, the rank itself
    // (canonical index of ties); the per-class node lists in rank order (class segments at cls_off, first clsprefix[n][d]
    // entries valid) replace TableLaunch::cls_list
    const int32_t *rk_pos, *rk_rank;    // (the lists themselves travel as TableLaunch::cls_list)
    int32_t N;
    // REST (Open-Gpu-Share + required anti-affinity on node-level topology keys): mask rows per term class; GPU signatures (gpu-mem per device in gcd units, device count); the pool's devices
    const int32_t* xrows;           // per term class, <= 63 entries: mask row that must be clear | mask row the pod sets << 16 | (zone key + 1) << 28
    const uint2* gsig;              // [G]
    // extra resources = ephemeral storage (gcd units) + SIMON_MAX_SCALAR extended resources: requests [X][8], the pool's
    // allocatable and Requested at the start [N][8] (component 0 = ephemeral storage, 1.. = extended resources)
    const uint32_t *xsig, *xalloc, *i_xused;
    const int32_t* zdom;            // [NZ][N] domain of a node under a zone-like key (-1: no label)
    unsigned long long* gpu_slices; // [S][P] by pod id: devices Reserve booked (simon_batch_out.gpu_slices), written when TableScalars::static_tables & 8
    // SPREAD (soft PodTopologySpread constraints, generation 7): per pod class [soft constraints..., counted terms...] in sp_ent
    // (soft: term slot | maxSkew << 16 | SIMON_SPREAD_DUP_KEY bit 30; counted: term slot | multiplicity << 16), each with its term's
    // row: kind (1 hostname-like row, 2 zone-like row) | row << 2 | zone key slot << 16 | (node set + 1) << 19; Go's math.Log table;
    // node sets; zone domain of a class
    const uint32_t* foldx;          // [K][ceil(K / 32)] (TableScalars::static_tables & 32): bit S of row L = a pod of signature L on a node excludes signature S from it
    const int2* sp_ent;             // x = the entry, y = its term's row (kind | row << 2 | zone key slot << 16 | (node set + 1) << 19)
    const double* spread_log;       // [N + 1] math.Log(float64(i + 2)) (simon_class_tables.spread_log)
    const uint64_t* node_sets;      // [R][set_words]
    int32_t set_words;
    const signed char* cls_zdom;    // [NZK][Cn] domain of an internal node class under zone key slot z (-1: the label is missing)
    const int32_t* gpu_cnt;         // [N]
    const uint32_t *gpu_devtot, *i_gused;   // [N] per-device total, [N][8] used at the start (gcd units)
    // REST && SPREAD in one instantiation (round 6): PodRowC::rest keeps the REST descriptor (per pod: its GPU request is part of it), the
    // SPREAD descriptor travels here, by pod id
    const int32_t* sp_word;         // [P]
};

struct TableLaunch {
    const TableCold* cold;      // device pointer
    const int32_t* cls_list; const PodRowC* pods; const int32_t* orders; const int32_t* perm; int32_t* place_step;
    const unsigned long long* ws_off;   // [n_blocks] byte offset of a workgroup's slice of ws (table_ws_bytes of its own scenario)
    unsigned char* ws;   // HBM workspace: byte table + node state (+ per-16 summary entries and counters when coarse) of every scenario
    int team;            // waves per scenario: 0 / 1 = one (the throughput shape); kTeamWaves = team mode (SPREAD only; small batches)
    bool spread;         // some pod class carries soft spread constraints (generation 7; implies coarse; together with rest: simon_table_rs.hip)
    bool aff;            // some pod class carries required-affinity entries (REST)
    bool rest;           // some pods need the per-node filters of the REST path (implies coarse)
    bool coarse;         // two-level summary: LDS entries cover 64 positions, per-16 entries live in the workspace (tcarve)
    bool lds_x;          // generation 6 with its mask rows, row totals and the canonical index of every position in LDS (table_kernel: LDSX; batches resident at once)
    bool lds_ws;         // generation 4 with the scenario's workspace in LDS (table_kernel: LDSWS; small batches of small problems): the dynamic LDS of the launch = summaries + the largest workspace
    TableScalars sc;
};

constexpr int kTableMaxNodes = 4095;    // canonical index and padded position are 12-bit fields of the arg-max keys
constexpr int kTableMaxNodesCoarse = 8191;    // ... 13-bit fields with the two-level summary
constexpr int kTableMaxPadded = 4096;   // class-major padded positions of one scenario (classes padded to 16)
constexpr int kTableMaxPaddedCoarse = 8192;   // ... with the two-level summary (classes padded to 64)
constexpr int kTableMaxSigs = 1023;     // two signatures per lane in registers + further groups of 128 read from TableCold::sigs per cycle (two with group 0's
                                        // round trip, the rest one each); 10 bits of the pod row name the signature
constexpr int kTableFastSigs = 384;     // ... the signatures whose rows travel in ONE round trip per cycle
constexpr size_t kTableLdsPerCU = 160 * 1024;
constexpr int kTableMaxGpuSigs = 32;    // distinct (gpu-mem, gpu-count) requests: one mask row and one lane each
constexpr int kTableMaxXres = 32;       // distinct (ephemeral-storage, extended-resource) requests: one mask row and one lane each
constexpr int kTableMaxTerms = 120;     // node-level anti-affinity terms: two mask rows each
constexpr int kSpreadMaxZoneDom = 16;    // domains of a zone-like key of a soft spread constraint (one u32 counter each)
constexpr int kSpreadMaxHostTerms = 4096, kSpreadMaxZoneTerms = 1024, kSpreadMaxZoneKeys = 3;
#ifndef SIMON_SPREAD_TAB_MAX
#define SIMON_SPREAD_TAB_MAX 512          // (tests build a library with a tiny table to drive every pod through the general walk)
#endif
constexpr int kSpreadTabMax = SIMON_SPREAD_TAB_MAX;       // entries of spread_select's per-pod score table in LDS: classes x (largest counter + 1, a power of two)
#ifndef SIMON_SPREAD_TAB_MAX2
#define SIMON_SPREAD_TAB_MAX2 2048
#endif
constexpr int kSpreadTabMax2 = SIMON_SPREAD_TAB_MAX2;    // ... of the instantiations for 65 .. 128 node classes (table_kernel: CN2)
constexpr int kTeamWaves = 4;           // team mode (table_kernel: NW): waves per scenario, one per SIMD of the CU.  8 and 16 were built and measured
                                        // SLOWER on every batch (profiles/r04/r04b_*: once the walks are a quarter, the leader's chain and the barriers decide)
constexpr int kTeamWavesMax = 16;       // (the exchange slots are sized for it: a wider team is one more translation unit, simon_table_team<N>.hip)
constexpr int kTableMaxClasses = 64;    // internal node classes = distinct (column content, allocatable) pairs: one lane each in the REST select and the SPREAD walks
constexpr int kTableMaxClassesSpread = 128;  // ... two per lane in the REST select and in the SPREAD walks that score soft constraints only (CN2, round 6)
constexpr int kTableMaxClassesPlain = 256;   // ... up to four per lane where only the prologue and the class terms' re-base are lane-shaped (no REST rows, no SPREAD; 129 .. 256: end of round 6)

size_t table_lds_bytes(int K, int ni_max, int Cn, bool coarse, bool rest, int nzk = -1);   // LDS per workgroup for padded scenario sizes up to ni_max (nzk >= 0: SPREAD; | 0x100: second score table; | 0x200: team mode; | 0x400: CN2's larger score table)
size_t table_ws_bytes(int K, int ni, bool nzeq, bool coarse, int Cn, int M, int NZ, int TH = 0, int TZ = 0);  // HBM workspace of ONE scenario with ni padded positions
// launches n_blocks scenarios (one 64-thread workgroup each), scenario of block b = a.perm[b]
hipError_t launch_table(const TableLaunch& a, int n_blocks, bool has_mask, bool nzeq, bool has_pin, size_t lds_bytes, hipStream_t st);
// the same for a.team == kTeamWaves (64 * team threads per scenario; SPREAD problems only): simon_table_team4.hip
hipError_t launch_table_team4(const TableLaunch& a, int n_blocks, bool has_mask, bool nzeq, size_t lds_bytes, hipStream_t st);
// generation 4 with the workspace in LDS (simon_table_lds.hip; lds_bytes = table_lds_bytes(...) rounded up to 128 + the largest table_ws_bytes of the batch)
hipError_t launch_table_lds(const TableLaunch& a, int n_blocks, bool nzeq, size_t lds_bytes, hipStream_t st);
// generation 6 with the mask rows in LDS (simon_table_restlds.hip; lds_bytes = table_lds_bytes(...) rounded up to 128 + table_ldsx_bytes(...))
hipError_t launch_table_rest_lds(const TableLaunch& a, int n_blocks, bool nzeq, size_t lds_bytes, hipStream_t st);
inline size_t table_ldsx_bytes(int ni_max, int M) {   // rows [M][ni_max / 16] u16, row totals [M] u32, canonical indices [ni_max] u16 (each rounded up to 128 bytes)
    return ((((size_t)(ni_max >> 4) * M * 2) + 127) & ~(size_t)127) + (((size_t)M * 4 + 127) & ~(size_t)127) + (((size_t)ni_max * 2 + 127) & ~(size_t)127);
}
// generation 6, one wave per scenario (simon_table_rest.hip: the REST instantiations, a translation unit of their own since round 5)
hipError_t launch_table_rest(const TableLaunch& a, int n_blocks, bool nzeq, size_t lds_bytes, hipStream_t st);
hipError_t launch_table_cls4(const TableLaunch& a, int n_blocks, bool nzeq, size_t lds_bytes, hipStream_t st);    // generation 4 for 129 .. 256 node classes (simon_table_cls4.hip)
hipError_t launch_table_rest2(const TableLaunch& a, int n_blocks, bool nzeq, size_t lds_bytes, hipStream_t st);   // ... for 65 .. 128 node classes (simon_table_rest2.hip)
// generation 7, one wave per scenario (simon_table_spread.hip: the SPREAD instantiations, a translation unit of their own)
hipError_t launch_table_spread(const TableLaunch& a, int n_blocks, bool has_mask, bool nzeq, size_t lds_bytes, hipStream_t st);
// generation 7 over the position-mask rows of generation 6 (REST && SPREAD: simon_table_rs.hip; <= 64 node classes)
hipError_t launch_table_rs(const TableLaunch& a, int n_blocks, bool nzeq, size_t lds_bytes, hipStream_t st);
hipError_t launch_table_spread2(const TableLaunch& a, int n_blocks, bool nzeq, size_t lds_bytes, hipStream_t st);   // ... for 65 .. 128 node classes (simon_table_spread2.hip)
// placement[s][pod] = place_step[s][inverse order of s][pod]: the kernel records placements by scheduling STEP (coalesced)
hipError_t launch_unpermute(const int32_t* place_step, const int32_t* inv_orders, const ScenarioDesc* scen, int S, int P,
                            int32_t* placement, hipStream_t st);

}  // namespace simon
