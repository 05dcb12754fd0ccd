/* cabi_terms.c -- TEST: a plain-C consumer of the C-ABI that attaches EVERY optional array of simon_nodes_soa / simon_pods_soa /
 * simon_class_tables by name -- the topology-term tables (InterPodAffinity, PodTopologySpread, NodePorts), the static score tables and
 * Open-Local -- i.e. what integration/go/hipengine/flatten_terms.go fills and Flat.cNodes / cPods / cTables attach.  It stands in for
 * the cgo host while no Go toolchain exists: the same fields, the same calls (load, run_batch, explain, explain_local_detail).
 *
 * Fixture "SIMONFX3" (tests/golden/make_cabi_fixture.py, golden values from the oracle), little endian:
 *   int32 N P Cp Cn S n_orders n_scalar n_topo_keys n_terms n_node_sets n_local_specs n_fields
 *   n_fields x { char name[24]; int32 which (0 nodes, 1 pods, 2 tables); int32 pad; int64 nbytes; data padded to 8 bytes }
 *   int32 scen[S][2], orders[n_orders][P]; golden int32 unscheduled[S], placement[S][P]
 *   explain: int32 scenario n_failed has_detail; int32 failed[n_failed]; uint16 codes[n_failed][n] (padded to 8); int64 detail[n_failed][n][4] if has_detail
 *
 *   cc -O1 -I include tests/cabi/cabi_terms.c -L open-simulator_amd/csrc -lsimon_hip -o tests/cabi/cabi_terms && ./cabi_terms fixture.bin ... */
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "simon_hip.h"

typedef struct field { const char* name; int which; size_t off; } field;
#define ND(f) {#f, 0, offsetof(simon_nodes_soa, f)}
#define PD(f) {#f, 1, offsetof(simon_pods_soa, f)}
#define TB(f) {#f, 2, offsetof(simon_class_tables, f)}
static const field FIELDS[] = {
    ND(alloc_cpu), ND(alloc_mem), ND(alloc_eph), ND(alloc_pods), ND(init_req_cpu), ND(init_req_mem), ND(init_req_eph), ND(init_nz_cpu),
    ND(init_nz_mem), ND(init_npods), ND(node_class), ND(scalar_alloc), ND(init_scalar_req), ND(gpu_cnt), ND(gpu_mem_total), ND(init_gpu_used),
    ND(local_flags), ND(local_vg_cnt), ND(local_vg_cap), ND(init_vg_req), ND(local_vg_name), ND(local_dev_cnt), ND(local_dev_cap),
    ND(local_dev_media), ND(init_dev_alloc), ND(topo_dom), ND(topo_n_dom),
    PD(req_cpu), PD(req_mem), PD(req_eph), PD(nz_cpu), PD(nz_mem), PD(scalar_req), PD(pod_class), PD(preset_node), PD(gate_node), PD(gpu_mem),
    PD(gpu_cnt), PD(pin_node), PD(gpu_index),
    TB(static_mask), TB(static_reason), TB(simon_raw), TB(const_score), TB(node_affinity_raw), TB(taint_prefer_raw), TB(static_add),
    TB(term_topo_key), TB(term_node_set), TB(node_sets), TB(match_off), TB(match_idx), TB(anti_off), TB(anti_idx), TB(port_off), TB(port_idx),
    TB(aff_off), TB(aff_idx), TB(class_flags), TB(pref_off), TB(pref_idx), TB(pref_w), TB(own_off), TB(own_idx), TB(own_w), TB(spread_hard_off),
    TB(spread_hard_idx), TB(spread_hard_skew), TB(spread_hard_self), TB(spread_hard_set), TB(spread_soft_off), TB(spread_soft_idx),
    TB(spread_soft_skew), TB(local_spec_of), TB(local_specs), TB(topo_is_hostname), TB(spread_log),
};

static void* slurp(FILE* f, size_t bytes) {
    void* p = malloc(bytes ? bytes : 1);
    if (bytes && fread(p, 1, bytes, f) != bytes) { fprintf(stderr, "short fixture\n"); exit(1); }
    return p;
}

#define FAIL(...) do { fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); return 1; } while (0)

static int run(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) { perror(path); return 1; }
    char magic[8];
    int32_t h[12];
    if (fread(magic, 1, 8, f) != 8 || memcmp(magic, "SIMONFX3", 8) || fread(h, 4, 12, f) != 12) FAIL("%s: not a SIMONFX3 fixture", path);
    const int N = h[0], P = h[1], S = h[4], n_orders = h[5], n_fields = h[11];
    simon_nodes_soa nd; simon_pods_soa pd; simon_class_tables tb;
    memset(&nd, 0, sizeof nd); memset(&pd, 0, sizeof pd); memset(&tb, 0, sizeof tb);
    nd.struct_size = sizeof nd; nd.n_nodes = N; nd.n_scalar = h[6]; nd.n_topo_keys = h[7];
    pd.struct_size = sizeof pd; pd.n_pods = P;
    tb.struct_size = sizeof tb; tb.n_pod_classes = h[2]; tb.n_node_classes = h[3]; tb.n_terms = h[8]; tb.n_node_sets = h[9]; tb.n_local_specs = h[10];
    int attached = 0;
    for (int i = 0; i < n_fields; ++i) {
        char name[24]; int32_t which[2]; int64_t nbytes;
        if (fread(name, 1, 24, f) != 24 || fread(which, 4, 2, f) != 2 || fread(&nbytes, 8, 1, f) != 1) FAIL("%s: short field header", path);
        void* data = slurp(f, (size_t)((nbytes + 7) & ~7ll));
        const field* hit = NULL;
        for (size_t k = 0; k < sizeof FIELDS / sizeof FIELDS[0]; ++k)
            if (FIELDS[k].which == which[0] && !strncmp(FIELDS[k].name, name, 24)) hit = &FIELDS[k];
        if (!hit) FAIL("%s: the fixture carries field '%.24s' the header does not declare", path, name);
        char* base = which[0] == 0 ? (char*)&nd : which[0] == 1 ? (char*)&pd : (char*)&tb;
        memcpy(base + hit->off, &data, sizeof(void*));              /* every member in the table is a pointer */
        ++attached;
    }
    int32_t* scen = slurp(f, (size_t)S * 8);
    int32_t* orders = slurp(f, (size_t)n_orders * P * 4);
    int32_t* g_un = slurp(f, (size_t)S * 4);
    int32_t* g_pl = slurp(f, (size_t)S * P * 4);
    int32_t ex[3];
    if (fread(ex, 4, 3, f) != 3) FAIL("%s: short explain header", path);
    const int n_e = scen[2 * ex[0]], k = ex[1];
    int32_t* g_failed = slurp(f, (size_t)k * 4 + ((k & 1) ? 4 : 0));
    uint16_t* g_codes = slurp(f, (((size_t)k * n_e * 2) + 7) & ~(size_t)7);
    int64_t* g_detail = ex[2] ? slurp(f, (size_t)k * n_e * 32) : NULL;
    fclose(f);

    simon_ctx* c = simon_ctx_create(0);
    if (!c) FAIL("simon_ctx_create(0) failed");
#define TRY(call) do { int r_ = (call); if (r_ < 0) { fprintf(stderr, "%s: %s -> %d: %s\n", path, #call, r_, simon_last_error(c)); simon_ctx_destroy(c); return 1; } } while (0)
    TRY(simon_load_nodes(c, &nd)); TRY(simon_load_pods(c, &pd)); TRY(simon_load_class_tables(c, &tb));
    int32_t* un = malloc((size_t)S * 4); int64_t* uc = malloc((size_t)S * 8); int64_t* um = malloc((size_t)S * 8); int32_t* pl = malloc((size_t)S * P * 4);
    simon_batch_out out; memset(&out, 0, sizeof out);
    out.struct_size = sizeof out; out.unscheduled = un; out.used_cpu = uc; out.used_mem = um; out.placement = pl;
    TRY(simon_run_batch(c, (const simon_scenario*)scen, S, orders, n_orders, &out));
    for (int s = 0; s < S; ++s) {
        if (un[s] != g_un[s]) FAIL("%s: scenario %d unscheduled %d, fixture %d", path, s, un[s], g_un[s]);
        for (int p = 0; p < P; ++p)
            if (pl[(size_t)s * P + p] != g_pl[(size_t)s * P + p]) FAIL("%s: scenario %d pod %d on node %d, fixture %d", path, s, p, pl[(size_t)s * P + p], g_pl[(size_t)s * P + p]);
    }
    if (k > 0) {      /* FitError inputs incl. what Open-Local's reasons carry (simon_explain_local_detail, ABI v5) */
        int32_t* failed = malloc((size_t)k * 4); uint16_t* codes = malloc((size_t)k * n_e * 2); int64_t* detail = calloc((size_t)k * n_e * 4, 8);
        simon_scenario sc = {n_e, scen[2 * ex[0] + 1]};
        int nf = simon_explain(c, sc, orders + (size_t)sc.order_id * P, failed, codes, k);
        if (nf < k || memcmp(failed, g_failed, (size_t)k * 4) || memcmp(codes, g_codes, (size_t)k * n_e * 2)) FAIL("%s: explain differs from the fixture (n_failed %d)", path, nf);
        int rows = simon_explain_local_detail(c, detail, k);
        if (rows < 0) FAIL("%s: simon_explain_local_detail -> %d: %s", path, rows, simon_last_error(c));
        if (g_detail && (rows != k || memcmp(detail, g_detail, (size_t)k * n_e * 32))) FAIL("%s: Open-Local error sizes differ from the fixture (%d rows)", path, rows);
        if (!g_detail && rows != 0) FAIL("%s: %d detail rows on a problem without local storage", path, rows);
        free(failed); free(codes); free(detail);
    }
    simon_stats st;
    TRY(simon_get_stats(c, &st));
    printf("%s: %d fields attached, %d scenarios x %d pods ok (kernel variant %d generation %d), %d failed pods explained%s\n", path, attached, S, P,
           st.kernel_variant, st.kernel_generation, k, g_detail ? " with Open-Local sizes" : "");
    simon_ctx_destroy(c);
    return 0;
#undef TRY
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s fixture.bin [...]\n", argv[0]); return 1; }
    if (simon_hip_version() != SIMON_HIP_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 1; }
    if (simon_hip_device_count() < 1) { printf("cabi_terms: library loaded (ABI %d), no GPU visible\n", simon_hip_version()); return 77; }
    for (int i = 1; i < argc; ++i)
        if (run(argv[i])) return 1;
    printf("cabi_terms ok\n");
    return 0;
}
