/*
 * cabi_smoke.c -- a plain-C consumer of libsimon_hip.so (include/simon_hip.h), built with gcc:
 *
 *     gcc -O1 -std=c11 -I include tests/cabi/cabi_smoke.c -L open-simulator_amd/csrc -lsimon_hip \
 *         -Wl,-rpath,$PWD/open-simulator_amd/csrc -lpthread -o tests/cabi/cabi_smoke
 *     tests/cabi/cabi_smoke tests/golden/cabi_kav.bin tests/golden/cabi_config2_sweep.bin
 *
 * It stands in for the cgo host (pkg/simulator/core.go:67 Simulate, pkg/apply/apply.go:203-259 the add-nodes loop)
 * while no Go toolchain exists: everything a Go caller would do goes through the same entry points, in the same
 * order, with caller-owned buffers -- simon_load_* -> simon_run_batch -> simon_min_plan -> simon_explain /
 * simon_explain_loaded, the device group (two members on device 0), and two contexts on two pthreads.
 * Expected values come from the committed fixtures (tests/golden/make_cabi_fixture.py wrote them from the oracle).
 * A "SIMONFX2" fixture additionally carries every optional array integration/go/hipengine/engine.go fills (ephemeral storage,
 * extended resources, Open-Gpu-Share nodes / pods incl. arriving gpu-index lists, preset / gate / pin nodes, static masks and
 * reasons, const scores) and the golden simon_batch_out.gpu_slices: the struct usage of the Go shim, executed from C.
 *
 * Exit codes: 0 ok, 77 no usable GPU (the binary linked and ran: what the CPU-only test checks), 1 mismatch / error.
 */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "simon_hip.h"

typedef struct fixture {
    int32_t N, P, Cp, Cn, S, n_orders;
    int64_t *alloc_cpu, *alloc_mem, *req_cpu, *req_mem, *simon_raw, *g_used_cpu, *g_used_mem;
    int32_t *alloc_pods, *node_class, *pod_class, *scen, *orders, *g_unsched, *g_place;
    int32_t plan_found, plan_scenario, plan_n_nodes;
    int32_t ex_scenario, ex_n_failed, *ex_failed;
    uint16_t* ex_codes;
    /* SIMONFX2 extension (all NULL / 0 for SIMONFX1) */
    int32_t K, has_gpu, has_mask;
    int64_t *alloc_eph, *req_eph, *scalar_alloc, *scalar_req, *gpu_mem_total, *pod_gpu_mem, *const_score;
    int32_t *gpu_cnt, *pod_gpu_cnt, *preset, *gate, *pin;
    uint32_t* gpu_index;
    uint64_t *static_mask, *g_slices;
    uint8_t* static_reason;
} fixture;

static void* slurp(FILE* f, size_t bytes) {
    void* p = malloc(bytes ? bytes : 1);
    if (!p || fread(p, 1, bytes, f) != bytes) { fprintf(stderr, "fixture truncated\n"); exit(1); }
    return p;
}

static void load_fixture(const char* path, fixture* x) {
    FILE* f = fopen(path, "rb");
    if (!f) { perror(path); exit(1); }
    char magic[8];
    memset(x, 0, sizeof *x);
    if (fread(magic, 1, 8, f) != 8 || (memcmp(magic, "SIMONFX1", 8) && memcmp(magic, "SIMONFX2", 8))) { fprintf(stderr, "%s: bad magic\n", path); exit(1); }
    const int ext = magic[7] == '2';
    int32_t hdr[6];
    if (fread(hdr, 4, 6, f) != 6) exit(1);
    x->N = hdr[0]; x->P = hdr[1]; x->Cp = hdr[2]; x->Cn = hdr[3]; x->S = hdr[4]; x->n_orders = hdr[5];
    const size_t N = x->N, P = x->P, S = x->S;
    x->alloc_cpu = slurp(f, N * 8); x->alloc_mem = slurp(f, N * 8); x->alloc_pods = slurp(f, N * 4); x->node_class = slurp(f, N * 4);
    x->req_cpu = slurp(f, P * 8); x->req_mem = slurp(f, P * 8); x->pod_class = slurp(f, P * 4);
    x->simon_raw = slurp(f, (size_t)x->Cp * x->Cn * 8);
    x->scen = slurp(f, S * 8); x->orders = slurp(f, (size_t)x->n_orders * P * 4);
    x->g_unsched = slurp(f, S * 4); x->g_used_cpu = slurp(f, S * 8); x->g_used_mem = slurp(f, S * 8); x->g_place = slurp(f, S * P * 4);
    int32_t t[3];
    if (fread(t, 4, 3, f) != 3) exit(1);
    x->plan_found = t[0]; x->plan_scenario = t[1]; x->plan_n_nodes = t[2];
    if (fread(t, 4, 2, f) != 2) exit(1);
    x->ex_scenario = t[0]; x->ex_n_failed = t[1];
    x->ex_failed = slurp(f, (size_t)x->ex_n_failed * 4);
    x->ex_codes = slurp(f, (size_t)x->ex_n_failed * (size_t)x->scen[2 * x->ex_scenario] * 2);
    if (ext) {
        int32_t h[3];
        if (fread(h, 4, 3, f) != 3) exit(1);
        x->K = h[0]; x->has_gpu = h[1]; x->has_mask = h[2];
        const size_t K = x->K, words = (N + 63) / 64, Cp = x->Cp;
        x->alloc_eph = slurp(f, N * 8); x->req_eph = slurp(f, P * 8);
        x->scalar_alloc = slurp(f, K * N * 8); x->scalar_req = slurp(f, K * P * 8);
        x->preset = slurp(f, P * 4); x->gate = slurp(f, P * 4); x->pin = slurp(f, P * 4);
        x->const_score = slurp(f, Cp * 8);
        if (x->has_gpu) {
            x->gpu_cnt = slurp(f, N * 4); x->gpu_mem_total = slurp(f, N * 8);
            x->pod_gpu_mem = slurp(f, P * 8); x->pod_gpu_cnt = slurp(f, P * 4); x->gpu_index = slurp(f, P * 4);
            x->g_slices = slurp(f, S * P * 8);
        }
        if (x->has_mask) { x->static_mask = slurp(f, Cp * words * 8); x->static_reason = slurp(f, Cp * N); }
    }
    fclose(f);
}

#define CHECK(cond, ...) do { if (!(cond)) { fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); return 1; } } while (0)

static void fill_inputs(const fixture* x, simon_nodes_soa* nd, simon_pods_soa* pd, simon_class_tables* tb) {
    memset(nd, 0, sizeof *nd); memset(pd, 0, sizeof *pd); memset(tb, 0, sizeof *tb);
    nd->struct_size = sizeof *nd; nd->n_nodes = x->N;
    nd->alloc_cpu = x->alloc_cpu; nd->alloc_mem = x->alloc_mem; nd->alloc_pods = x->alloc_pods; nd->node_class = x->node_class;
    pd->struct_size = sizeof *pd; pd->n_pods = x->P;
    pd->req_cpu = x->req_cpu; pd->req_mem = x->req_mem; pd->pod_class = x->pod_class;
    tb->struct_size = sizeof *tb; tb->n_pod_classes = x->Cp; tb->n_node_classes = x->Cn; tb->simon_raw = x->simon_raw;
    /* the optional arrays of a SIMONFX2 fixture: what Flat.cNodes / cPods / cTables of the Go shim attach */
    nd->alloc_eph = x->alloc_eph; pd->req_eph = x->req_eph;
    nd->n_scalar = x->K;
    if (x->K > 0) { nd->scalar_alloc = x->scalar_alloc; pd->scalar_req = x->scalar_req; }
    pd->preset_node = x->preset; pd->gate_node = x->gate; pd->pin_node = x->pin;
    tb->const_score = x->const_score;
    if (x->has_gpu) {
        nd->gpu_cnt = x->gpu_cnt; nd->gpu_mem_total = x->gpu_mem_total;
        pd->gpu_mem = x->pod_gpu_mem; pd->gpu_cnt = x->pod_gpu_cnt; pd->gpu_index = x->gpu_index;
    }
    if (x->has_mask) { tb->static_mask = x->static_mask; tb->static_reason = x->static_reason; }
}

static int compare_batch(const fixture* x, const int32_t* un, const int64_t* uc, const int64_t* um, const int32_t* pl, const char* who) {
    for (int s = 0; s < x->S; ++s) {
        CHECK(un[s] == x->g_unsched[s], "%s: scenario %d unscheduled %d, fixture %d", who, s, un[s], x->g_unsched[s]);
        CHECK(uc[s] == x->g_used_cpu[s] && um[s] == x->g_used_mem[s], "%s: scenario %d used cpu/mem differ", who, s);
        for (int p = 0; p < x->P; ++p)
            CHECK(pl[(size_t)s * x->P + p] == x->g_place[(size_t)s * x->P + p], "%s: scenario %d pod %d on node %d, fixture %d", who, s, p,
                  pl[(size_t)s * x->P + p], x->g_place[(size_t)s * x->P + p]);
    }
    return 0;
}

/* one Simulate()-batch through one context, as the cgo shim would drive it */
static int run_single(const fixture* x, int device, int rounds, const char* who) {
    simon_ctx* c = simon_ctx_create(device);
    CHECK(c, "%s: simon_ctx_create(%d) failed", who, device);
    simon_nodes_soa nd; simon_pods_soa pd; simon_class_tables tb;
    fill_inputs(x, &nd, &pd, &tb);
    int rc = 0;
#define TRY(call) do { int r_ = (call); if (r_ < 0) { fprintf(stderr, "%s: %s -> %d: %s\n", who, #call, r_, simon_last_error(c)); rc = 1; goto done; } } while (0)
    const size_t S = x->S, P = x->P;
    int32_t* un = malloc(S * 4); int64_t* uc = malloc(S * 8); int64_t* um = malloc(S * 8); int32_t* pl = malloc(S * P * 4);
    uint64_t* gs = x->g_slices ? malloc(S * P * 8) : NULL;
    TRY(simon_load_nodes(c, &nd)); TRY(simon_load_pods(c, &pd)); TRY(simon_load_class_tables(c, &tb));
    for (int r = 0; r < rounds && !rc; ++r) {
        simon_batch_out out; memset(&out, 0, sizeof out);
        out.struct_size = sizeof out; out.unscheduled = un; out.used_cpu = uc; out.used_mem = um; out.placement = pl;
        out.gpu_slices = gs;                                  /* non-NULL: the run records the devices Reserve books (ABI v4) */
        memset(pl, 0x7f, S * P * 4);
        if (gs) memset(gs, 0x7f, S * P * 8);
        TRY(simon_run_batch(c, (const simon_scenario*)x->scen, x->S, x->orders, x->n_orders, &out));
        rc = compare_batch(x, un, uc, um, pl, who);
        if (rc) break;
        if (gs) {                                             /* the whole matrix, and one row through its own entry point */
            if (memcmp(gs, x->g_slices, S * P * 8)) { fprintf(stderr, "%s: gpu_slices differ from the fixture\n", who); rc = 1; break; }
            TRY(simon_fetch_gpu_slices(c, 0, gs));
            if (memcmp(gs, x->g_slices, P * 8)) { fprintf(stderr, "%s: fetch_gpu_slices row differs\n", who); rc = 1; break; }
        }
        simon_plan plan;
        TRY(simon_min_plan(c, 100, 100, &plan));
        if (plan.found != x->plan_found || (plan.found && (plan.scenario != x->plan_scenario || plan.n_nodes != x->plan_n_nodes))) {
            fprintf(stderr, "%s: plan {%d, s=%d, n=%d}, fixture {%d, s=%d, n=%d}\n", who, plan.found, plan.scenario, plan.n_nodes,
                    x->plan_found, x->plan_scenario, x->plan_n_nodes);
            rc = 1; break;
        }
        /* one row through simon_fetch_placement */
        TRY(simon_fetch_placement(c, x->S - 1, pl));
        if (memcmp(pl, x->g_place + (S - 1) * P, P * 4)) { fprintf(stderr, "%s: fetch_placement row differs\n", who); rc = 1; break; }
    }
    if (!rc) {   /* FitError inputs: per-node failure codes of the pods one scenario leaves out, both entry points */
        const int n_e = x->scen[2 * x->ex_scenario], k = x->ex_n_failed;
        int32_t* failed = malloc((size_t)k * 4); uint16_t* codes = malloc((size_t)k * n_e * 2);
        for (int pass = 0; pass < 2 && !rc; ++pass) {
            memset(codes, 0xff, (size_t)k * n_e * 2);
            int nf;
            if (pass == 0) {
                simon_scenario sc = {n_e, x->scen[2 * x->ex_scenario + 1]};
                nf = simon_explain(c, sc, x->orders + (size_t)sc.order_id * P, failed, codes, k);
            } else {
                nf = simon_explain_loaded(c, x->ex_scenario, failed, codes, k);
            }
            if (nf < 0) { fprintf(stderr, "%s: explain pass %d -> %d: %s\n", who, pass, nf, simon_last_error(c)); rc = 1; break; }
            if (nf != x->g_unsched[x->ex_scenario] || memcmp(failed, x->ex_failed, (size_t)k * 4) || memcmp(codes, x->ex_codes, (size_t)k * n_e * 2)) {
                fprintf(stderr, "%s: explain pass %d differs from the fixture (n_failed %d)\n", who, pass, nf); rc = 1;
            }
        }
        free(failed); free(codes);
    }
    if (!rc) {   /* ABI v6: pods of unequal priority -> simon_fetch_preempt_risk; entries of quantity 0 (all-clear here: the fixture's results must not move) */
        int32_t* prio = malloc(P * 4); uint8_t* ent = calloc(P, 1); uint8_t* risk = malloc(S);
        simon_batch_out out; memset(&out, 0, sizeof out);
        out.struct_size = sizeof out; out.unscheduled = un; out.used_cpu = uc; out.used_mem = um; out.placement = pl;
        for (size_t p = 0; p < P; ++p) prio[p] = (int32_t)(p % 3);
        TRY(simon_set_scalar_entries(c, ent));
        TRY(simon_set_pod_priorities(c, prio, 1));          /* a pod of priority 1 was bound before the stream */
        TRY(simon_run_batch(c, (const simon_scenario*)x->scen, x->S, x->orders, x->n_orders, &out));
        rc = compare_batch(x, un, uc, um, pl, who);
        TRY(simon_fetch_preempt_risk(c, risk));
        for (size_t s2 = 0; s2 < S && !rc; ++s2) {
            /* recomputed here from the placement row: some pod failed while a pod of lower priority was placed before it */
            int low = 1, want = 0;
            const int32_t* ord = x->orders + (size_t)x->scen[2 * s2 + 1] * P;
            for (size_t i = 0; i < P; ++i) {
                const int32_t v = pl[s2 * P + ord[i]], pr = prio[ord[i]];
                if (v >= 0) { if (pr < low) low = pr; }
                else if (v == SIMON_UNSCHEDULED && pr > low) want = 1;
            }
            if (risk[s2] != want) { fprintf(stderr, "%s: preempt_risk[%zu] = %d, expected %d\n", who, s2, risk[s2], want); rc = 1; }
        }
        TRY(simon_set_pod_priorities(c, NULL, 0)); TRY(simon_set_scalar_entries(c, NULL));
        free(prio); free(ent); free(risk);
    }
    if (!rc) {
        simon_stats st;
        TRY(simon_get_stats(c, &st));
        printf("%s: %d scenarios x %d pods ok (kernel variant %d, %.3f ms)\n", who, x->S, x->P, st.kernel_variant, st.kernel_ms);
    }
done:
    free(un); free(uc); free(um); free(pl); free(gs);
    simon_ctx_destroy(c);
    return rc;
#undef TRY
}

/* the multi-device entry: the batch dealt over a device list, minimum plan reduced across the members */
static int run_group(const fixture* x, const int32_t* ids, int n_dev) {
    if (x->S < n_dev) return 0;
    simon_group* g = simon_group_create(ids, n_dev);
    CHECK(g, "simon_group_create failed");
    simon_nodes_soa nd; simon_pods_soa pd; simon_class_tables tb;
    fill_inputs(x, &nd, &pd, &tb);
    const size_t S = x->S, P = x->P;
    int32_t* un = malloc(S * 4); int64_t* uc = malloc(S * 8); int64_t* um = malloc(S * 8); int32_t* pl = malloc(S * P * 4);
    int rc = 0;
#define TRY(call) do { int r_ = (call); if (r_ < 0) { fprintf(stderr, "group: %s -> %d: %s\n", #call, r_, simon_group_last_error(g)); rc = 1; goto done; } } while (0)
    TRY(simon_group_load_nodes(g, &nd)); TRY(simon_group_load_pods(g, &pd)); TRY(simon_group_load_class_tables(g, &tb));
    simon_batch_out out; memset(&out, 0, sizeof out);
    out.struct_size = sizeof out; out.unscheduled = un; out.used_cpu = uc; out.used_mem = um; out.placement = pl;
    TRY(simon_group_run_batch(g, (const simon_scenario*)x->scen, x->S, x->orders, x->n_orders, &out));
    rc = compare_batch(x, un, uc, um, pl, "group");
    if (!rc) {
        simon_plan plan; int32_t vg = -1;
        TRY(simon_group_min_plan(g, 100, 100, 100, &plan, &vg));
        if (plan.found != x->plan_found || (plan.found && (plan.scenario != x->plan_scenario || plan.n_nodes != x->plan_n_nodes))) {
            fprintf(stderr, "group: plan {%d, s=%d, n=%d}, fixture {%d, s=%d, n=%d}\n", plan.found, plan.scenario, plan.n_nodes, x->plan_found,
                    x->plan_scenario, x->plan_n_nodes);
            rc = 1;
        }
    }
    if (!rc) printf("group of %d members: %d scenarios ok\n", simon_group_size(g), x->S);
done:
    free(un); free(uc); free(um); free(pl);
    simon_group_destroy(g);
    return rc;
#undef TRY
}

typedef struct thread_arg { const fixture* x; int device; int rc; char name[32]; } thread_arg;
static void* thread_main(void* p) {
    thread_arg* a = p;
    a->rc = run_single(a->x, a->device, 3, a->name);
    return NULL;
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s fixture.bin [fixture.bin ...]\n", argv[0]); return 1; }
    if (simon_hip_version() != SIMON_HIP_ABI_VERSION) { fprintf(stderr, "ABI mismatch: header %d, library %d\n", SIMON_HIP_ABI_VERSION, simon_hip_version()); return 1; }
    if (simon_hip_device_count() < 1) { printf("cabi_smoke: library loaded (ABI %d), no GPU visible\n", simon_hip_version()); return 77; }
    for (int i = 1; i < argc; ++i) {
        fixture x;
        load_fixture(argv[i], &x);
        printf("== %s: %d nodes, %d pods, %d scenarios\n", argv[i], x.N, x.P, x.S);
        if (run_single(&x, 0, 2, "single context")) return 1;
        const int32_t ids[2] = {0, 0};
        if (run_group(&x, ids, 2)) return 1;
        /* two contexts on two OS threads (the header: distinct contexts may live on distinct threads) */
        thread_arg a[2] = {{&x, 0, 0, "thread A"}, {&x, 0, 0, "thread B"}};
        pthread_t th[2];
        for (int t = 0; t < 2; ++t) pthread_create(&th[t], NULL, thread_main, &a[t]);
        for (int t = 0; t < 2; ++t) pthread_join(th[t], NULL);
        if (a[0].rc || a[1].rc) return 1;
    }
    printf("cabi_smoke ok\n");
    return 0;
}
