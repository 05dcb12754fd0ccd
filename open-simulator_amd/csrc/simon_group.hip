// simon_group.hip -- several devices behind ONE handle (include/simon_hip.h, "device groups").
//
// The reference's add-nodes loop calls Simulate once per candidate cluster size, serially
// (pkg/apply/apply.go:203-259).  Scenarios are independent, so a group deals scenario s of a batch
// to member s % n_dev (with the batch ordered count-major every member sees every node count:
// balanced, cost grows with the count), replicates the immutable inputs, runs the members
// concurrently -- one host thread per member for the duration of a call, every member on its own
// HIP device and stream.  The one cross-device exchange of the path is the minimum-node plan
// (north_star: "RCCL all-gather over xGMI only to collect the global minimum-node plan"):
//   * members on DISTINCT devices: one ncclAllGather of the 8-byte plan key (n_nodes << 32 | scenario, written by plan_kernel)
//     straight from device memory, on every member's own stream (enqueued for all members by ONE thread inside ncclGroupStart /
//     ncclGroupEnd: no rank can miss the collective), through communicators made once by ncclCommInitAll at
//     group creation; every member then holds all keys, member 0's copy is read back and the lexicographic minimum
//     (n_nodes, global scenario) picks the winner, whose own context fills in the occupancy figures;
//   * one member, members sharing a device (tests), RCCL not loadable / failing, or SIMON_GROUP_RCCL=0: the same
//     minimum over the members' host-side plans, no collective.
// simon_group_collective() says which one the last simon_group_min_plan took.  librccl is bound at run time (dlopen):
// a single-GPU host needs neither the library nor a communicator.  (With one PROCESS per GPU the same record travels
// through torch.distributed's RCCL: bench.py / open-simulator_amd/sweep.py.)
//
// Scenario work is built on the public single-device entry points only; the group's own HIP state is the gather buffers.
#include "../../include/simon_hip.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

struct simon_group {
    std::vector<simon_ctx*> ctx;
    std::vector<int32_t> device;
    std::string err;
    int32_t S = 0, P = 0;                       // last loaded batch: global scenario count; pods
    std::vector<std::vector<simon_scenario>> part;   // per member: its scenarios, in global order
    bool have_results = false, have_placement = false, have_slices = false;
    // RCCL (bound at run time): communicators of the members, gather buffers [n_dev] keys on every member's device
    void* rccl = nullptr;
    ncclResult_t (*p_init_all)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*p_all_gather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*p_destroy)(ncclComm_t) = nullptr;
    ncclResult_t (*p_abort)(ncclComm_t) = nullptr;
    ncclResult_t (*p_group_start)() = nullptr;
    ncclResult_t (*p_group_end)() = nullptr;
    const char* (*p_errstr)(ncclResult_t) = nullptr;
    std::vector<ncclComm_t> comm;
    std::vector<void*> d_gather;
    bool comm_ok = false;
    bool fault_inject = false;                  // env SIMON_GROUP_FAULT_INJECT (tests): the next collective is treated as failed after it was enqueued
    int32_t collective = 0;                     // what the last min_plan did: 0 host reduction, 1 RCCL all-gather
};

namespace {

int gfail(simon_group* g, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (g) g->err = buf;
    return code;
}

// run fn(member) on every member concurrently; first failing member's code + message win.  Nothing C++ crosses the C ABI: a
// member body that throws (bad_alloc of its staging vectors) reports SIMON_ENOMEM, and a thread that cannot be started
// (std::system_error) makes the caller run that member itself.
template <class F>
int on_all(simon_group* g, const char* what, F fn) {
    const int n = (int)g->ctx.size();
    int rcs[64] = {0};                                       // simon_group_create admits at most 64 members
    auto guarded = [&](int i) {
        try { rcs[i] = fn(i); }
        catch (const std::bad_alloc&) { rcs[i] = SIMON_ENOMEM; }
        catch (...) { rcs[i] = SIMON_ENOMEM; }
    };
    if (n == 1) {
        guarded(0);
    } else {
        std::thread th[64];
        bool started[64] = {false};
        for (int i = 0; i < n; ++i) {
            try { th[i] = std::thread(guarded, i); started[i] = true; }
            catch (...) { started[i] = false; }
        }
        for (int i = 0; i < n; ++i) if (!started[i]) guarded(i);
        for (int i = 0; i < n; ++i) if (started[i]) th[i].join();
    }
    for (int i = 0; i < n; ++i)
        if (rcs[i] < 0) {
            const char* why = rcs[i] == SIMON_ENOMEM ? "out of host memory" : simon_last_error(g->ctx[i]);
            return gfail(g, rcs[i], "%s: member %d (device %d): %s", what, i, g->device[i], why);
        }
    return SIMON_OK;
}

// Communicators for members on distinct devices (SIMON_GROUP_RCCL=1: also for a single member -- the test hook that runs the
// collective path on a one-GPU box; =0: never).  Any failure leaves comm_ok false: the host reduction is always available.
void rccl_setup(simon_group* g) {
    const int n = (int)g->ctx.size();
    const char* knob = getenv("SIMON_GROUP_RCCL");
    if (knob && knob[0] == '0') return;
    if (n < 2 && !(knob && knob[0] == '1')) return;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < i; ++j)
            if (g->device[i] == g->device[j]) return;           // one communicator rank per device
    // the copy the host process has mapped already, if any (a Python host with torch: torch/lib/librccl.so -- two copies of RCCL in
    // one process would each bring their own bootstrap state), else the system's
    g->rccl = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
    if (!g->rccl) g->rccl = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
    if (!g->rccl) g->rccl = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!g->rccl) g->rccl = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!g->rccl) return;
    g->p_init_all = reinterpret_cast<decltype(g->p_init_all)>(dlsym(g->rccl, "ncclCommInitAll"));
    g->p_all_gather = reinterpret_cast<decltype(g->p_all_gather)>(dlsym(g->rccl, "ncclAllGather"));
    g->p_destroy = reinterpret_cast<decltype(g->p_destroy)>(dlsym(g->rccl, "ncclCommDestroy"));
    g->p_errstr = reinterpret_cast<decltype(g->p_errstr)>(dlsym(g->rccl, "ncclGetErrorString"));
    g->p_abort = reinterpret_cast<decltype(g->p_abort)>(dlsym(g->rccl, "ncclCommAbort"));
    g->p_group_start = reinterpret_cast<decltype(g->p_group_start)>(dlsym(g->rccl, "ncclGroupStart"));
    g->p_group_end = reinterpret_cast<decltype(g->p_group_end)>(dlsym(g->rccl, "ncclGroupEnd"));
    // ncclCommAbort is REQUIRED: it is what ends a half-entered collective on the failure path (without it the communicators would stay
    // non-null and ncclCommDestroy on them could block in rccl_teardown) -- a library without the symbol reduces on the host
    if (!g->p_init_all || !g->p_all_gather || !g->p_destroy || !g->p_group_start || !g->p_group_end || !g->p_abort) return;
    try { g->comm.assign(n, nullptr); g->d_gather.assign(n, nullptr); } catch (...) { return; }
    std::vector<int> devs(g->device.begin(), g->device.end());
    if (g->p_init_all(g->comm.data(), n, devs.data()) != ncclSuccess) { g->comm.clear(); return; }
    for (int i = 0; i < n; ++i) {
        if (hipSetDevice(g->device[i]) != hipSuccess || hipMalloc(&g->d_gather[i], (size_t)n * sizeof(unsigned long long)) != hipSuccess) return;
    }
    g->comm_ok = true;
    g->fault_inject = getenv("SIMON_GROUP_FAULT_INJECT") != nullptr;
}

void rccl_teardown(simon_group* g) {
    for (size_t i = 0; i < g->d_gather.size(); ++i)
        if (g->d_gather[i]) { (void)hipSetDevice(g->device[i]); (void)hipFree(g->d_gather[i]); }
    if (g->p_destroy) for (ncclComm_t c : g->comm) if (c) (void)g->p_destroy(c);
    if (g->rccl) dlclose(g->rccl);
}

// simon_batch_out as the caller filled it, widened to the current layout: a v3 struct (no gpu_slices member) reads as gpu_slices = NULL
bool batch_out_view(const simon_batch_out* out, simon_batch_out* v) {
    if (out->struct_size == sizeof(simon_batch_out)) { *v = *out; return true; }
    if (out->struct_size != SIMON_BATCH_OUT_SIZE_V3) return false;
    std::memset(v, 0, sizeof *v);
    std::memcpy(v, out, SIMON_BATCH_OUT_SIZE_V3);
    v->struct_size = sizeof *v;
    return true;
}

}  // namespace

extern "C" {

simon_group* simon_group_create(const int32_t* device_ids, int32_t n_dev) {
    if (!device_ids || n_dev <= 0 || n_dev > 64) return nullptr;
    simon_group* g = new (std::nothrow) simon_group();
    if (!g) return nullptr;
    try {
        g->ctx.reserve(n_dev);
        g->device.reserve(n_dev);
        g->part.resize(n_dev);
    } catch (...) { delete g; return nullptr; }
    for (int i = 0; i < n_dev; ++i) {
        simon_ctx* c = simon_ctx_create(device_ids[i]);
        if (!c) { simon_group_destroy(g); return nullptr; }
        g->ctx.push_back(c);                                // capacity reserved above: cannot throw
        g->device.push_back(device_ids[i]);
    }
    rccl_setup(g);
    return g;
}

void simon_group_destroy(simon_group* g) {
    if (!g) return;
    rccl_teardown(g);
    for (simon_ctx* c : g->ctx) simon_ctx_destroy(c);
    delete g;
}

int32_t simon_group_collective(simon_group* g) { return g ? g->collective : -1; }

const char* simon_group_last_error(simon_group* g) { return g ? g->err.c_str() : "null group"; }
int32_t simon_group_size(simon_group* g) { return g ? (int32_t)g->ctx.size() : 0; }
simon_ctx* simon_group_member(simon_group* g, int32_t i) { return (g && i >= 0 && i < (int32_t)g->ctx.size()) ? g->ctx[i] : nullptr; }

int simon_group_load_nodes(simon_group* g, const simon_nodes_soa* nodes) {
    if (!g || !nodes) return SIMON_EINVAL;
    g->have_results = false;
    return on_all(g, "load_nodes", [&](int i) { return simon_load_nodes(g->ctx[i], nodes); });
}

int simon_group_load_pods(simon_group* g, const simon_pods_soa* pods) {
    if (!g || !pods) return SIMON_EINVAL;
    g->have_results = false;
    g->P = pods->n_pods;
    return on_all(g, "load_pods", [&](int i) { return simon_load_pods(g->ctx[i], pods); });
}

int simon_group_set_scalar_entries(simon_group* g, const uint8_t* entries) {
    if (!g) return SIMON_EINVAL;
    g->have_results = false;
    return on_all(g, "set_scalar_entries", [&](int i) { return simon_set_scalar_entries(g->ctx[i], entries); });
}

int simon_group_set_pod_priorities(simon_group* g, const int32_t* priority, int32_t init_min_priority) {
    if (!g) return SIMON_EINVAL;
    return on_all(g, "set_pod_priorities", [&](int i) { return simon_set_pod_priorities(g->ctx[i], priority, init_min_priority); });
}

int simon_group_load_class_tables(simon_group* g, const simon_class_tables* tables) {
    if (!g || !tables) return SIMON_EINVAL;
    g->have_results = false;
    return on_all(g, "load_class_tables", [&](int i) { return simon_load_class_tables(g->ctx[i], tables); });
}

int simon_group_load_scenarios(simon_group* g, const simon_scenario* scen, int32_t S, const int32_t* orders, int32_t n_orders) {
    if (!g || !scen || S <= 0 || !orders || n_orders <= 0) return g ? gfail(g, SIMON_EINVAL, "group load_scenarios: bad arguments") : SIMON_EINVAL;
    const int n = (int)g->ctx.size();
    if (S < n) return gfail(g, SIMON_EINVAL, "group load_scenarios: %d scenarios for %d members (every member needs one)", S, n);
    // the previous batch is gone as soon as one member may hold the new one: a failed load leaves the group without scenarios
    // (run_loaded / fetch_results then refuse) instead of members holding a mix of old and new batches
    g->S = 0;
    g->have_results = false;
    g->have_placement = false;
    try {
        for (int i = 0; i < n; ++i) g->part[i].clear();
        for (int s = 0; s < S; ++s) g->part[s % n].push_back(scen[s]);
    } catch (...) {
        return gfail(g, SIMON_ENOMEM, "group load_scenarios: out of host memory");
    }
    int rc = on_all(g, "load_scenarios", [&](int i) {
        return simon_load_scenarios(g->ctx[i], g->part[i].data(), (int32_t)g->part[i].size(), orders, n_orders);
    });
    if (rc == SIMON_OK) g->S = S;
    return rc;
}

int simon_group_run_loaded(simon_group* g, int32_t want_placement) {
    if (!g) return SIMON_EINVAL;
    if (g->S <= 0) return gfail(g, SIMON_ESTATE, "group run_loaded: no scenarios loaded");
    int rc = on_all(g, "run_loaded", [&](int i) { return simon_run_loaded(g->ctx[i], want_placement); });
    g->have_results = rc == SIMON_OK;
    g->have_placement = g->have_results && (want_placement & SIMON_WANT_PLACEMENT) != 0;
    g->have_slices = g->have_results && (want_placement & SIMON_WANT_GPU_SLICES) != 0;
    return rc;
}

int simon_group_fetch_results(simon_group* g, simon_batch_out* caller_out) {
    if (!g || !caller_out) return SIMON_EINVAL;
    simon_batch_out view, *out = &view;
    if (!batch_out_view(caller_out, out)) return gfail(g, SIMON_EINVAL, "simon_batch_out size mismatch");
    if (!g->have_results) return gfail(g, SIMON_ESTATE, "group fetch_results: nothing has run");
    if (out->placement && !g->have_placement) return gfail(g, SIMON_ESTATE, "group fetch_results: the last run skipped placements");
    const int n = (int)g->ctx.size();
    const size_t P = (size_t)g->P;
    return on_all(g, "fetch_results", [&](int i) {
        const size_t Si = g->part[i].size();
        std::vector<int32_t> un(Si);
        std::vector<int64_t> uc(Si), um(Si), uv(out->used_vg ? Si : 0);
        std::vector<int32_t> pl(out->placement ? Si * P : 0);
        std::vector<uint64_t> gs(out->gpu_slices ? Si * P : 0);
        simon_batch_out o{};
        o.struct_size = sizeof o;
        o.unscheduled = un.data(); o.used_cpu = uc.data(); o.used_mem = um.data();
        o.used_vg = out->used_vg ? uv.data() : nullptr;
        o.placement = out->placement ? pl.data() : nullptr;
        o.gpu_slices = out->gpu_slices ? gs.data() : nullptr;
        int rc = simon_fetch_results(g->ctx[i], &o);
        if (rc) return rc;
        for (size_t k = 0; k < Si; ++k) {                       // member-local index k = global scenario k * n + i
            const size_t s = k * n + i;
            if (out->unscheduled) out->unscheduled[s] = un[k];
            if (out->used_cpu) out->used_cpu[s] = uc[k];
            if (out->used_mem) out->used_mem[s] = um[k];
            if (out->used_vg) out->used_vg[s] = uv[k];
            if (out->placement) memcpy(out->placement + s * P, pl.data() + k * P, P * sizeof(int32_t));
            if (out->gpu_slices) memcpy(out->gpu_slices + s * P, gs.data() + k * P, P * sizeof(uint64_t));
        }
        return (int)SIMON_OK;
    });
}

int simon_group_run_batch(simon_group* g, const simon_scenario* scen, int32_t S, const int32_t* orders, int32_t n_orders,
                          simon_batch_out* out) {
    if (!g || !out) return SIMON_EINVAL;
    simon_batch_out view;
    if (!batch_out_view(out, &view)) return gfail(g, SIMON_EINVAL, "simon_batch_out size mismatch");
    int rc = simon_group_load_scenarios(g, scen, S, orders, n_orders);
    if (rc) return rc;
    rc = simon_group_run_loaded(g, (view.placement ? SIMON_WANT_PLACEMENT : 0) | (view.gpu_slices ? SIMON_WANT_GPU_SLICES : 0));
    if (rc) return rc;
    return simon_group_fetch_results(g, &view);
}

int simon_group_fetch_placement(simon_group* g, int32_t scenario, int32_t* placement) {
    if (!g || !placement) return SIMON_EINVAL;
    if (!g->have_results || !g->have_placement) return gfail(g, SIMON_ESTATE, "group fetch_placement: no placements on the devices");
    if (scenario < 0 || scenario >= g->S) return gfail(g, SIMON_EINVAL, "group fetch_placement: scenario out of range");
    const int n = (int)g->ctx.size(), i = scenario % n;
    int rc = simon_fetch_placement(g->ctx[i], scenario / n, placement);
    return rc ? gfail(g, rc, "fetch_placement: member %d: %s", i, simon_last_error(g->ctx[i])) : SIMON_OK;
}

int simon_group_fetch_preempt_risk(simon_group* g, uint8_t* risk) {
    if (!g || !risk) return SIMON_EINVAL;
    if (!g->have_results) return gfail(g, SIMON_ESTATE, "group fetch_preempt_risk: nothing has run");
    const int n = (int)g->ctx.size();
    return on_all(g, "fetch_preempt_risk", [&](int i) {
        std::vector<uint8_t> r(g->part[i].size());
        int rc = simon_fetch_preempt_risk(g->ctx[i], r.data());
        if (rc) return rc;
        for (size_t k = 0; k < r.size(); ++k) risk[k * n + i] = r[k];       // member-local index k = global scenario k * n + i
        return (int)SIMON_OK;
    });
}

int simon_group_fetch_gpu_slices(simon_group* g, int32_t scenario, uint64_t* slices) {
    if (!g || !slices) return SIMON_EINVAL;
    if (!g->have_results) return gfail(g, SIMON_ESTATE, "group fetch_gpu_slices: nothing has run");
    if (scenario < 0 || scenario >= g->S) return gfail(g, SIMON_EINVAL, "group fetch_gpu_slices: scenario out of range");
    const int n = (int)g->ctx.size(), i = scenario % n;
    int rc = simon_fetch_gpu_slices(g->ctx[i], scenario / n, slices);
    return rc ? gfail(g, rc, "fetch_gpu_slices: member %d: %s", i, simon_last_error(g->ctx[i])) : SIMON_OK;
}

int simon_group_min_plan(simon_group* g, int32_t max_cpu_pct, int32_t max_mem_pct, int32_t max_vg_pct, simon_plan* best,
                         int32_t* vg_pct) {
    if (!g || !best) return SIMON_EINVAL;
    if (!g->have_results) return gfail(g, SIMON_ESTATE, "group min_plan: nothing has run");
    const int n = (int)g->ctx.size();
    g->collective = 0;
    if (g->comm_ok) {
        // every member: plan_kernel leaves its 8-byte key in device memory; then ONE all-gather of the keys, device to device
        void* d_key[64] = {nullptr};
        void* stream[64] = {nullptr};
        int rc = on_all(g, "min_plan (plan kernel)", [&](int i) { return simon_min_plan_device(g->ctx[i], max_cpu_pct, max_mem_pct, max_vg_pct, &d_key[i], &stream[i]); });
        if (rc) return rc;
        // ONE thread enqueues the collective for every member inside ncclGroupStart / ncclGroupEnd (the single-process multi-GPU
        // form RCCL documents): no member can miss the collective because its host thread failed to start or its device could not
        // be selected -- either every rank is enqueued or, on any error, the communicators are aborted (which ends whatever was
        // launched, so no stream is left waiting for a peer) and this call and all later ones reduce on the host instead.
        ncclResult_t nr = g->p_group_start();
        int bad = -1;
        if (nr == ncclSuccess) {
            for (int i = 0; i < n; ++i) {
                const ncclResult_t r = g->p_all_gather(d_key[i], g->d_gather[i], 1, ncclUint64, g->comm[i], (hipStream_t)stream[i]);
                if (r != ncclSuccess) { nr = r; bad = i; break; }   // (no further rank is enqueued behind a failed one; the group is still closed, then aborted)
            }
            const ncclResult_t r = g->p_group_end();
            if (r != ncclSuccess && nr == ncclSuccess) nr = r;
        }
        if (g->fault_inject && nr == ncclSuccess) { nr = ncclInternalError; g->fault_inject = false; }   // (test hook: the abort + host fall-back below, with the collective in flight)
        hipError_t he = hipSuccess;
        if (nr == ncclSuccess)
            for (int i = 0; i < n && he == hipSuccess; ++i) {
                if ((he = hipSetDevice(g->device[i])) == hipSuccess) he = hipStreamSynchronize((hipStream_t)stream[i]);
                if (he != hipSuccess) bad = i;
            }
        unsigned long long keys[64];
        if (nr == ncclSuccess && he == hipSuccess &&
            (hipSetDevice(g->device[0]) != hipSuccess || hipMemcpy(keys, g->d_gather[0], (size_t)n * sizeof keys[0], hipMemcpyDeviceToHost) != hipSuccess))
            he = hipErrorUnknown;
        if (nr != ncclSuccess || he != hipSuccess) {
            // first-run safety: never leave a half-entered collective behind
            for (ncclComm_t& c : g->comm) if (c) { (void)g->p_abort(c); c = nullptr; }
            g->comm_ok = false;
            (void)gfail(g, SIMON_ENODEV, "min_plan: RCCL all-gather failed (member %d: %s); reduced on the host instead", bad,
                        nr != ncclSuccess ? (g->p_errstr ? g->p_errstr(nr) : "nccl error") : hipGetErrorString(he));
            goto host_reduction;
        }
        g->collective = 1;
        memset(best, 0, sizeof *best);
        best->scenario = -1;
        if (vg_pct) *vg_pct = 0;
        int w = -1;
        unsigned long long wn = 0, ws = 0;
        for (int i = 0; i < n; ++i) {                          // key = n_nodes << 32 | member-local scenario; ~0 = no qualifying scenario
            if (keys[i] == ~0ull) continue;
            const unsigned long long nn = keys[i] >> 32, gs = (keys[i] & 0xFFFFFFFFull) * (unsigned long long)n + (unsigned long long)i;
            if (w < 0 || nn < wn || (nn == wn && gs < ws)) { w = i; wn = nn; ws = gs; }
        }
        if (w < 0) return SIMON_OK;
        int32_t vgw = 0;
        rc = simon_min_plan_vg(g->ctx[w], max_cpu_pct, max_mem_pct, max_vg_pct, best, &vgw);   // the winner's own record (occupancy figures)
        if (rc) return gfail(g, rc, "min_plan: member %d: %s", w, simon_last_error(g->ctx[w]));
        best->scenario = (int32_t)ws;
        if (vg_pct) *vg_pct = vgw;
        return SIMON_OK;
    }
host_reduction:
    simon_plan plans[64];                                   // at most 64 members (simon_group_create)
    int32_t vg[64] = {0};
    memset(plans, 0, sizeof plans);
    int rc = on_all(g, "min_plan", [&](int i) { return simon_min_plan_vg(g->ctx[i], max_cpu_pct, max_mem_pct, max_vg_pct, &plans[i], &vg[i]); });
    if (rc) return rc;
    memset(best, 0, sizeof *best);
    best->scenario = -1;
    if (vg_pct) *vg_pct = 0;
    // lexicographic minimum (n_nodes, global scenario index): the rule of simon_min_plan carried across members
    for (int i = 0; i < n; ++i) {
        if (!plans[i].found) continue;
        const int32_t gs = plans[i].scenario * n + i;
        if (!best->found || plans[i].n_nodes < best->n_nodes || (plans[i].n_nodes == best->n_nodes && gs < best->scenario)) {
            *best = plans[i];
            best->scenario = gs;
            if (vg_pct) *vg_pct = vg[i];
        }
    }
    return SIMON_OK;
}

}  // extern "C"
