/*
 * simon_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, single-threaded) of open-simulator's pod-to-node assignment hot path,
 * used only as the parity checker by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg.  Nothing under open-simulator_amd/ may include, link or call it.
 *
 * Parity status: "parity unpinned" at node-placement level -- the reference's only test
 * (pkg/simulator/core_test.go:32-591) asserts feasibility COUNTS, never which node a pod lands
 * on, the vendored scheduler's own tests are stripped, and there is no Go toolchain in the build
 * container to run the reference.  The oracle is therefore pinned against (a) the hand-derived
 * known-answer vector of SURVEY.md section 8(c), (b) the count-level expectations of
 * core_test.go "simple" and the example/ gpushare fixtures, restated as SoA inputs in
 * tests/golden/ (see tests/golden/README.md).
 *
 * It shares only the plain-data input/output structs with include/simon_hip.h.
 */
#ifndef SIMON_ORACLE_H
#define SIMON_ORACLE_H

#include "../include/simon_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Run S scenarios serially.  fail_* are optional: when non-NULL, the per-node failure codes of
 * the first max_failed unscheduled pods OF SCENARIO explain_scenario are written. */
int simon_oracle_run(const simon_nodes_soa* nodes, const simon_pods_soa* pods,
                     const simon_class_tables* tables, const simon_scenario* scen, int32_t S,
                     const int32_t* orders, int32_t n_orders, simon_batch_out* out,
                     int32_t explain_scenario, int32_t* failed_pods, uint16_t* fail_codes,
                     int32_t max_failed, int32_t* n_failed_out);

/* Where the NEXT explaining run leaves what Open-Local's error texts carry: [max_failed][n_nodes of the scenario][4] int64
 * {SIMON_LOCAL_ERR_*, a, b, c} (include/simon_hip.h), zero-initialised by the caller; NULL switches it off. */
void simon_oracle_set_local_detail(int64_t* buf);
/* ABI v6 inputs of the NEXT simon_oracle_run / _run_ranked of the calling thread (they travel outside simon_pods_soa in the library too:
 * simon_set_scalar_entries, simon_set_pod_priorities); risk_out [S] receives simon_fetch_preempt_risk's flags.  NULLs clear. */
void simon_oracle_set_v6(const uint8_t* entries, const int32_t* priority, int32_t init_min_priority, uint8_t* risk_out);

/* The same with a canonical node order per scenario: node_rank[s][j] = position of pool node j in scenario s's nodeTree
 * order (V/internal/cache/node_tree.go:119-143); selectHost's first maximum is taken in that order. */
int simon_oracle_run_ranked(const simon_nodes_soa* nodes, const simon_pods_soa* pods,
                            const simon_class_tables* tables, const simon_scenario* scen, int32_t S,
                            const int32_t* orders, int32_t n_orders, const int32_t* node_rank,
                            simon_batch_out* out, int32_t explain_scenario, int32_t* failed_pods, uint16_t* fail_codes,
                            int32_t max_failed, int32_t* n_failed_out);

/* Per-node score breakdown for ONE pod against the CURRENT (initial) node state of the first
 * n_nodes pool nodes: feasible[j], la[j], ba[j], sn[j], total[j] (all int64 [n_nodes]).
 * Used to check the hand-derived known-answer vector. */
int simon_oracle_score_pod(const simon_nodes_soa* nodes, const simon_pods_soa* pods,
                           const simon_class_tables* tables, int32_t n_nodes, int32_t pod,
                           int64_t* feasible, int64_t* la, int64_t* ba, int64_t* sn, int64_t* total);

/* Same breakdown after the first n_before pods of `order` (NULL = identity) were scheduled by the
 * oracle; codes_out (optional, [n_nodes]) receives the per-node SIMON_FAIL_* codes of `pod`. */
int simon_oracle_score_pod_after(const simon_nodes_soa* nodes, const simon_pods_soa* pods,
                                 const simon_class_tables* tables, int32_t n_nodes, const int32_t* order,
                                 int32_t n_before, int32_t pod, int64_t* feasible, int64_t* la, int64_t* ba,
                                 int64_t* sn, int64_t* total, uint16_t* codes_out);

/* The add-nodes search result over a finished batch (pkg/apply/apply.go:203-259, 689-775). */
int simon_oracle_min_plan(const simon_nodes_soa* nodes, const simon_scenario* scen, int32_t S,
                          const simon_batch_out* out, int32_t max_cpu_pct, int32_t max_mem_pct,
                          simon_plan* best);
/* ... with the MaxVG cap (pkg/apply/apply.go:712-716, 747-771); vg_pct may be NULL. */
int simon_oracle_min_plan_vg(const simon_nodes_soa* nodes, const simon_scenario* scen, int32_t S,
                             const simon_batch_out* out, int32_t max_cpu_pct, int32_t max_mem_pct, int32_t max_vg_pct,
                             simon_plan* best, int32_t* vg_pct);

/* apiresource.Quantity as (unscaled int64 value, decimal scale) or (value, binary) -- enough of
 * vendor/k8s.io/apimachinery/pkg/api/resource/quantity.go to restate SimonPlugin.Score. */
typedef struct simon_quantity {
    int64_t value; /* unscaled */
    int32_t scale; /* decimal exponent (int64Amount.scale); quantities parsed from binary-SI strings have scale 0 */
} simon_quantity;

/* SimonPlugin.Score raw value for one (pod, node): iterates the node's allocatable resources
 * (pkg/simulator/plugin/simon.go:45-68).  n_res parallel arrays; pod_has_any_request = 0 gives
 * MaxNodeScore (simon.go:47-49). */
int64_t simon_oracle_simon_raw(const simon_quantity* pod_req, const simon_quantity* node_alloc,
                               int32_t n_res, int32_t pod_has_any_request);

/* splitmix64 synthetic generator of SURVEY.md section 8(d); twin of open-simulator_amd/synth.py */
uint64_t simon_oracle_splitmix64(uint64_t* state);
int simon_oracle_gen_nodes(uint64_t seed, int32_t n_het, int32_t n_total, int64_t* alloc_cpu,
                           int64_t* alloc_mem, int32_t* alloc_pods, int32_t* node_class);
int simon_oracle_gen_pods(uint64_t seed, int32_t n_pods, int64_t* req_cpu, int64_t* req_mem);

#ifdef __cplusplus
}
#endif
#endif
