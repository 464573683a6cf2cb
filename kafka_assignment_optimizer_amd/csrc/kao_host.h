// kao_host.h -- host-side internals shared by the translation units behind the C ABI (include/kao.h):
//   kao_model.cpp    the model on the host: validation, bands, dense -> rack-major index, infeasibility proofs, the closed-form
//                    upper bound (kao_upper_bound)
//   kao_session.cpp  runtime (device, error text, arena / stream pools), K-eval plans, canonical tie-break, sessions
//                    (K-search / K-eval steps, K-bound launches, prices)
//   kao_solve.cpp    the solve loops on top of sessions: kao_solve, kao_solve_multi (RCCL), kao_solve_capped
// Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/kao.h"
#include "kao_internal.h"

namespace kao {

// ---- runtime (kao_session.cpp) ----
int fail(int code, const std::string &msg);   // records the text kao_last_error returns; returns `code`
#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return ::kao::fail(KAO_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)
double now_s();
int cur_device();                 // the calling thread's device: t_device when set, else the process default (kao_init)
constexpr int kMaxDevices = 64;   // device ordinals the per-device tables cover
int num_cu(int device);
int require_init();               // kao_init on first use, then hipSetDevice(cur_device())
bool is_init();
extern thread_local int t_device; // per-thread override: kao_solve_multi drives several devices from one process
extern thread_local double g_timing[16];
extern thread_local double g_lp[8];       // KAO-LP in the last kao_solve (kao_last_solve_lp)
extern thread_local double g_profile[8];  // K-search as the last profiled kao_solve ran it (kao_last_solve_profile)   // wall-clock breakdown of the last solve (kao_last_solve_timing)

// ---- the model on the host (kao_model.cpp) ----
int validate(const kao_topic *t);
void derive_bounds(const kao_topic *t, int32_t o[8]);
// Host-side image of one topic in both index spaces.
struct PreparedTopic {
    TopicDev d{};
    std::vector<uint16_t> int_of;   // dense -> internal
    std::vector<uint16_t> ext_of;   // internal -> dense
    std::vector<int32_t> rack_size; // [R]
    std::vector<uint16_t> cur_int;  // [P*4] internal
    std::vector<uint8_t> rack_of;   // [B]
    std::vector<uint16_t> cur_dense;// [P*rf_cur]
    std::vector<uint32_t> bw_int, bw_dense;  // broker weights bw | bwl << 16 per internal / dense index (empty = none)
};
int prepare(const kao_topic *t, uint64_t seed, PreparedTopic &pt);
std::string infeasible_reason(const kao_topic *t);
int64_t upper_bound(const kao_topic *t);
int64_t upper_bound_w(const kao_topic *t);     // ... of a topic that may carry broker weights
uint64_t neighbours_in_range(uint32_t it0, uint32_t iters, int rf, int n_brokers, int n_partitions, int scan2_max = kScanTwoSlots);
int auto_period_log2(int P, int RF);
bool dual_supported(const kao_topic *t, bool session_bw = false);       // within K-bound's limits (session_bw: the session carves broker weights)

}  // namespace kao

using namespace kao;

struct kao_eval_plan {
    PreparedTopic pt;
    TopicDev *d_topic = nullptr;
    uint8_t *d_rackof = nullptr;
    uint16_t *d_curd = nullptr;
    int4 *d_map = nullptr;
    uint32_t *d_bwd = nullptr;      // broker weights (dense) when the topic has them
    int32_t *d_overflow = nullptr;  // set by K-eval when a candidate overflows a 16-bit per-broker counter (P*RF > 65535 only)
    int64_t map_n = -1;
    int map_blocks = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    int cands_per_block = 32;
    bool cur_in_lds = true;
    bool coop = false;            // the last batch ran one candidate per workgroup (k_eval<NE, true>)
    int device = 0;
};

struct kao_session {
    int device = 0;              // HIP device this session lives on
    int n_topics = 0;
    kao_opts opts{};
    std::vector<PreparedTopic> pts;
    std::vector<kao_topic> topics;  // shallow copies (pointers not retained for device work)
    std::vector<int64_t> ub;
    std::vector<char> topic_global;  // per topic: runs with its assignment in global memory
    std::vector<char> topic_curg;    // per topic: working assignment in LDS, current assignment from global memory (k_search_curg)
    std::vector<char> topic_infeasible;  // per topic: proven infeasible by counting (kao_check_infeasible)
    std::vector<char> dual_ok;           // per topic: within K-bound's limits
    std::vector<int64_t> h_dual_target;  // staging for kao_session_bound_step
    std::vector<int32_t> h_dual_ids;
    bool multi_off = false;              // k_bound_multi gave up once in this session (kao_session_bounds): the step kernels from then on
    std::vector<int2> h_wide_map;        // sliced K-bound: {topic, slice} per workgroup (staging, like h_dual_ids)
    uint64_t wide_ctl_i32 = 0, wide_map_i32 = 0;   // int32 offsets of the control blocks / the map inside d_dual
    std::vector<int32_t> dual_flags, dual_iters;
    int total_restarts = 0;
    // Topics are bucketed by LDS footprint into launch groups (a 3000-partition topic must not impose its LDS carve
    // and its 2 waves per workgroup on 200 small topics); one K-search + one K-eval launch per group per step.
    struct LaunchGroup {
        int maxP = 0, maxBx = 0, maxB = 0, maxR = 0;
        bool wide = false;   // some topic of the group has 512 replica slots or more (several tournament slots per lane)
        int waves = kWaves;  // restarts per K-search workgroup: 4, 2 or 1 -- the largest whose LDS carve fits 160 KiB
        int nw = kRFP;       // replica words per partition of the group's topics: 4 or 8 (template instantiation)
        int rf_uniform = -1; // the RF all topics of the group share (0: mixed; -1: no topic yet)
        bool global_a = false;   // topic too large for LDS: assignment + current words stay in global memory
        bool cur_global = false; // (round 5) only the current-assignment words stay in global memory / L2, the working words are in LDS (~4,900 .. 9,800 partitions)
        int team = 0;            // > 0 (global_a only): every restart is searched by a TEAM of that many wavefronts (k_team), one workgroup per restart
        bool cur_in_lds = true;  // K-eval stages the current assignment in LDS (false: reads it from global)
        bool eval_coop = false;  // K-eval: one candidate per workgroup, wavefronts cooperating (few large candidates)
        int smap_off = 0, smap_n = 0, emap_off = 0, emap_n = 0;
    };
    std::vector<LaunchGroup> groups;
    int blocks_search = 0, blocks_eval = 0;
    // device memory: one read-only arena (instance tables, uploaded with ONE H2D copy) and one mutable
    // arena (restart states, snapshots, results); the pointers below are carved from them
    void *arena_ro = nullptr, *arena_rw = nullptr;
    size_t arena_ro_bytes = 0, arena_rw_bytes = 0;
    TopicDev *d_topics = nullptr;
    int2 *d_smap = nullptr;
    int4 *d_emap = nullptr;
    uint32_t *d_cur = nullptr;
    uint16_t *d_ext = nullptr;
    int32_t *d_rsz = nullptr;
    uint8_t *d_rackof = nullptr;
    uint16_t *d_curd = nullptr;
    unsigned char *d_state = nullptr;
    uint16_t *d_best = nullptr;
    int32_t *d_info = nullptr;
    int32_t *d_obj = nullptr;
    int32_t *d_viol = nullptr;
    // read-back block (contiguous): [keys u64[T]] [drift i32 (16 B)] [win_viol i32[8T]] [win_assign u16[sum P*RF]]
    unsigned char *d_readback = nullptr;
    size_t readback_bytes = 0, rb_viol_off = 0, rb_assign_off = 0;
    // K-bound: multipliers + directions per topic; targets and workgroup->topic ids (host-written before a launch);
    // read-back block [best_L i64[T]] [info i32[4T]]
    int32_t *d_dual = nullptr;
    // search prices, double buffered: K-bound launch n exports into half (n & 1) while K-search reads the half of the last
    // launch whose results the host has merged (price_read); topics K-bound never covered read zeros
    int32_t *d_price = nullptr;
    size_t price_half_i32 = 0;
    int price_read = 0;          // half K-search reads
    int price_write_last = -1;   // half the K-bound launch in flight (or the last finished one) writes
    bool priced = false;         // K-search launches carry prices
    bool any_bw = false;         // some topic carries broker weights (their LDS table is carved in every launch group)
    uint16_t *d_int = nullptr;   // dense -> internal broker index per topic
    uint32_t *d_bw = nullptr, *d_bwd = nullptr;   // broker weights per internal / dense index (topics with has_bw)
    long long *d_dual_target = nullptr;
    int32_t *d_dual_ids = nullptr;
    unsigned char *d_dual_rb = nullptr;
    size_t dual_rb_bytes = 0;
    uint64_t bound_launches = 0;
    bool dual_state_init = false;   // the K-bound state in HBM has been cleared (first launch or kao_session_set_dual_state)
    size_t dual_bytes = 0;
    hipStream_t stream_bound = nullptr;   // K-bound runs beside K-search on its own stream (it occupies one CU per topic)
    hipEvent_t ev_bound0 = nullptr, ev_bound1 = nullptr, ev_search = nullptr;
    bool bound_inflight = false;
    bool bound_no_wait = false;  // kao_solve's deterministic schedule: no event wait on the search stream before a K-bound launch
    int bound_iters_last = 0;
    double bound_ms_last = 0;
    unsigned long long *d_keys = nullptr;
    unsigned long long *d_keys_glob = nullptr;  // receive buffer of the cross-GPU min-allreduce (kao_solve_multi)
    int32_t *d_drift = nullptr;
    int32_t *d_win_viol = nullptr;
    uint16_t *d_win_assign = nullptr;
    std::vector<unsigned char> h_readback;
    hipStream_t stream = nullptr;
    uint32_t launch = 0;
    uint32_t gen = 0;            // generation of the population (kao_session_new_generation)
    bool reinit = false;         // the next step re-initialises every restart (first launch of a new generation)
    size_t best_bytes = 0;       // size of the snapshot pool d_best
    // profiling
    std::vector<hipEvent_t> ev;  // triples
    int ev_pending = 0;
    double ms_search = 0, ms_eval = 0;
    uint64_t eval_bytes_per_launch = 0;
    uint64_t delta_total = 0, search_bytes_total = 0;
};

namespace kao {
// the topic's winning assignment (dense [P*RF]) as of the last finished launch
int session_topic_best(kao_session *s, int i, uint16_t *out);
// every restart's best feasible objective (-1 = none yet) / one restart's best snapshot, as of the last finished launch
int session_restart_objs(kao_session *s, int i, std::vector<int32_t> &objs);
int session_restart_best(kao_session *s, int i, int restart, uint16_t *out);
// an assignment found outside K-search (KAO-CX, another GPU) becomes the topic's incumbent and elite
int session_adopt_external(kao_session *s, int i, const uint16_t *assign, int64_t objective, uint64_t *key_out);
}  // namespace kao
