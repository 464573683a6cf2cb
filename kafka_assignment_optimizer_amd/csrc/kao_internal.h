// kao_internal.h -- structures shared by the host side (kao_model.cpp, kao_session.cpp, kao_solve.cpp; see kao_host.h) and the gfx950 kernels
// (kao_kernels.hip).  Not part of the C ABI.
#pragma once
#include <stdint.h>

#include <unordered_map>

#include "../../include/kao.h"

namespace kao {

constexpr int kRFP = 4;          // replica words per partition of the common case (RF <= 4); topics with 5..8 replicas use 8 (kMaxRF)
constexpr int kMaxRF = 8;
constexpr int kWaves = 4;        // wavefronts per K-eval workgroup; K-search uses 4, 2 or 1 (largest that fits LDS)
constexpr int kTeamMax = 8;      // wavefronts of a K-search team (k_team: several wavefronts on ONE restart, topics in global memory)
constexpr int kMaxRacks = 255;   // rack ids are u8, 0xFF marks a padding slot
constexpr int kRackTab = 256;    // entries of the per-rack LDS tables (rack sizes, K, RT)
constexpr uint32_t kNoneW = 0xFFFFFFFFu;  // empty slot in the LDS word layout (x | rack << 16)
constexpr uint32_t kKeyNull = 0xFFFFFFFFu;
constexpr int kDBias = 32768;
constexpr int kDualLog2 = 16;
constexpr int kDualScale = 1 << kDualLog2; // fixed point of the dual multipliers (K-bound): 65536 = 1.  (4096 until round 2: Polyak steps
                                          // below one unit of the last place truncate to zero and the iterate freezes -- 300 x 2000: stuck at
                                          // 14831.74 for 110,000 iterations; with 2^16 the same iteration reaches the LP optimum 14826.0)
constexpr int kDualStage = 100;           // level control: iterations per stage (K-bound)
constexpr int kDualDeflP = 2048;          // topics with more partitions use the long deflection memory (kao_bound.hip::db_defl)
constexpr int kScanTwoSlots = 24576;     // REPLACE scan: two tournament slots per iteration up to this many replica slots, one beyond (oracle/kao_port.c SCAN_TWO_SLOTS)
constexpr int kDualClamp = 1 << 26;       // |multiplier| <= this (32-bit headroom of the priced values)
constexpr int kDualQuarterLog2 = kDualLog2 - 2;  // kDualScale / 4: the quarter grid of the rounding probes and the search prices
constexpr int kDualProbes = 2;            // probes per K-bound launch: multipliers rounded to the quarter grid, then the half grid
constexpr uint32_t kExternalRestart = 0xFFFFFu;  // restart id of a best key adopted from another GPU (kao_solve_multi)
constexpr uint32_t kObjCap = 0xFFFFFFu;   // packed best key: viol(20) << 44 | (kObjCap - obj) << 20 | restart(20)

// Device-side descriptor of one topic.  Read once per workgroup (wave-uniform -> SGPRs).
struct TopicDev {
    // ---- internal (rack-major padded) index space, used by K-search -------------------
    int32_t P, RF, R, m, Bx;     // m = largest rack; x = rack*m + j, j < rack_size[rack]
    uint32_t magic;              // floor(2^32/m)+1 : rack(x) = mulhi(x, magic), exact for x < 2^16
    int32_t rep_lo, rep_hi, lead_lo, lead_hi, rack_lo, rack_hi, prack_lo, prack_hi;
    int32_t w00, w01, w10, w11;  // w[cur_role][new_role]
    uint32_t seed_lo, seed_hi;   // per-topic seed
    int32_t n_restarts;          // restarts (wavefronts) searching this topic
    int32_t restart_base;        // index of restart 0 in the per-restart arrays
    uint32_t cur_off;            // cur_pool   : first WORD of the topic (nw words x | rack << 16 per partition, 16-byte aligned)
    uint32_t ext_off;            // ext_pool   : internal -> dense (u16[Bx])
    uint32_t rsz_off;            // rsz_pool   : rack sizes (int32[R])
    uint64_t state_off;          // state_pool : BYTE offset of restart 0 (uint2 per partition, or uint4 words when the
                                 //              topic runs with its assignment in global memory)
    uint64_t best_off;           // best_pool  : first u16 of restart 0's dense snapshot ([P*RF] per restart)
    // ---- dense index space, used by K-eval ---------------------------------------------
    int32_t B, rf_cur;
    uint32_t rackof_off;         // rackof_pool: u8[B]
    uint32_t curd_off;           // curd_pool  : u16[P*rf_cur] dense current assignment
    uint32_t win_off;            // winners    : first u16 of this topic's winning assignment ([P*RF])
    uint32_t dual_off;           // dual_pool  : first int32 of a[B] l[B] da[B] dl[B] g[kRackTab] dg[kRackTab] lv[8] ra[B] rl[B] rg[kRackTab]
    int32_t period_log2;         // penalty sawtooth: restart rho has period 2^(period_log2 + (rho & 3)) iterations
    uint32_t price_off;          // price_pool : first int32 of this topic's search prices pa[B] pl[B] pg[kRackTab] (fixed point
                                 //              kDualScale, dense broker index); K-bound's epilogue or the host writes them
    uint32_t int_off;            // int_pool   : dense -> internal broker index (u16[B]); elite re-seeding reads dense snapshots
    int32_t nw;                  // replica words per partition in K-search / K-canon: 4 (RF, current RF <= 4) or 8
    int32_t has_bw;              // 1 = the topic carries broker weights (kao_topic.broker_w / broker_wl)
    uint32_t bw_off;             // bw_pool  : packed bw | bwl << 16 per INTERNAL index (u32[Bx])
    uint32_t bwd_off;            // bwd_pool : the same per DENSE index (u32[B]) for K-eval
    uint32_t cnt_off;            // cnt_pool : NR[B] NL[B] NK[kRackTab] of the sliced K-bound (k_bound_step), zero between steps; x 3 (the
                                 //            rotating buffers of k_bound_multi), then its shadow area a[B] l[B] g[kRackTab] ra[B] rl[B] rg[kRackTab]
    uint32_t hole_off;           // cur_pool   : first word of the topic's hole list {n leader holes, n follower holes, partitions with a leader
                                 //              hole ascending, partitions with a follower hole ascending} (round 4: the init walks it instead of
                                 //              inspecting every partition twice)
    int32_t pad_[1];
};

struct SearchParams {
    int32_t obj_scale, lam_min, lam_max;         // (the sawtooth period is per topic: TopicDev::period_log2)
    int32_t maxR;                // largest rack count of the launch group (sizes the per-rack LDS tables)
    int32_t wide;                // 1 = the group holds a topic of >= 512 replica slots (k_search<..., kWide = true>)
    uint32_t launch;             // launch number (global iteration = launch*iters + i)
    uint32_t iters;
    int32_t init;                // 1 = build the initial state of every restart first; 2 = K-init has built it (launch_init), the rest of an initialising launch remains
    int32_t maxP, maxBx;         // LDS carve sizes
    int32_t bw;                  // 1 = topics of this launch carry broker weights (priced instantiation: the weight table is carved)
    uint32_t gen;                // generation of the population: salts the tie-break hash of the initial state (init = 1 with gen > 0 =
                                 //     kao_solve re-initialises every restart after a population converged without a proof)
    int32_t scan2_max;           // REPLACE scan covers the tournament's TWO best slots on topics of at most this many replica slots (kScanTwoSlots)
    int32_t cur_global;          // 1 = the launch group keeps only the working assignment in LDS (k_search_curg): the current assignment is read from global memory
    int32_t elite;               // 1 = restarts that trail their topic's best feasible objective may re-seed from it (KAO-LS
                                 //     "elite" rule)
};

struct SearchPools {
    const TopicDev *topics;
    const int2 *block_map;       // per workgroup: {topic, first restart}
    const uint32_t *cur_pool;
    const uint16_t *ext_pool;
    const int32_t *rsz_pool;
    unsigned char *state_pool;
    uint16_t *best_pool;
    int32_t *restart_info;       // [n_restarts_total][4] = {best_obj, V, obj, accepted}
    int32_t *drift;              // [1] counter
    const int32_t *price_pool;   // search prices (k_search<*, true> only), see TopicDev::price_off
    const uint32_t *bw_pool;     // broker weights per internal index (k_search<*, true> only)
    const uint16_t *int_pool;    // dense -> internal broker index per topic
    const uint16_t *elite_assign;// [sum P*RF] every topic's best feasible assignment so far (dense; k_gather's output), at win_off
    const unsigned long long *elite_key;  // [n_topics] packed key of that assignment as of the previous step (~0 = none)
};

struct EvalPools {
    const TopicDev *topics;
    const int4 *block_map;       // per workgroup: {topic, first candidate (within topic), n candidates, out base}
    const uint8_t *rackof_pool;
    const uint16_t *curd_pool;
    const uint16_t *cand;        // candidates; topic t's candidate c at cand + best_off(t) + c*P*RF
    int32_t *objective;          // [n] or nullptr
    int32_t *violations;         // [n*8] or nullptr
    unsigned long long *best_key;// [n_topics] or nullptr (atomicMin of the packed key)
    int32_t maxP, maxB;
    int32_t cur_in_lds;          // 1 = stage the current assignment in LDS (fits); 0 = read it from global memory
    int32_t coop;                // 1 = one candidate per workgroup, its wavefronts cooperating (k_eval<NE, true>): few large candidates
    const uint32_t *bwd_pool;    // broker weights per dense index (topics with has_bw)
    int32_t *overflow;           // [1] or nullptr: set when a candidate puts more than 65,535 replicas on one broker (the 16-bit
                                 //     halves of the per-broker counters would carry; only possible when P*RF > 65535)
    int32_t rf_uniform;          // the replication factor every topic of the launch has, or 0 (mixed): RF 3 runs an instantiation whose slot loops have no guards
};

struct BoundPools {
    const TopicDev *topics;
    const int32_t *ids;          // per workgroup: topic
    const uint8_t *rackof_pool;
    const uint16_t *curd_pool;
    const uint16_t *ext_pool;    // rack-major internal index -> dense broker (members of every rack)
    const int32_t *rsz_pool;     // rack sizes
    int32_t *dual_pool;          // multipliers and previous directions, see TopicDev::dual_off
    const long long *target;     // [n_topics] incumbent objective the Polyak step aims at
    long long *best_L;           // [n_topics] smallest dual value so far (fixed point, kDualScale)
    int32_t *info;               // [n_topics][4] = {iterations so far, flags of the last launch, -, -}
    int32_t iters;               // iterations this launch
    int32_t maxB, maxP, maxR;    // LDS carve sizes
    int32_t cur_in_lds;          // 1 = the current assignment (8 B per partition) is staged in LDS too
    int32_t *price_pool;         // K-bound's epilogue exports the multipliers, rounded to the quarter grid, as search prices
    int32_t export_prices;       // 1 = the multipliers of the record dual value, 2 = the last iterate, 0 = no export
    int32_t ne;                  // 4 or 8: replica slots per partition the launch is instantiated for (k_bound<NE>)
    const uint32_t *bwd_pool;    // broker weights per dense index (TopicDev::bwd_off), nullptr when no topic of the session has any
};

// K-bound on several workgroups per topic (k_bound_step: one iteration per launch, partitions sliced over workgroups)
struct BoundWide {
    const int2 *map;             // per workgroup: {topic, slice}
    int32_t *cnt_pool;           // subproblem counts meeting in HBM, see TopicDev::cnt_off
    long long *ctl;              // [n_topics][16] control block: value sum, ticket, bad, stop | k_bound_multi's half (see kao_bound.hip)
    int32_t chunk;               // partitions per slice (a multiple of 64)
};

// per-rack LDS tables of K-search (rack sizes, K, RT): entries for `maxR` racks plus one for the padding marker, rounded to 64
constexpr int search_rack_tab(int maxR) { return ((maxR < 1 ? 1 : maxR) + 1 + 63) & ~63; }
// `team` > 0: the workgroup is a team of that many wavefronts on ONE restart (k_team; topics in global memory), `waves` is ignored
// cur_global: only the working assignment in LDS, the current assignment read from global memory (k_search_curg, round 5)
size_t search_lds_bytes(int maxP, int maxBx, int waves, bool global_a, bool priced = false, int nw = 4, bool bw = false, int maxR = kRackTab - 1, int team = 0, bool cur_global = false);
size_t eval_lds_bytes(int maxP, int maxB, bool cur_in_lds, int ne = 4);
void launch_search(const SearchPools &pools, const SearchParams &prm, int n_blocks, int waves, bool global_a, bool priced, int nw, void *stream, int team = 0);
bool launch_init(const SearchPools &pools, const SearchParams &prm, int n_blocks, int per_block, bool priced, int nw, void *stream);   // K-init (topics in global memory)
void launch_eval(const EvalPools &pools, int n_blocks, int ne, void *stream);
// copy every topic's winning snapshot (restart id in its packed key) and violation row into contiguous
// read-back buffers: one D2H instead of two per topic
void launch_gather(const TopicDev *topics, int n_topics, const unsigned long long *keys, const uint16_t *best_pool,
                   const int32_t *viol, uint16_t *win_assign, int32_t *win_viol, void *stream);

void launch_adopt_global(unsigned long long *keys, const unsigned long long *glob, int n, void *stream);

// K-bound: Lagrangian dual bound, one workgroup (`waves` wavefronts) per listed topic
size_t bound_lds_bytes(int maxB, int maxP, int maxR, bool cur_in_lds, int ne = 4, bool hbw = false, bool dirs = false);
void launch_bound_center(const BoundPools &pools, int n_topics, void *stream);   // the common shifts, in front of every K-bound launch
void launch_bound(const BoundPools &pools, int n_blocks, int waves, void *stream);
// the same iteration as a sequence of launches: begin, pools.iters x step, the probes, finish (n_blocks = all slices of the
// n_topics listed topics)
void launch_bound_wide(const BoundPools &pools, const BoundWide &wide, int n_topics, int n_blocks, int waves, void *stream);
bool launch_bound_multi(const BoundPools &pools, const BoundWide &wide, int n_topics, int n_blocks, int waves, void *stream);

// canonical tie-break on the device (kao_canonicalize): one wavefront, assignment words in global memory
size_t canon_lds_bytes(int maxBx);
void launch_canon(const TopicDev *topic, const uint32_t *cur_words, const uint16_t *ext, const int32_t *rsz, uint32_t *A, int maxBx,
                  int nw, int32_t *status, void *stream);


// ---- KAO-LP (kao_lp.hip): interior-point solve of the compact LP relaxation; its row duals are multipliers for K-bound ----
struct LpCtx;
// One LP over several devices (round 6): every shard holds a contiguous range of the topic's partitions; the coupling rows and the global
// variables are replicated.  The sums over the partitions meet through `coll` at fixed points of the iteration (the Schur complement once
// per factorisation, the coupling right-hand side once per solve, the scalar records of the reductions): every shard calls allreduce with
// its own rank, buffer and stream; the call returns once the collective is ENQUEUED on every shard's stream (kao_solve.cpp: RCCL, or the
// loop-back table on logical shards).  Sums are taken in rank order, so all shards hold the same bits afterwards.
struct LpColl {
    virtual int allreduce(int rank, double *buf, size_t n, bool is_min, void *stream) = 0;
    virtual ~LpColl() {}
};
struct LpShard { int p0, p1, rank; LpColl *coll; };
// A set of shard contexts driven as ONE solve (kao_solve.cpp LpFanImpl): lp_open_fan returns shard 0's context, whose lp_begin /
// lp_enqueue_mark / lp_poll_mark / lp_finish / lp_primal / lp_abort / lp_close forward here -- the caller (kao_solve's loop) sees one LP.
struct LpFan {
    virtual int begin(double tol, int maxit, double pert, uint32_t salt) = 0;
    virtual int enqueue_mark(int k, int slot) = 0;
    virtual int poll_mark(int slot, int *status, int *iterations, double deadline) = 0;
    virtual int finish(int32_t *multipliers, double stats[8], double *trace) = 0;
    virtual int primal(uint8_t *q, int32_t *zq) = 0;
    virtual void abort() = 0;
    virtual void close() = 0;        // closes every shard (the forwarding context included) and the fan itself
    virtual ~LpFan() {}
};
extern thread_local bool t_lp_inner;   // set while the fan itself calls the lp_* functions on its shards (no forwarding then)
void lp_set_fan(LpCtx *c, LpFan *fan);
int lp_open_fan(const kao_topic *t, const int *devices, int n_dev, LpCtx **out);   // kao_solve.cpp
int lp_open(const kao_topic *t, LpCtx **out, const LpShard *shard = nullptr);
// multipliers (host): a[B] l[B] g[R] in K-bound's fixed point; stats[8], trace: see kao_lp.hip
int lp_solve(LpCtx *c, double tol, int maxit, int32_t *multipliers, double stats[8], double *trace, double pert = 0.0, uint32_t salt = 0);   // one shot
// incremental (kao_solve runs the iterations beside K-search): begin, enqueue k iterations (asynchronous, no host round trip inside),
// poll (waits; *status 0 = running, 1 converged, 2 iteration limit, 3 stalled), finish (multipliers of the last finite iterate)
int lp_begin(LpCtx *c, double tol, int maxit, double pert = 0.0, uint32_t salt = 0);   // pert > 0: costs + pert * hash(variable, salt) (the primal side)
int lp_enqueue(LpCtx *c, int k);
int lp_poll(LpCtx *c, int *status, int *iterations);
int lp_enqueue_mark(LpCtx *c, int k, int slot);                      // enqueue + a mark (ring slot 0..31) that lp_poll_mark waits for
int lp_poll_mark(LpCtx *c, int slot, int *status, int *iterations, double deadline = 0);   // deadline (now_s() clock, 0 = none): past it the solve is aborted instead of waited for (*status = 4)
int lp_finish(LpCtx *c, int32_t *multipliers, double stats[8], double *trace);
inline double lp_default_pert(const kao_topic *t) { const double e = 100.0 / ((double)t->n_partitions * t->rf); return e < 1e-2 ? e : 1e-2; }   // oracle/kao_lp.py default_pert
int lp_primal(LpCtx *c, uint8_t *q, int32_t *zq);   // quantised primal iterate: q[(2 rf_cur + 2 R) * P] centi-units, zq[2 B] inflows (host memory)
// q / zq -> an assignment (kao_round.cpp; specification oracle/kao_lp.py round_primal).  fallback (dense indices, may be null): the rows
// fractional partitions keep.  rep[4] = {fractional partitions, over-inflow placements, unplaced, rows taken from fallback}
// max_free: without a fallback, more fractional partitions than this are not completed at all (the completion costs ~0.1 ms a partition):
// rep[0] = their number, rep[3] = -1, `out` is left incomplete
int lp_round_assignment(const kao_topic *t, const uint8_t *q, const int32_t *zq, const uint16_t *fallback, uint16_t *out, int32_t rep[4], int max_free = 1 << 30);
void lp_abort(LpCtx *c);   // stop flag up from the host: enqueued iterations turn into no-ops
void lp_close(LpCtx *c);
// the dense piece (kao_chol.hip): Cholesky of the n x n matrix in the lower triangle of S (row-major, n a multiple of 64, diag0 its
// diagonal) -> L below, L^T above, Linv[n / 64][64][64] the inverses of the diagonal tiles; then S x = r in place (xz[2 n]: scratch).
// stop / gate: device scalars (may be null): every kernel returns at once when *stop != 0 / when *gate == 0.  `stream` is a hipStream_t.
void chol_enqueue(void *stream, const double *stop, double *S, int n, const double *diag0, double *Linv);
void trsv_enqueue(void *stream, double *stop, const double *gate, const double *S, int n, double *r, const double *Linv, double *xz);

// ---- KAO-CX (kao_cycle.hip): cyclic-exchange improvement of a feasible assignment ----
bool cycle_supported(const kao_topic *t);
int cycle_improve(const kao_topic *t, uint16_t *assign, int32_t max_rounds, double deadline, int64_t *objective, int32_t stats[8]);
// the same with the device buffers kept between calls (kao_solve); the topic must outlive the context
struct CycleCtx;
CycleCtx *cycle_open(const kao_topic *t, int *rc_out);
// `poll` (may be null) is called between rounds: the caller keeps its other streams busy (kao_solve: K-bound); `pairs`: at a fixpoint
// of the plain layers the compound-edge layer may take a round (only with KAO_CX_PAIRS=1; kao_solve allows it for the elite's
// descents, not for the further starting points: a pass costs 0.15 s of host time at 300 x 2000)
int cycle_run(CycleCtx *c, uint16_t *assign, int32_t max_rounds, double deadline, int64_t *objective, int32_t stats[8],
              int (*poll)(void *) = nullptr, void *poll_arg = nullptr, bool pairs = true);
void cycle_close(CycleCtx *c);
// helpers of the host side (kao_solve.cpp) for the device translation units
// compound edges of leader-balanced pairs (kao_pairs.cpp; host only): the cheapest pair behind every edge x -> z
struct PairEdge { int32_t cost; int32_t set; int32_t p, q; uint16_t rowp[8], rowq[8]; };
int pair_edges(const kao_topic *t, const uint16_t *assignment, int gmin, std::unordered_map<uint32_t, PairEdge> &edges, int64_t stats[4]);
int api_fail(int code, const char *msg);
int api_require_init();
double api_now_s();

}  // namespace kao
