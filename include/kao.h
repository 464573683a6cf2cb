/* kao.h -- C ABI of libkao.so, the MI355X (gfx950) solver for the Kafka partition-assignment
 * 0-1 model of killerwhile/kafka-assignment-optimizer.
 *
 * What this boundary replaces.  The reference snapshot (/root/reference = README.md + one
 * image) contains no code, hence no FFI declaration to copy.  Its only solver seam is
 * "lp_solve is used behind the scene to solve the generated linear equation"
 * (README.md:135-136): the generated 0-1 model (README.md:144-185) goes in, one 0/1 value per
 * variable plus the objective comes out.  libkao.so replaces exactly that seam; each entry
 * point below cites the README lines it stands in for.  The structured instance (kao_topic)
 * is the compact form of the same model: variable t<T>b<B>p<P>[_l] (README.md:146,
 * README.md:182-184) == "broker B holds a replica of partition P of topic T [as leader]"
 * == assignment[P*rf + k] == B with k == 0 for the `_l` variable.
 *
 * Conventions: plain C, little-endian integers, caller owns every buffer, the library never
 * frees caller memory, no exceptions cross the ABI, return 0 = success / negative = error
 * (kao_strerror).  One process drives one GPU (kao_init(device)); sessions are independent.
 * Every compute entry point runs on the GPU; there is no CPU fallback -- without a usable
 * device the calls fail with KAO_ERR_NO_DEVICE.
 */
#ifndef KAO_H
#define KAO_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KAO_VERSION 103 /* 0.1.3: kao_lp_round (KAO-LP's primal side); 0.1.2: kao_lp_bound, kao_session_set_dual_state (KAO-LP, round 5) */
#define KAO_NONE 0xFFFFu /* "no broker": replica on a broker outside the target set / empty slot */
#define KAO_MAX_RF 8     /* replica slots per partition supported by the gfx950 kernels (RF <= 4: one 128-bit word group per
                            partition; 5..8: two) */
#define KAO_MAX_RACKS 255

enum {
    KAO_OK = 0,
    KAO_ERR_INVALID = -1,     /* bad argument */
    KAO_ERR_UNSUPPORTED = -2, /* RF > KAO_MAX_RF, racks > KAO_MAX_RACKS, broker tables too large for LDS */
    KAO_ERR_NO_DEVICE = -3,   /* no gfx950 device / HIP runtime failure at init */
    KAO_ERR_HIP = -4,         /* HIP runtime error (message via kao_last_error) */
    KAO_ERR_NOMEM = -5,
    KAO_ERR_NOT_INIT = -6
};

enum { /* kao_result.status */
    KAO_STATUS_OPTIMAL_PROVEN = 0,    /* feasible and objective == upper_bound */
    KAO_STATUS_FEASIBLE_BOUND_GAP = 1,/* feasible, objective < upper_bound (bound not tight or not optimal) */
    KAO_STATUS_NO_FEASIBLE = 2,       /* search found no feasible assignment (lp_solve would say "infeasible"
                                         only if truly so; this is not a proof) */
    KAO_STATUS_TIME_LIMIT = 3,        /* feasible, stopped by the time limit before target_objective */
    KAO_STATUS_INFEASIBLE_PROVEN = 4  /* a counting argument (kao_check_infeasible) shows that no assignment satisfies
                                         the rows: what lp_solve reports as "This problem is infeasible" */
};

/* One topic's sub-problem.  Topics are independent in the README model: every variable and
 * row carries the topic prefix t1... (README.md:146-184). */
typedef struct kao_topic {
    int32_t n_brokers;        /* B: size of the TARGET broker set (--broker-list, README.md:48) */
    int32_t n_racks;          /* R <= KAO_MAX_RACKS */
    int32_t n_partitions;     /* P */
    int32_t rf;               /* target replication factor (README.md:148-151), <= KAO_MAX_RF */
    int32_t rf_cur;           /* replication factor of `current` (README.md:9: RF may change) */
    const uint8_t *rack_of;   /* [B] dense rack index of each target broker (README.md:27-29) */
    const uint16_t *current;  /* [P*rf_cur] dense broker index, slot 0 = preferred leader
                                 (README.md:52-63); KAO_NONE = broker not in the target set; a broker may not
                                 appear twice in one partition (KAO_ERR_INVALID) */
    int32_t w[2][2];          /* objective weights w[cur_role][new_role], role 0 leader, 1 follower
                                 (README.md:145-146); default {{4,1},{2,2}}; each 0..1023 and
                                 n_partitions * rf * largest weight < 2^24 */
    /* band right-hand sides; -1 = derive floor/ceil of the average (README.md:159-160,
       164-165, 174-175, 179) */
    int32_t rep_lo, rep_hi;     /* C3 replicas per broker   (README.md:158-161) */
    int32_t lead_lo, lead_hi;   /* C4 leaders per broker    (README.md:163-166) */
    int32_t rack_lo, rack_hi;   /* C6 replicas per rack     (README.md:173-176) */
    int32_t prack_lo, prack_hi; /* C7 replicas per partition per rack (README.md:178-180) */
    /* Optional broker weights (NULL = none): extra objective coefficients on EVERY variable of a broker -- broker_w[b] on
       each replica placed on b (t?b<b>p? and t?b<b>p?_l), broker_wl[b] on each leader (t?b<b>p?_l) -- i.e. plain
       coefficients of the README's `max:` row (README.md:145-146).  0..1023 each.  kao_solve_capped uses them to price
       cluster-wide per-broker caps; K-bound and KAO-CX take them into their priced values (round 3); weighted topics are
       not canonicalised. */
    const int32_t *broker_w;    /* [n_brokers] or NULL */
    const int32_t *broker_wl;   /* [n_brokers] or NULL */
} kao_topic;

typedef struct kao_opts {
    uint64_t seed;            /* search is deterministic in (seed, restarts, iters_per_launch) */
    double time_limit_s;      /* kao_solve: wall-clock limit; <= 0 = default 10 s */
    int32_t restarts;         /* parallel restarts (wavefronts) per topic; <= 0 = auto: one round of resident wavefronts
                                 over all topics, fewer for very large topics (depth over breadth): a session takes four per
                                 compute unit on a topic whose assignment lives in HBM (>= 32,768 replica slots; that fills
                                 every SIMD with a wavefront waiting on global loads), kao_solve one per compute unit */
    int32_t iters_per_launch; /* local-search iterations per K-search launch; <= 0 = 512 (sessions) / 128 (kao_solve) */
    int32_t max_launches;     /* kao_solve: stop after this many launches; <= 0 = unlimited */
    int32_t obj_scale;        /* S in cost = lam*violation - S*objective; <= 0 = 4 */
    int32_t lam_min, lam_max; /* penalty sawtooth bounds; <= 0 = 1 / 40 */
    int32_t period_log2;      /* sawtooth period = 2^(period_log2 + (restart & 3)); <= 0 = per topic by size:
                                 floor(log2(2 * n_partitions * rf)) clamped to 8..16 */
    int32_t stop_at_bound;    /* kao_solve: 1 = stop as soon as every topic is OPTIMAL_PROVEN */
    int32_t profile;          /* 1 = bracket every kernel with HIP events (kao_session_stats) */
    int32_t dual_iters;       /* kao_solve: K-bound (Lagrangian dual bound) iterations per launch for topics whose
                                 feasible incumbent is below the bound; 0 = 128 for the first launch, then adapted to
                                 about 10 ms per launch; < 0 = never run K-bound */
    int32_t elite_period;     /* every elite_period-th K-search launch, restarts that trail their topic's best feasible
                                 objective re-seed from that assignment with probability 1/2 ("go with the winners");
                                 0 = kao_solve: about one penalty period of the largest topic, sessions: never; < 0 = never */
    int32_t use_prices;       /* kao_solve: feed K-bound's multipliers (rounded to quarters) back into K-search as Lagrangian
                                 prices of the broker / rack rows (augmented-Lagrangian search); 0 = yes, < 0 = no */
    int32_t use_cycles;       /* kao_solve: KAO-CX (kao_improve_cycles) on incumbents the search has stopped improving;
                                 0 = yes, < 0 = no */
    int32_t islands;          /* kao_solve, experimental: > 1 = search every topic as this many independent copies (own seed,
                                 restarts, elite, K-bound trajectory, KAO-CX) that share one copy's restart budget and answer
                                 with the best; certificates are shared.  0 / 1 = off (measured: no gain) */
    int32_t schedule;         /* kao_solve / kao_solve_multi: 0 = DETERMINISTIC (default): every decision of the solve loop -- which
                                 incumbents K-bound is aimed at and for how many iterations, when its certificates and prices are
                                 merged, when KAO-CX runs and for how many rounds, when elites are exchanged -- is keyed to counts
                                 of launches / iterations / rounds, never to the clock, so the same (instance, options, seed, device
                                 size) gives the same answer whatever the launch timing; the clock (time_limit_s) only decides when
                                 to stop.  1 = wall-clock adaptive (round-2 behaviour: K-bound merged whenever it has finished,
                                 launch lengths adapted to measured times, KAO-CX in time slices): a few percent more work per
                                 second, answers on large topics vary by a unit or two between runs */
    int32_t team;             /* topics whose assignment lives in global memory (beyond ~9,000 partitions): wavefronts that search
                                 ONE restart together (k_team: every wavefront proposes a move per iteration against the same
                                 state, proposals that share no partition / broker / rack with a lower-numbered wavefront's are
                                 all applied).  0 / 1 = one wavefront per restart (k_search; the default: measured on 1000 x 30000 and
                                 1000 x 100000, teams bought neither iterations per second nor a better incumbent in 3 s),
                                 2..8 = team size (4 at most for RF 5..8).  Deterministic either way; other values are KAO_ERR_INVALID.  (Was reserved0.) */
    const int64_t *target_objective; /* kao_solve: optional [n_topics]; a topic counts as done once its feasible
                                        objective reaches this value (e.g. a known optimum); NULL = use the bound */
} kao_opts;

typedef struct kao_result {
    int32_t status;            /* KAO_STATUS_* */
    int32_t best_restart;      /* which restart produced the answer (-1: adopted from another GPU, kao_solve_multi) */
    int64_t objective;         /* value of the README objective (README.md:145-146) */
    int64_t upper_bound;       /* min(closed-form bound (kao_upper_bound), K-bound dual certificate) */
    int32_t violations[8];     /* [0] total, [1..7] = C1..C7 magnitudes of the returned assignment */
    double seconds_to_best;    /* wall time from entry to the launch that produced `objective` */
    uint16_t *assignment;      /* [P*rf] caller-allocated; dense broker index, slot 0 = leader */
} kao_result;

typedef struct kao_stats {
    uint64_t launches;          /* K-search launches */
    uint64_t delta_candidates;  /* neighbours delta-evaluated by K-search, per restart and iteration: REPLACE = all B
                                   brokers of one slot (scan blocks; of TWO slots on topics of at most 24,576 replica slots --
                                   counted as 2 B although the second slot is scanned only when a second lane takes part in
                                   the tournament: an upper bound on small tournaments) or 64 lanes x 4 (sample blocks),
                                   EXCHANGE = all P*rf partner slots, LEADER-SWAP = 64 x (rf-1) */
    uint64_t full_candidates;   /* complete candidates fully evaluated by K-eval */
    double ms_search;           /* HIP-event time of K-search launches (profile=1) */
    double ms_eval;             /* HIP-event time of K-eval launches (profile=1) */
    uint64_t search_bytes_algo; /* algorithmic bytes of the K-search launches */
    uint64_t eval_bytes_algo;   /* algorithmic bytes of the K-eval launches */
    int32_t n_restarts_total;
    int32_t lds_bytes_search;   /* dynamic LDS per K-search workgroup (largest launch group) */
    int32_t blocks_search;      /* workgroups per K-search launch */
    int32_t drift;              /* restarts whose incrementally tracked (V, objective) disagreed with the
                                   from-scratch recount at the end of a launch; must be 0 */
    int32_t launch_groups;      /* topics are bucketed by LDS footprint; one K-search + K-eval launch per group */
    int32_t reserved;
} kao_stats;

typedef struct kao_session kao_session;
typedef struct kao_eval_plan kao_eval_plan;

/* ---- lifetime ------------------------------------------------------------------------- */
int kao_init(int device);            /* select the HIP device of this process; idempotent */
void kao_shutdown(void);
int kao_version(void);
const char *kao_strerror(int code);  /* static storage */
const char *kao_last_error(void);    /* thread-local detail of the last failure */
int kao_device_name(char *buf, int len);

/* ---- host-side model helpers (no device work) ------------------------------------------ */
/* Fill the derived band values (out[0..7] = rep_lo,rep_hi,lead_lo,lead_hi,rack_lo,rack_hi,
 * prack_lo,prack_hi), honouring overrides >= 0.  README.md:158-180. */
int kao_derive_bounds(const kao_topic *t, int32_t out[8]);
/* Upper bound on the objective: every partition keeps its best surviving replicas (coupling rows dropped),
 * minus the cheapest way to perform the evictions / leader changes the bands force (kao_model.cpp). */
int kao_upper_bound(const kao_topic *t, int64_t *ub);
/* Necessary conditions of the model checked by counting (band capacities per broker / rack / partition).  Returns 1
 * and a reason in `why` (may be NULL) if the topic is provably infeasible, 0 if no condition fails (it may still be
 * infeasible for subtler reasons), < 0 on error. */
int kao_check_infeasible(const kao_topic *t, char *why, int why_len);
/* Canonical tie-break among equal-objective feasible assignments (lowest broker index for newly
 * placed replicas, retained followers keep their order): reproduces README.md:88 `[8,1]`.
 * Runs on the GPU (k_canon: the REPLACE scan with "violation delta == 0" as the filter); any topic size. */
int kao_canonicalize(const kao_topic *t, uint16_t *assignment);

/* ---- K-eval: full evaluation of complete candidates (README.md:145-180 in one pass) ---- */
/* One candidate from host memory.  Stands in for "substitute the 0/1 vector into every row of
 * the generated model".  Per-broker counters are 16 + 16 bits (replicas | leaders): a candidate that puts more than
 * 65,535 replicas on one broker (possible only when P*RF > 65535) is reported as KAO_ERR_UNSUPPORTED, never mis-counted. */
int kao_evaluate(const kao_topic *t, const uint16_t *assignment, int64_t *objective, int32_t violations[8]);
/* n candidates [n][P*rf] from host memory; objective[n], violations[n*8]. */
int kao_evaluate_batch(const kao_topic *t, const uint16_t *candidates, int64_t n, int32_t *objective,
                       int32_t *violations);
/* Device-resident batches: tables uploaded once, candidates/outputs are DEVICE pointers. */
int kao_eval_plan_create(const kao_topic *t, kao_eval_plan **out);
int kao_eval_plan_run(kao_eval_plan *p, const void *d_candidates, int64_t n, void *d_objective /* int32[n] */,
                      void *d_violations /* int32[n*8] or NULL */, void *d_best_key /* uint64[1] or NULL */);
int kao_eval_plan_sync(kao_eval_plan *p, double *ms_last /* HIP-event ms of the last run, or NULL */);
void kao_eval_plan_destroy(kao_eval_plan *p);

/* ---- K-search sessions: resident parallel-restart local search ------------------------- */
/* Replaces lp_solve's solve() (README.md:135-136) for a batch of topics. */
int kao_session_create(const kao_topic *topics, int32_t n_topics, const kao_opts *opts, kao_session **out);
/* One step = one K-search launch (iters_per_launch iterations for every restart of every topic)
 * + one K-eval launch over every restart's best snapshot with the wavefront/block min-reduce
 * into one packed key per topic.  Asynchronous on the session's stream. */
int kao_session_step(kao_session *s);
int kao_session_sync(kao_session *s);
/* Starts a new GENERATION of the population: the next kao_session_step re-initialises every restart of every topic from the
 * current assignment (best insertion; the tie-break hash carries the generation number, so the new population differs from
 * the first), and the best snapshots and packed best keys of the old generation are dropped -- read kao_session_best first.
 * K-bound's state, the search prices and the launch counter carry on.  kao_solve does this when a population has converged
 * on an incumbent it cannot prove optimal (the incumbent itself is kept on the host); also a test hook of the bit-exact
 * replay (oracle/kao_port.c re-initialises the same way). */
int kao_session_new_generation(kao_session *s);
/* Copy back per-topic bests (results[n_topics], assignment buffers caller-allocated). */
int kao_session_best(kao_session *s, kao_result *results);
/* Per-topic packed best keys as the device holds them (uint64[n_topics]); the value a
 * min-allreduce across GPUs operates on: viol(20b) << 44 | (0xFFFFFF - objective) << 20 | restart. */
int kao_session_best_keys(kao_session *s, uint64_t *keys);
/* The same keys where they live: DEVICE pointer to uint64[n_topics] (valid for the session's lifetime; written by K-eval on
 * the session's stream -- kao_session_sync first).  What a one-process-per-GPU host hands to ncclAllReduce(..., ncclMin)
 * without a host round trip (kafka_assignment_optimizer_amd/multigpu.py::allreduce_best_resident). */
int kao_session_device_keys(kao_session *s, void **d_keys);
int kao_session_stats(kao_session *s, kao_stats *out);
/* K-bound: the optimality certificate beyond the closed-form bound (kao_upper_bound).  lp_solve proves optimality
 * by branch-and-bound over the LP relaxation (README.md:135-136); K-bound instead minimises the Lagrangian dual of
 * the same 0-1 model (README.md:144-185) on the device: rows C3, C4, C6 priced with integer multipliers, rows
 * C1, C2, C5, C7 kept in an exactly solved per-partition subproblem.  Every dual value is an upper bound on the
 * optimum, so floor(min dual) is a certificate whatever the multipliers.
 * One launch runs up to `iters` iterations for every topic i with target[i] >= 0 (the incumbent objective the
 * step length aims at; pass -1 to skip a topic).  Topics outside K-bound's limits (broker and rack tables
 * beyond 160 KiB of LDS: about 8,000 brokers; n_partitions*rf > 2^21 (round 6; 2^20 in rounds 4-5, 2^17 before), or n_partitions*rf*(largest weight) > 2^25; a weight outside 0..255) are skipped.  Asynchronous, on a stream of its own: K-bound
 * occupies one compute unit per topic -- one per 512 partitions when a launch holds a topic of more than 2,048 partitions
 * (then every iteration is a kernel launch of its own, see kao_bound.hip) -- and runs beside K-search (kao_session_step); a
 * new launch first waits for the
 * previous K-bound launch (it continues from the multipliers that one left in HBM). */
int kao_session_bound_step(kao_session *s, const int64_t *target, int32_t iters);
/* Search prices.  K-search can carry Lagrangian prices of the coupling rows in its move cost: delta = lam * dViolation
 * - S * dObjective + S * dPrice.  A row's multiplier (a[b]: replicas on broker b, l[b]: leaders on b, g[r]: replicas in rack
 * r) is charged for a unit that enters the row and refunded for one that leaves, but only where the row's count leaves or
 * re-enters its band -- exactly where the violation changes (inside a slack band a unit is free): broker-, rack- and
 * direction-specific penalty weights lam +- multiplier.  With near-optimal multipliers the chain steps an improvement
 * needs (objective down a little, excess moved to another broker) become neutral moves.
 * kao_session_set_prices: a[n_brokers], l[n_brokers], g[n_racks] of one topic from the host (fixed point, 65536 = 1; any
 * values are valid -- prices steer the search, they never change what is reported).  kao_session_adopt_prices: use what
 * the last finished K-bound launch exported (the multipliers of its record dual value, rounded to the quarter grid) for every topic it covered;
 * waits for the K-bound launch in flight.  Both take effect from the next kao_session_step. */
int kao_session_set_prices(kao_session *s, int32_t topic, const int32_t *a, const int32_t *l, const int32_t *g);
int kao_session_adopt_prices(kao_session *s);
/* Test hook: the prices K-search currently carries for one topic (zeros until set or adopted); outputs may be NULL. */
int kao_session_prices(kao_session *s, int32_t topic, int32_t *a, int32_t *l, int32_t *g);
/* 1 while a K-bound launch is still running, 0 when none is (its results can be read without waiting), < 0 = error. */
int kao_session_bound_busy(kao_session *s);
/* Wait for the K-bound launch in flight (if any) and read the certificates back: upper_bound[i] = min(closed-form bound, floor(best dual value));
 * flags[i]: 1 = the last K-bound launch closed the gap to its target, 2 = dual optimum reached (zero subgradient),
 * 4 = no bound (a partition subproblem is infeasible), 8 = topic outside K-bound's limits; iters[i] = K-bound
 * iterations so far.  Any output pointer may be NULL. */
int kao_session_bounds(kao_session *s, int64_t *upper_bound, int32_t *flags, int32_t *iters);
/* Test hook: K-bound state of one topic -- multipliers a[n_brokers], l[n_brokers], g[n_racks] (fixed point, 65536 = 1)
 * and the smallest dual value so far in the same fixed point (INT64_MAX-like before the first iteration). */
int kao_session_dual_state(kao_session *s, int32_t topic, int32_t *a, int32_t *l, int32_t *g, int64_t *best_dual);
/* Test hook / KAO-LP: overwrite K-bound's current multipliers of one topic (a[n_brokers], l[n_brokers], g[n_racks], fixed point
 * 65536 = 1, each clamped to +-2^26); the direction memory and the level control start afresh, the record dual value and the
 * certificate stay.  The next kao_session_bound_step evaluates the dual function there (after the common shifts) and goes on
 * from there.  Waits for a K-bound launch in flight. */
int kao_session_set_dual_state(kao_session *s, int32_t topic, const int32_t *a, const int32_t *l, const int32_t *g);
/* KAO-LP: the certificate from the model's LP relaxation (round 5).  lp_solve's proof of optimality rests on the LP relaxation of
 * the generated model (README.md:135-136, README.md:144-185); K-bound's subgradient iteration approaches that LP's value from
 * above and stalls short of it on slack-band and on large topics.  kao_lp_bound solves the LP itself on the device -- in compact
 * form (new placements pooled per partition and rack; oracle/kao_lp.py states the rows) by a block-structured interior-point
 * method (kao_lp.hip) -- takes its row duals as multipliers, rounds them to K-bound's fixed point and lets K-bound evaluate the
 * dual function there IN INTEGERS: *bound = floor(that value) is a valid upper bound on the optimum whatever the floating-point
 * solve did.  multipliers (may be NULL): a[n_brokers], l[n_brokers], g[n_racks]; stats (may be NULL): [0] interior-point
 * iterations, [1] README objective of the primal iterate, [2] of the dual iterate (the LP value to ~1e-7), [3] 0 converged /
 * 1 iteration limit / 3 stalled (the last finite iterate was used), [4] mu, [5] / [6] relative primal / dual infeasibility,
 * [7] milliseconds of the interior-point solve.  tol <= 0: 1e-7; max_iters <= 0: 80.  KAO_ERR_UNSUPPORTED: outside K-bound's
 * limits or more than ~4,700 brokers. */
int kao_lp_bound(const kao_topic *t, double tol, int32_t max_iters, int64_t *bound, int64_t *best_dual, int32_t *multipliers, double stats[8]);
/* Test hook: the interior-point solve alone, with its trace -- trace[5 * i .. 5 * i + 4] = (mu, primal objective, dual objective,
 * relative primal infeasibility, relative dual infeasibility) of iterate i in the LP's min form (README objective = -value),
 * room for max_iters + 2 iterates; stats and multipliers as kao_lp_bound (any may be NULL).  The parity tests hold the trace against
 * oracle/kao_lp_port.c's. */
int kao_lp_trace(const kao_topic *t, double tol, int32_t max_iters, double *trace, double stats[8], int32_t *multipliers);
/* KAO-LP, the primal side (round 5): an assignment from the LP relaxation.  lp_solve returns the optimum of the generated model
 * (README.md:135-136); on every topic measured the model's LP (README.md:144-185, all rows relaxed to [0, 1]) has INTEGRAL optimal
 * vertices that attain it -- but its optimal face is huge (ties everywhere) and an interior-point iterate ends at the face's
 * analytic centre, fractional in most partitions.  kao_lp_round perturbs the costs by pert * h(variable, salt), h in [0, 1) a
 * hash, which leaves (generically) one optimal vertex; the iterate converges to it, is rounded on the host (new replicas of a rack
 * handed to that rack's brokers by their inflows; specification oracle/kao_lp.py round_primal) and evaluated exactly by K-eval.
 * assignment [P*rf] (dense broker indices, leader first) is overwritten: on entry it may hold a FALLBACK -- the rows that partitions
 * with fractional variables keep (use_fallback != 0; e.g. an incumbent) --, otherwise such partitions are completed together (a search
 * over which current replicas they keep, DESIGN.md section 4b''; far from a vertex: their heaviest options).
 * The result can violate band rows when partitions were fractional: violations[8] as kao_evaluate (violations[0] = total).
 * pert <= 0: min(1e-2, 100 / (P * rf)); tol <= 0: 1e-8; max_iters <= 0: 150.  stats (may be NULL): [0] interior-point iterations,
 * [1] status (0 converged / 1 iteration limit / 3 stalled), [2] fractional partitions, [3] replicas placed beyond a broker's
 * inflow, [4] rows taken from the fallback, [5] milliseconds of the interior-point solve, [6] milliseconds of the rounding,
 * [7] the perturbation used.  An optimality proof needs kao_lp_bound's certificate beside it (objective == bound). */
int kao_lp_round(const kao_topic *t, double pert, uint32_t salt, double tol, int32_t max_iters, int32_t use_fallback, uint16_t *assignment,
                 int64_t *objective, int32_t violations[8], double stats[8]);
/* Test hook: the host half of kao_lp_round alone -- from a quantised iterate to an assignment (no device needed).  q[(2 rf_cur + 2 R) * P]:
 * centi-units min(250, rint(100 x)) of f_j (row j), l_j (row rf_cur + j), yf_r (row 2 rf_cur + r), yl_r (row 2 rf_cur + R + r), each row P
 * long; zq[2 B]: rint(zf_b), rint(zl_b).  assignment / use_fallback as kao_lp_round; rep[4] = {fractional partitions, placements beyond
 * an inflow, unplaced, rows taken from the fallback}.  use_fallback == 2: only the band repair at the end of the rounding, on the complete
 * assignment passed in (q, zq unused, may be NULL).  The parity tests hold it against oracle/kao_lp.py round_primal / repair_bands on the
 * scalar restatement's iterate. */
int kao_lp_round_host(const kao_topic *t, const uint8_t *q, const int32_t *zq, int32_t use_fallback, uint16_t *assignment, int32_t rep[4]);
/* Test hook: KAO-LP's dense kernels alone (kao_chol.hip) on the caller's symmetric positive definite matrix A[n][n] (row-major, the lower
 * triangle is read; n a multiple of 64, at most 10,240): factor[n][n] = L in the lower triangle and L^T tile-wise in the upper one, linv[n / 64]
 * [64][64] = the inverses of L's diagonal tiles, x[n] = the solution of A x = rhs (rhs NULL: all ones), ms[2] = HIP-event milliseconds of
 * the factorisation and of the two triangular solves (second of two runs).  Any output may be NULL. */
int kao_dense_spd_test(const double *A, int32_t n, const double *rhs, double *factor, double *linv, double *x, double ms[2]);
/* Test hook (round 6): the LP of ONE topic solved by n_dev SHARDS -- shard r holds the partitions [P r / n_dev, P (r + 1) / n_dev) with their
 * variables and local rows; the 3R + 2B coupling rows and the global variables are replicated; per interior-point iteration the shards'
 * parts of the Schur complement meet in one all-reduce (f64 sum), the coupling right-hand sides in one per solve, the scalar records of the
 * reductions in one each; the Cholesky and the triangular solves run replicated.  `devices` may repeat a device with KAO_RCCL_LOOPBACK=1
 * (logical shards through the loop-back table); distinct devices go through RCCL.  pert as kao_lp_round (0: its default; < 0: the model's
 * own LP).  *bound = the certificate (K-bound's integer dual value at the common row duals), assignment / objective / violations = the
 * rounded iterate as kao_lp_round (assignment may be NULL), stats[8] = {iterations, status, fractional partitions, collectives issued,
 * README objective of the dual iterate, milliseconds of the solve, milliseconds of certificate + rounding, perturbation}.
 * kao_solve_multi races whole solves on a replicated large topic by default and runs this sharded solve instead with KAO_MULTI_LP=shard;
 * unmeasured on more than one GPU. */
int kao_lp_sharded_test(const kao_topic *t, const int32_t *devices, int32_t n_dev, double pert, uint32_t salt, double tol, int32_t max_iters,
                        int64_t *bound, uint16_t *assignment, int64_t *objective, int32_t violations[8], double stats[8]);
/* One-shot K-bound on one topic: `launches` launches of `iters` iterations towards `target`.
 * *bound = floor(best dual / 65536) (not combined with kao_upper_bound); multipliers, if not NULL, receives
 * a[n_brokers], l[n_brokers], g[n_racks]. */
int kao_dual_bound(const kao_topic *t, int64_t target, int32_t iters, int32_t launches, int64_t *bound,
                   int64_t *best_dual, int32_t *iters_done, int32_t *flags, int32_t *multipliers);
/* Test hook: state of one restart -- final[P*rf], best[P*rf] (dense), info = {best_obj, V, obj, accepted}. */
int kao_session_restart_state(kao_session *s, int32_t topic, int32_t restart, uint16_t *final_state,
                              uint16_t *best_state, int32_t info[4]);
void kao_session_destroy(kao_session *s);

/* Whole job: create, step until target/time limit, read back, destroy.  What runs inside (round 5): K-search + K-eval on every topic;
 * K-bound beside them (the certificate of small topics); KAO-CX on stalled incumbents; and for every topic of at least 2,048 replica
 * slots that is not closed at once ONE interior-point solve of the model's LP with slightly perturbed costs (KAO-LP): its row duals
 * become the certificate (and the search prices), its iterate -- rounded as kao_lp_round does -- the incumbent; status OPTIMAL_PROVEN means
 * objective == certificate.  Topics from 32,768 slots get that solve before any K-search launch when time_limit_s >= 1 (huge ones: 1.5).
 * The schedule is keyed to counts, not to the clock (kao_opts.schedule = 0): same input, same seed, same answer. */
int kao_solve(const kao_topic *topics, int32_t n_topics, const kao_opts *opts, kao_result *results);
/* Whole job on several GPUs of ONE process (one host thread drives all of them; launches of every device are enqueued
 * before any is waited for).  devices[n_dev] = HIP device ordinals.
 *   n_topics >= n_dev: topics are independent sub-problems (README.md:146-184) and are dealt to the devices (longest
 *     processing time first by brokers x partitions); no exchange on the data path.
 *   n_topics <  n_dev: every device searches every topic with its own seed; every elite_period launches the packed best
 *     keys are min-allreduced on the device-resident buffers (ncclAllReduce, ncclUint64, ncclMin over xGMI) and each topic's
 *     winning assignment is broadcast (ncclBroadcast), so that trailing restarts on EVERY device re-seed from the global
 *     best; certificates are shared (any device's bound is valid).  K-bound runs on the first device only.
 * Listing a device twice gives logical shards on one device (the same flow through plain copies; for tests on one GPU).
 * librccl.so is loaded on first use.  Results as kao_solve. */
int kao_solve_multi(const kao_topic *topics, int32_t n_topics, const int32_t *devices, int32_t n_dev, const kao_opts *opts,
                    kao_result *results);
/* Cluster-wide per-broker load caps (BASELINE config 5; SURVEY.md section 8e "when it does NOT shard"): on top of every
 * topic's own rows, sum over ALL topics of the replicas on broker b <= replica_cap[b] (-1 = no cap for b).  All topics must
 * share one broker set (same n_brokers, same dense index).  The caps couple the topics; they are priced: every round solves
 * the topics independently (kao_solve / kao_solve_multi when n_dev > 1) with broker weights M - mu[b], adds up the broker
 * loads of the answers (the step that is an allreduce(SUM, int32[B]) when topics are sharded over processes) and raises mu[b]
 * on overloaded brokers (projected subgradient with a diminishing step; once a round respects every cap its plan is kept as
 * incumbent and the prices are relaxed again to look for a cheaper one).  results[i] = the best plan found that respects
 * the caps (status FEASIBLE_BOUND_GAP / NO_FEASIBLE; objective = README objective without the weights);
 * *lagrangian_bound (may be NULL) = smallest Lagrangian dual value seen (an upper bound on the capped optimum when every
 * topic's priced sub-problem was proven optimal in that round, else INT64_MAX).  devices / n_dev as kao_solve_multi
 * (NULL / 0 = the current device).  When the `max_rounds` price rounds end without a cap-respecting plan, up to max_rounds / 2 further
 * rounds only RAISE prices until a round in which every topic is feasible respects every cap. */
int kao_solve_capped(const kao_topic *topics, int32_t n_topics, const int32_t *replica_cap, const int32_t *devices, int32_t n_dev,
                     const kao_opts *opts, int32_t max_rounds, kao_result *results, int64_t *lagrangian_bound);
/* ---- KAO-CX: cyclic-exchange improvement of a feasible assignment --------------------------------
 * Stands in for nothing in the reference (lp_solve returns the exact optimum, README.md:135-136): it is the intensification
 * step that lets the device reach optima K-search's one- and two-slot moves do not -- on rigid instances (replicas per
 * broker fixed exactly) the last improvements are cyclic exchanges over 4..10 partitions.  From a FEASIBLE `assignment`
 * ([P*rf], overwritten by the improved one): transfer graphs on the brokers (follower moves, role swaps, band slack), cheapest
 * paths of <= 8 edges by three min-plus squarings, improving cycles and seed rows priced by their closures, candidates
 * unrolled into slot changes and scored exactly by K-eval; rounds until nothing improves or `max_rounds` (<= 0: no limit).
 * kao_solve calls it for unproven topics whose search has stalled.  stats (may be NULL): [0] rounds, [1] improving rounds,
 * [2] realisations evaluated, [3] improving ones, [4] candidates priced > 0, [5] compounds merged, [6] objective before,
 * [7] objective after.  Broker weights (kao_topic.broker_w / broker_wl) enter every edge and seed price.
 * KAO_ERR_UNSUPPORTED: rf < 2, rf > 8 or brokers + racks > 2047. */
int kao_improve_cycles(const kao_topic *t, uint16_t *assignment, int32_t max_rounds, int64_t *objective, int32_t stats[8]);
/* Parity hooks of KAO-CX (tests): the cost matrix of `layer` (0 = follower moves, 1 = role swaps) after `level` squarings
 * (0..3) as dist[n*n], n = B + R + 1 (nodes B .. B+R-1 = the racks' slack nodes, B+R = the global slack node; 1 << 17 = none),
 * the midpoints mid[n*n] (level >= 1; may be NULL) and the slot p*rf+k behind every level-0 edge slot[n*n] (0xFFFFFFFF = none;
 * may be NULL). */
int kao_cycle_matrices(const kao_topic *t, const uint16_t *assignment, int32_t layer, int32_t level, int32_t *dist, int32_t *mid,
                       uint32_t *slot);
/* The seed table table[P * n_cfg * 2] = (total, completing broker) per partition and configuration (oracle/kao_cycle.py
 * gives the numbering); *n_cfg is always set; table may be NULL to query n_cfg only. */
int kao_cycle_seeds(const kao_topic *t, const uint16_t *assignment, int32_t *table, int32_t *n_cfg);
/* Prototype hook of the next KAO-CX layer (host only, no GPU; specification: oracle/kao_cycle_pairs.py): the
 * COMPOUND EDGES of leader-balanced pairs of leader transfers around a feasible `assignment` -- partition p led by u hands the
 * leadership to v, partition q led by v hands it to u (half-moves of objective gain >= gmin each; a generic entering follower
 * may take the broker the partner releases), and a pair whose net replica effect is one unit x -> z is an edge of cost
 * -(gain of both rows).  cost[B*B] (row x, column z; INT32_MAX = none); stats (may be NULL): [0] half-moves, [1] pairs joined,
 * [2] edges, [3] pairs with no net replica effect and a positive gain.  Not used by kao_solve yet. */
int kao_cycle_pair_edges(const kao_topic *t, const uint16_t *assignment, int32_t gmin, int32_t *cost, int64_t stats[4]);

/* Diagnostic: runs the two collectives kao_solve_multi uses (ncclAllReduce(ncclUint64, ncclMin) and ncclBroadcast) on
 * small resident buffers of the listed distinct devices and checks the results.  0 = ok. */
int kao_rccl_selftest(const int32_t *devices, int32_t n_dev);
/* Test hook.  With KAO_RCCL_LOOPBACK=1 in the environment kao_solve_multi and kao_rccl_selftest take their collectives from an
 * in-process loop-back table instead of librccl: the device list may then name one device several times ("ranks" on one GPU)
 * and the grouped ncclAllReduce(ncclUint64, ncclMin) / ncclBroadcast call sequence of the elite exchange runs exactly as it
 * does between distinct GPUs, the data moving through plain copies when a group closes.  out[0] / out[1] = all-reduces /
 * broadcasts the loop-back table has completed in this process. */
int kao_rccl_loopback_counts(uint64_t out[2]);
/* Wall-clock breakdown of this thread's last kao_solve / kao_solve_multi, seconds from its entry:
 * out[0] session ready (instance prepared + uploaded), out[1] last improving launch finished (time-to-best),
 * out[2] results read back, out[3] returned (buffers released); out[4] = launches run; out[5] = neighbours K-search
 * delta-evaluated in those launches (kao_stats.delta_candidates, all devices), out[6] = K-bound launches, out[7] = elite
 * exchanges between GPUs (kao_solve_multi), out[8] = K-bound iterations summed over the topics, out[9] = KAO-CX calls,
 * out[10] = KAO-CX calls that improved an incumbent, out[11] = K-search iterations per restart, out[12] = generations started
 * after the first (kao_session_new_generation), out[13] = KAO-CX runs from further starting points (other restarts' best
 * snapshots; included in out[9]), out[14] = KAO-LP solves that delivered multipliers, out[15] = their interior-point iterations. */
int kao_last_solve_timing(double out[16]);
/* KAO-LP in this thread's last kao_solve: out[0] solves that delivered multipliers, out[1] their interior-point iterations, out[2] iterates
 * rounded into an assignment (kao_lp_round's primal side inside the solve), out[3] of those adopted as a topic's incumbent, out[4]
 * partitions with fractional variables summed over the rounded iterates; out[5..7] reserved (0). */
int kao_last_solve_lp(double out[8]);
/* K-search as THIS thread's last kao_solve ran it (only when that solve had kao_opts.profile = 1: every K-search / K-eval launch is
 * then bracketed by HIP events on the session's stream): out[0] = HIP-event milliseconds of all K-search launches, out[1] = of all
 * K-eval launches, out[2] = K-search launches (turns of the loop that launched none -- a huge topic while KAO-CX / KAO-LP have the GPU
 * -- are not counted), out[3] = restarts, out[4] = algorithmic bytes of those launches (neighbours x (8 RF + 10)),
 * out[5] = neighbours delta-evaluated, out[6] = dynamic LDS per K-search workgroup, out[7] = workgroups per launch.  All zero when
 * the solve was not profiled. */
int kao_last_solve_profile(double out[8]);

#ifdef __cplusplus
}
#endif
#endif /* KAO_H */
