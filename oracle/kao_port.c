/* kao_port.c -- CPU restatement (plain C) of the hot path -- TEST INFRASTRUCTURE ONLY.
 *
 * Two things live here, both used only by tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg (never by the product library):
 *
 *  1. kao_port_eval(): full evaluation of one compact candidate against the README model
 *     (objective README.md:145-146; C1 README.md:148-151; C2 153-156; C3 158-161;
 *     C4 163-166; C5 168-171; C6 173-176; C7 178-180).  Same definition as
 *     oracle/kao_oracle.py::verify(), checked against it in tests.
 *
 *  2. kao_port_search(): a scalar replay of the device's parallel-restart local search
 *     ("KAO-LS", specified in DESIGN.md section 4): per iteration the 64 wavefront lanes of the
 *     GPU (a plain loop here) score slots / scan brokers or partner slots / sample proposals,
 *     delta-evaluated against the same state, arg-min by packed key, accept rule,
 *     best-feasible snapshot.  Given the same seed it must produce bit-identical states to
 *     the HIP kernel.
 *
 * PARITY STATUS: parity unpinned beyond KAT-1 -- the reference's solver (lp_solve 5.5,
 * README.md:135-136) is absent; see oracle/kao_oracle.py header.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define NONE16 0xFFFFu
#define RFP 8 /* replica slots per partition held by the replay (the device holds 4 words per partition when RF and the
                 current RF are <= 4, else 8: ls_topic.nw) */
#define LANES 64

typedef struct {
    int32_t n_brokers, n_racks, n_partitions, rf, rf_cur;
    const uint8_t *rack_of;   /* [B] */
    const uint16_t *current;  /* [P*rf_cur] dense index or NONE16 */
    int32_t w[2][2];          /* w[cur_role][new_role] */
    int32_t rep_lo, rep_hi, lead_lo, lead_hi, rack_lo, rack_hi, prack_lo, prack_hi;
    const int32_t *broker_w;  /* [B] or NULL: extra objective weight of every replica on the broker (kao_topic.broker_w) */
    const int32_t *broker_wl; /* [B] or NULL: ... of every leader on the broker */
} port_topic;

typedef struct {
    uint64_t seed;      /* per-topic seed */
    int32_t obj_scale;  /* S: objective weights are multiplied by S inside the search cost */
    int32_t lam_min;    /* penalty per unit of violation: sawtooth lam_min..lam_max */
    int32_t lam_max;
    int32_t period_log2;/* sawtooth period of restart rho = 2^(period_log2 + (rho & 3)) iterations: the first
                           half ramps lam_min -> lam_max, the second half holds lam_max */
    int32_t team;       /* wavefronts searching the restart together (k_team, kao_opts.team): 0 / 1 = one (k_search) */
} port_params;

static inline int band(int c, int lo, int hi) {
    int a = c - hi, b = lo - c;
    return (a > 0 ? a : 0) + (b > 0 ? b : 0);
}

/* ------------------------------------------------------------------ full evaluation */
int kao_port_eval(const port_topic *t, const uint16_t *assign, int64_t *objective, int32_t viol[8]) {
    const int B = t->n_brokers, R = t->n_racks, P = t->n_partitions, RF = t->rf;
    int32_t *cr = (int32_t *)calloc((size_t)B, sizeof(int32_t));
    int32_t *cl = (int32_t *)calloc((size_t)B, sizeof(int32_t));
    int32_t *rk = (int32_t *)calloc((size_t)R, sizeof(int32_t));
    int32_t *pr = (int32_t *)calloc((size_t)R, sizeof(int32_t));
    if (!cr || !cl || !rk || !pr) { free(cr); free(cl); free(rk); free(pr); return -1; }
    int64_t obj = 0;
    memset(viol, 0, 8 * sizeof(int32_t));
    for (int p = 0; p < P; ++p) {
        memset(pr, 0, (size_t)R * sizeof(int32_t));
        for (int k = 0; k < RF; ++k) {
            unsigned b = assign[p * RF + k];
            if (b >= (unsigned)B) {
                viol[1] += 1;
                if (k == 0) viol[2] += 1;
                continue;
            }
            cr[b] += 1;
            if (k == 0) cl[b] += 1;
            rk[t->rack_of[b]] += 1;
            pr[t->rack_of[b]] += 1;
            for (int j = 0; j < k; ++j)
                if (assign[p * RF + j] == b) { viol[5] += 1; break; }
            const uint16_t *cur = t->current + (size_t)p * t->rf_cur;
            int nr = k == 0 ? 0 : 1;
            for (int j = 0; j < t->rf_cur; ++j)
                if (cur[j] == b) { obj += t->w[j == 0 ? 0 : 1][nr]; break; }
            if (t->broker_w) obj += t->broker_w[b];
            if (t->broker_wl && k == 0) obj += t->broker_wl[b];
        }
        for (int r = 0; r < R; ++r) viol[7] += band(pr[r], t->prack_lo, t->prack_hi);
    }
    for (int b = 0; b < B; ++b) {
        viol[3] += band(cr[b], t->rep_lo, t->rep_hi);
        viol[4] += band(cl[b], t->lead_lo, t->lead_hi);
    }
    for (int r = 0; r < R; ++r) viol[6] += band(rk[r], t->rack_lo, t->rack_hi);
    viol[0] = viol[1] + viol[2] + viol[3] + viol[4] + viol[5] + viol[6] + viol[7];
    *objective = obj;
    free(cr); free(cl); free(rk); free(pr);
    return 0;
}

/* ------------------------------------------------------------------ KAO-LS replay */
typedef struct {
    int P, RF, R, m, Bx; /* m = max rack size; internal index x = rack*m + j */
    int nw;              /* replica words per partition on the device: 4, or 8 when RF or the current RF exceeds 4 (enters the hole hash) */
    uint32_t magic;      /* floor(2^32/m)+1: rack(x) = mulhi(x, magic) */
    int rack_size[256];
    uint16_t *int_of;    /* [B] dense -> internal */
    uint16_t *ext_of;    /* [Bx] internal -> dense (NONE16 for padding) */
    uint16_t *cur;       /* [P*RFP] internal */
    int w[2][2];
    int rep_lo, rep_hi, lead_lo, lead_hi, rack_lo, rack_hi, prack_lo, prack_hi;
    int *bw, *bwl;       /* [Bx] broker weights per internal index (zeros when the topic has none) */
} ls_topic;

typedef struct {
    uint16_t *A;   /* [P*RFP] internal, slots >= RF are NONE16 */
    uint32_t *C;   /* [Bx] cntR | cntL << 16 */
    int K[256];    /* replicas per rack */
    int V, obj;    /* current total violation magnitude, objective */
    int best_obj;  /* best feasible objective seen, -1 if none */
    uint16_t *best;/* [P*RF] dense snapshot */
    uint64_t n_eval, n_accept;
    uint64_t n_valid; /* neighbours that were real proposals (not null: broker already in the partition, padding slot, ...) */
    const int *PA, *PL, *PG; /* search prices in key units: replica price a[b] and leader price l[b] per internal index [Bx],
                                rack price g[r] [256]; NULL = unpriced */
} ls_state;
static inline int PAx(const ls_state *s, unsigned x) { return s->PA ? s->PA[x] : 0; }
static inline int PLx(const ls_state *s, unsigned x) { return s->PL ? s->PL[x] : 0; }
static inline int PGx(const ls_state *s, int r) { return s->PG ? s->PG[r] : 0; }
/* Price of one more (p_in) / one fewer (p_out) unit on a priced row whose count is c.  The multiplier applies only where
 * the count leaves or re-enters its band [lo, hi] -- i.e. exactly where the violation changes; inside a slack band a unit
 * costs nothing (a plain linear term would push counts of rows with a positive multiplier down to the lower band end). */
static inline int p_in(int c, int lo, int hi, int price) { return (c >= hi || c < lo) ? price : 0; }
static inline int p_out(int c, int lo, int hi, int price) { return (c > hi || c <= lo) ? -price : 0; }

static inline uint32_t fmix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}
static inline uint32_t mulhi(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }
static inline uint32_t xs32(uint32_t *s) {
    uint32_t x = *s; x ^= x << 13; x ^= x >> 17; x ^= x << 5; *s = x; return x;
}
static inline int rack_of_x(const ls_topic *t, unsigned x) { return (int)mulhi(x, t->magic); }
static inline int valid_x(const ls_topic *t, unsigned x) {
    int r = rack_of_x(t, x);
    return (int)(x - (unsigned)(r * t->m)) < t->rack_size[r];
}
static inline int role_w(const ls_topic *t, int p, unsigned x, int nr) {
    const uint16_t *c = t->cur + p * RFP;
    if (c[0] == x) return t->w[0][nr];
    for (int j = 1; j < RFP; ++j) if (c[j] == x) return t->w[1][nr];
    return 0;
}
/* role weight plus the broker's own weights: what a replica of p on x in role nr adds to the objective */
static inline int slot_w(const ls_topic *t, int p, unsigned x, int nr) { return role_w(t, p, x, nr) + t->bw[x] + (nr == 0 ? t->bwl[x] : 0); }
static inline int in_part(const uint16_t *a, unsigned x) {
    for (int j = 0; j < RFP; ++j) if (a[j] == x) return 1;
    return 0;
}
static inline int rack_count(const ls_topic *t, const uint16_t *a, int r) {
    int c = 0;
    for (int s = 0; s < RFP; ++s) c += (a[s] != NONE16 && rack_of_x(t, a[s]) == r);
    return c;
}

void *kao_port_ls_create(const port_topic *pt) {
    ls_topic *t = (ls_topic *)calloc(1, sizeof(ls_topic));
    const int B = pt->n_brokers;
    t->P = pt->n_partitions; t->RF = pt->rf; t->R = pt->n_racks;
    if (t->RF > RFP || pt->rf_cur > RFP || t->R > 255) { free(t); return NULL; }
    t->nw = (t->RF > 4 || pt->rf_cur > 4) ? 8 : 4;
    for (int b = 0; b < B; ++b) t->rack_size[pt->rack_of[b]] += 1;
    for (int r = 0; r < t->R; ++r) if (t->rack_size[r] > t->m) t->m = t->rack_size[r];
    if (t->m < 2) t->m = 2; /* floor(2^32/m)+1 must fit 32 bits: single-broker racks get a stride of 2 */
    t->Bx = t->R * t->m;
    t->magic = (uint32_t)(0x100000000ull / (uint64_t)t->m) + 1u;
    t->int_of = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)B);
    t->ext_of = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)t->Bx);
    memset(t->ext_of, 0xFF, sizeof(uint16_t) * (size_t)t->Bx);
    int fill[256] = {0};
    for (int b = 0; b < B; ++b) { /* dense order inside each rack is preserved */
        int r = pt->rack_of[b];
        int x = r * t->m + fill[r]++;
        t->int_of[b] = (uint16_t)x; t->ext_of[x] = (uint16_t)b;
    }
    t->cur = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)t->P * RFP);
    memset(t->cur, 0xFF, sizeof(uint16_t) * (size_t)t->P * RFP);
    for (int p = 0; p < t->P; ++p)
        for (int k = 0; k < pt->rf_cur; ++k) {
            unsigned b = pt->current[p * pt->rf_cur + k];
            if (b < (unsigned)B) t->cur[p * RFP + k] = t->int_of[b];
        }
    t->bw = (int *)calloc((size_t)t->Bx, sizeof(int)); t->bwl = (int *)calloc((size_t)t->Bx, sizeof(int));
    for (int b = 0; b < B; ++b) {
        if (pt->broker_w) t->bw[t->int_of[b]] = pt->broker_w[b];
        if (pt->broker_wl) t->bwl[t->int_of[b]] = pt->broker_wl[b];
    }
    memcpy(t->w, pt->w, sizeof(t->w));
    t->rep_lo = pt->rep_lo; t->rep_hi = pt->rep_hi; t->lead_lo = pt->lead_lo; t->lead_hi = pt->lead_hi;
    t->rack_lo = pt->rack_lo; t->rack_hi = pt->rack_hi; t->prack_lo = pt->prack_lo; t->prack_hi = pt->prack_hi;
    return t;
}

void kao_port_ls_destroy(void *h) {
    ls_topic *t = (ls_topic *)h;
    if (!t) return;
    free(t->int_of); free(t->ext_of); free(t->cur); free(t->bw); free(t->bwl); free(t);
}

/* rebuild counters, V and obj from A (the device does this wave-parallel at every launch) */
static void ls_recount(const ls_topic *t, ls_state *s) {
    memset(s->C, 0, sizeof(uint32_t) * (size_t)t->Bx);
    memset(s->K, 0, sizeof(s->K));
    int V = 0, obj = 0;
    for (int p = 0; p < t->P; ++p) {
        const uint16_t *a = s->A + p * RFP;
        for (int k = 0; k < t->RF; ++k) {
            unsigned x = a[k];
            s->C[x] += (k == 0) ? 0x10001u : 1u;
            s->K[rack_of_x(t, x)] += 1;
            obj += slot_w(t, p, x, k == 0 ? 0 : 1);
        }
        for (int r = 0; r < t->R; ++r) V += band(rack_count(t, a, r), t->prack_lo, t->prack_hi);
    }
    for (int x = 0; x < t->Bx; ++x) {
        if (!valid_x(t, (unsigned)x)) continue;
        V += band((int)(s->C[x] & 0xFFFF), t->rep_lo, t->rep_hi);
        V += band((int)(s->C[x] >> 16), t->lead_lo, t->lead_hi);
    }
    for (int r = 0; r < t->R; ++r) V += band(s->K[r], t->rack_lo, t->rack_hi);
    s->V = V; s->obj = obj;
}

static void ls_snapshot(const ls_topic *t, ls_state *s) {
    for (int p = 0; p < t->P; ++p)
        for (int k = 0; k < t->RF; ++k) s->best[p * t->RF + k] = t->ext_of[s->A[p * RFP + k]];
}

#define KEY_NULL 0xFFFFFFFFu
#define DBIAS 32768
static inline int d_band(int c, int d, int lo, int hi) { return band(c + d, lo, hi) - band(c, lo, hi); }

/* rebuild counters from A, skipping empty (NONE16) slots; V / obj are not meaningful while
 * holes remain, so they are left to ls_recount */
static void ls_count_partial(const ls_topic *t, ls_state *s) {
    memset(s->C, 0, sizeof(uint32_t) * (size_t)t->Bx);
    memset(s->K, 0, sizeof(s->K));
    for (int p = 0; p < t->P; ++p)
        for (int k = 0; k < t->RF; ++k) {
            unsigned x = s->A[p * RFP + k];
            if (x == NONE16) continue;
            s->C[x] += (k == 0) ? 0x10001u : 1u;
            s->K[rack_of_x(t, x)] += 1;
        }
}

static inline uint32_t make_key_tie(int lam, int S, int dV, int dObj, uint32_t tie, int dP) {
    int delta = lam * dV - S * dObj + dP;
    if (delta < -DBIAS) delta = -DBIAS;
    if (delta > DBIAS - 2) delta = DBIAS - 2;
    return ((uint32_t)(delta + DBIAS) << 8) | (tie & 0xFFu);
}

/* initial state of restart `rho` (generation `gen`: kao_solve starts a new generation -- every restart re-initialised, the
 * tie-break hash salted with the generation number -- when a population has converged without a proof; 0 = the first):
 * surviving current replicas stay in their slots; every hole (replica on a removed
 * broker, or a slot added by an RF increase) is filled, in (p,k) order, by BEST INSERTION: every valid broker not in
 * the partition is scored lam_max*dV - S*dObj of the insertion (64 per round on the GPU, lane = internal index), ties
 * broken by 8 hashed bits (per restart, hole and broker), then by the lowest lane.  Two passes: leader holes of all
 * partitions first, then follower holes. */
static void ls_init(const ls_topic *t, ls_state *s, const port_params *pp, uint32_t rho, uint32_t gen) {
    const uint32_t slo = (uint32_t)pp->seed, shi = (uint32_t)(pp->seed >> 32);
    for (int p = 0; p < t->P; ++p)
        for (int k = 0; k < RFP; ++k) s->A[p * RFP + k] = (k < t->RF) ? t->cur[p * RFP + k] : NONE16;
    ls_count_partial(t, s);
    for (int pass = 0; pass < 2; ++pass) /* leader holes first (pass 0), then follower holes: leaders are the scarcer resource */
    for (int p = 0; p < t->P; ++p) {
        uint16_t *a = s->A + p * RFP;
        for (int k = (pass == 0 ? 0 : 1); k < (pass == 0 ? 1 : t->RF); ++k) {
            if (a[k] != NONE16) continue;
            const uint32_t hmix = slo ^ fmix32(shi + rho * 0x9E3779B1u + (uint32_t)(p * t->nw + k) * 0x27D4EB2Fu + 0x5BD1E995u + gen * 0x632BE5ABu);
            uint32_t lane_key[LANES]; int lane_x[LANES];
            for (uint32_t l = 0; l < LANES; ++l) lane_key[l] = KEY_NULL;
            for (int base = 0; base < t->Bx; base += LANES)
                for (uint32_t l = 0; l < LANES; ++l) {
                    const unsigned x = (unsigned)base + l;
                    if ((int)x >= t->Bx || !valid_x(t, x) || in_part(a, x)) continue;
                    const int rn = rack_of_x(t, x);
                    int dV = d_band((int)(s->C[x] & 0xFFFF), +1, t->rep_lo, t->rep_hi)
                           + d_band(s->K[rn], +1, t->rack_lo, t->rack_hi)
                           + d_band(rack_count(t, a, rn), +1, t->prack_lo, t->prack_hi);
                    if (k == 0) dV += d_band((int)(s->C[x] >> 16), +1, t->lead_lo, t->lead_hi);
                    const uint32_t tie = fmix32(hmix + x * 0x165667B1u) >> 24;
                    const uint32_t key = make_key_tie(pp->lam_max, pp->obj_scale, dV, slot_w(t, p, x, k == 0 ? 0 : 1), tie,
                                                          p_in((int)(s->C[x] & 0xFFFF), t->rep_lo, t->rep_hi, PAx(s, x)) + p_in(s->K[rn], t->rack_lo, t->rack_hi, PGx(s, rn)) +
                                                          (k == 0 ? p_in((int)(s->C[x] >> 16), t->lead_lo, t->lead_hi, PLx(s, x)) : 0));
                    if (key < lane_key[l]) { lane_key[l] = key; lane_x[l] = (int)x; }
                }
            uint32_t best_key = KEY_NULL; int found = -1;
            for (uint32_t l = 0; l < LANES; ++l)
                if (lane_key[l] < best_key) { best_key = lane_key[l]; found = lane_x[l]; }
            a[k] = (uint16_t)found; /* found >= 0 whenever B >= RF */
            s->C[found] += (k == 0) ? 0x10001u : 1u;
            s->K[rack_of_x(t, (unsigned)found)] += 1;
        }
    }
    s->best_obj = -1; s->n_eval = 0; s->n_accept = 0; s->n_valid = 0;
}

typedef struct { int type, p, k, q, j; unsigned x; int dV, dObj; } proposal;

/* move type of global iteration `it` (wave-uniform on the GPU) */
#define SCAN_TWO_SLOTS 24576 /* REPLACE scan: two tournament slots per iteration on topics of at most this many replica slots (kScanTwoSlots) */
static inline int move_type(uint32_t it) {
    static const uint8_t pat[8] = {0, 0, 1, 0, 2, 0, 1, 0}; /* 0 replace, 1 exchange, 2 leader swap */
    return pat[it & 7];
}

/* per-lane LCG modulo 2^24 (one v_mad_u32_u24 on the GPU) and range reduction by the high bits of
 * a 24x24-bit product (one v_mul_hi_u32_u24 when n < 65536; mulhi(v24 << 8, n) in general): uniform-ish on [0, n) */
static inline uint32_t lcg24(uint32_t *s) { *s = (*s & 0xFFFFFFu) * 0x6D2B79u + 0x3C6EF3u; return *s; }
static inline uint32_t rnd24(uint32_t *s, uint32_t n) {
    const uint32_t v = lcg24(s);
    return (uint32_t)(((uint64_t)((v & 0xFFFFFFu) << 8) * (uint64_t)n) >> 32); /* floor(v24 * n / 2^24), any n < 2^32 */
}
#define REPL_G 4 /* candidate brokers per lane in a REPLACE iteration: 2 of any rack, 2 of the old broker's rack */

static inline uint32_t make_key(int lam, int S, int dV, int dObj, uint32_t lane, int dP) {
    int delta = lam * dV - S * dObj + dP;
    if (delta < -DBIAS) delta = -DBIAS;
    if (delta > DBIAS - 2) delta = DBIAS - 2;
    return ((uint32_t)(delta + DBIAS) << 8) | lane;
}

/* One lane's proposals of a sampled REPLACE (type 0) or LEADER-SWAP (type 2) iteration: draws from the lane's stream, delta-evaluates every candidate
 * against the current state and returns the lane's smallest key (KEY_NULL if none is valid), with the
 * matching proposal in *o.  Ties inside a lane go to the earlier candidate. */
static uint32_t ls_lane(const ls_topic *t, const ls_state *s, int type, uint32_t *rng, int lam, int S, uint32_t lane,
                        proposal *o, uint64_t *n_eval) {
    const int P = t->P, RF = t->RF;
    uint32_t best = KEY_NULL;
    const int p = (int)rnd24(rng, (uint32_t)P);
    const uint16_t *a = s->A + p * RFP;
    if (type == 0) { /* REPLACE (p,k) <- x_g, g = 0..3 */
        const int k = (int)rnd24(rng, (uint32_t)RF);
        const unsigned old = a[k];
        const int nr = k == 0 ? 0 : 1;
        const int ro = rack_of_x(t, old);
        const int g_old = slot_w(t, p, old, nr);
        const uint32_t co = s->C[old];
        int dV_old = d_band((int)(co & 0xFFFF), -1, t->rep_lo, t->rep_hi);
        if (k == 0) dV_old += d_band((int)(co >> 16), -1, t->lead_lo, t->lead_hi);
        const int dV_rack_old = d_band(s->K[ro], -1, t->rack_lo, t->rack_hi) + d_band(rack_count(t, a, ro), -1, t->prack_lo, t->prack_hi);
        int dP_old = p_out((int)(co & 0xFFFF), t->rep_lo, t->rep_hi, PAx(s, old));
        if (k == 0) dP_old += p_out((int)(co >> 16), t->lead_lo, t->lead_hi, PLx(s, old));
        const int dP_rack_old = p_out(s->K[ro], t->rack_lo, t->rack_hi, PGx(s, ro));
        for (int g = 0; g < REPL_G; ++g) {
            int r, j;
            if (g < 2) { r = (int)rnd24(rng, (uint32_t)t->R); j = (int)rnd24(rng, (uint32_t)t->m); }
            else { r = ro; j = (int)rnd24(rng, (uint32_t)t->rack_size[ro]); }
            const unsigned x = (unsigned)(r * t->m + j);
            *n_eval += 1;
            if (j >= t->rack_size[r] || in_part(a, x)) continue;
            ((ls_state *)s)->n_valid += 1;
            const uint32_t cn = s->C[x];
            int dV = dV_old + d_band((int)(cn & 0xFFFF), +1, t->rep_lo, t->rep_hi);
            if (k == 0) dV += d_band((int)(cn >> 16), +1, t->lead_lo, t->lead_hi);
            if (r != ro)
                dV += dV_rack_old + d_band(s->K[r], +1, t->rack_lo, t->rack_hi) + d_band(rack_count(t, a, r), +1, t->prack_lo, t->prack_hi);
            const int dObj = slot_w(t, p, x, nr) - g_old;
            int dP = dP_old + p_in((int)(cn & 0xFFFF), t->rep_lo, t->rep_hi, PAx(s, x));
            if (k == 0) dP += p_in((int)(cn >> 16), t->lead_lo, t->lead_hi, PLx(s, x));
            if (r != ro) dP += dP_rack_old + p_in(s->K[r], t->rack_lo, t->rack_hi, PGx(s, r));
            const uint32_t key = make_key(lam, S, dV, dObj, lane, dP);
            if (key < best) { best = key; o->type = 0; o->p = p; o->k = k; o->x = x; o->dV = dV; o->dObj = dObj; }
        }
        return best;
    }
    /* LEADER SWAP inside p: slot 0 <-> slot k, every k = 1..RF-1 is a candidate */
    const unsigned u = a[0];
    for (int k = 1; k < RF; ++k) {
        const unsigned v = a[k];
        *n_eval += 1;
        ((ls_state *)s)->n_valid += 1;
        const int dObj = role_w(t, p, v, 0) + role_w(t, p, u, 1) - role_w(t, p, u, 0) - role_w(t, p, v, 1) + t->bwl[v] - t->bwl[u];
        const int dV = d_band((int)(s->C[u] >> 16), -1, t->lead_lo, t->lead_hi) + d_band((int)(s->C[v] >> 16), +1, t->lead_lo, t->lead_hi);
        const uint32_t key = make_key(lam, S, dV, dObj, lane, p_out((int)(s->C[u] >> 16), t->lead_lo, t->lead_hi, PLx(s, u)) +
                                                                 p_in((int)(s->C[v] >> 16), t->lead_lo, t->lead_hi, PLx(s, v)));
        if (key < best) { best = key; o->type = 2; o->p = p; o->k = k; o->dV = dV; o->dObj = dObj; }
    }
    return best;
}

static void ls_apply(const ls_topic *t, ls_state *s, const proposal *o) {
    uint16_t *a = s->A + o->p * RFP;
    if (o->type == 0) {
        unsigned old = a[o->k];
        uint32_t d = (o->k == 0) ? 0x10001u : 1u;
        s->C[old] -= d; s->C[o->x] += d;
        s->K[rack_of_x(t, old)] -= 1; s->K[rack_of_x(t, o->x)] += 1;
        a[o->k] = (uint16_t)o->x;
    } else if (o->type == 1) {
        uint16_t *b = s->A + o->q * RFP;
        unsigned u = a[o->k], v = b[o->j];
        if ((o->k == 0) != (o->j == 0)) {
            unsigned lose = (o->k == 0) ? u : v, gain = (o->k == 0) ? v : u;
            s->C[lose] -= 0x10000u; s->C[gain] += 0x10000u;
        }
        a[o->k] = (uint16_t)v; b[o->j] = (uint16_t)u;
    } else {
        unsigned u = a[0], v = a[o->k];
        s->C[u] -= 0x10000u; s->C[v] += 0x10000u;
        a[0] = (uint16_t)v; a[o->k] = (uint16_t)u;
    }
    s->V += o->dV; s->obj += o->dObj;
}

/* ---------------------------------------------------------------------------------------------
 * One launch of KAO-LS.  Move type of iteration `it`: pattern R R X R L R X R.
 *   REPLACE, blocks of 8 iterations alternate between two styles ((it >> 3) & 1):
 *     scan   : a tournament over T random slots (lowest "removal score") picks the slots of its TWO best lanes (round 4; one
 *              slot until then: the per-iteration overhead -- loop head, penalty, accept, bookkeeping -- is paid once for
 *              twice the neighbours); for each of them, first the winner's, every target broker is delta-evaluated, 64
 *              per round (lane = internal index); the better of the two moves is the proposal (ties: the first slot's);
 *     sample : every lane proposes its own random slot and 4 candidate brokers (ls_lane).
 *   EXCHANGE : tournament slot (p,k), then every partner slot (q,j) is scanned (lane = partition q); topics with
 *              more than 512 partitions scan a random window of 512 (8 rounds).
 *   The tournament scores clamp(P*RF/256, 1, 16) slots per lane on the first T lanes -- a random partition and the ones that
 *   follow it, a random slot of each; the score is the cost of
 *   taking the replica out of its slot under the current penalty (for an EXCHANGE only the partition's rack spread counts).
 *   LEADER-SWAP : every lane a random partition, all RF-1 swaps (ls_lane).
 * --------------------------------------------------------------------------------------------- */

/* The proposal of ONE wavefront for iteration `it` (move type `type`, penalty `lam`) against the state *s, which it does not
 * change: draws from the wavefront's 64 lane streams `rng`, returns the wavefront's smallest key (KEY_NULL: no valid candidate)
 * and the matching move in *out. */
static uint32_t ls_propose(const ls_topic *t, ls_state *s, int type, uint32_t it, uint32_t *rng, int lam, int S, proposal *out) {
    const int P = t->P, RF = t->RF;
    int T = (P * RF) / 4; if (T < 4) T = 4; if (T > LANES) T = LANES; /* tournament size */
    int GA = (P * RF) / 256; if (GA < 1) GA = 1; if (GA > 16) GA = 16; /* slots scored per lane in the tournament */
    const int XW = 8; /* an EXCHANGE scans at most XW rounds of 64 partitions (a random window when P is larger) */
    {
        uint32_t best_key = KEY_NULL;
        proposal bp; memset(&bp, 0, sizeof bp);
        if (type == 2 || (type == 0 && ((it >> 3) & 1))) { /* per-lane proposals: LEADER SWAP, sampled REPLACE */
            for (uint32_t l = 0; l < LANES; ++l) {
                proposal o; memset(&o, 0, sizeof o);
                const uint32_t key = ls_lane(t, s, type, &rng[l], lam, S, l, &o, &s->n_eval);
                if (key < best_key) { best_key = key; bp = o; }
            }
        } else {
            /* ---- phase A: tournament over T random slots, lowest removal score wins ---- */
            uint32_t keyA = KEY_NULL; int p = 0, k = 0;
            uint32_t lkey[LANES]; int lp[LANES], lk[LANES];   /* every lane's best slot (the second scan slot comes from the second-best LANE) */
            for (uint32_t l = 0; l < LANES; ++l) { lkey[l] = KEY_NULL; lp[l] = 0; lk[l] = 0; }
            for (uint32_t l = 0; l < LANES; ++l)
              for (int ga = 0, p0 = 0; ga < GA; ++ga) {
                /* the lane draws ONE partition; its further slots come from the partitions that follow it (cyclically): on topics
                 * that live in HBM the 16 slots of a lane then share two cache lines instead of touching 16 */
                if (ga == 0) p0 = (int)rnd24(&rng[l], (uint32_t)P);
                const int pl = p0 + ga < P ? p0 + ga : p0 + ga - P;
                const int kl = (int)rnd24(&rng[l], (uint32_t)RF);
                if ((int)l >= T) continue;
                const uint16_t *al = s->A + pl * RFP;
                const unsigned old = al[kl];
                const int ro = rack_of_x(t, old);
                const uint32_t co = s->C[old];
                int dvo = d_band((int)(co & 0xFFFF), -1, t->rep_lo, t->rep_hi);
                if (kl == 0) dvo += d_band((int)(co >> 16), -1, t->lead_lo, t->lead_hi);
                const int dvr = d_band(s->K[ro], -1, t->rack_lo, t->rack_hi) + d_band(rack_count(t, al, ro), -1, t->prack_lo, t->prack_hi);
                /* removal score of the slot.  REPLACE: the replica leaves its broker and (at best) its rack.  EXCHANGE:
                 * broker and rack totals do not change, only the partition's own rack spread (C7) can improve. */
                const int dv7 = d_band(rack_count(t, al, ro), -1, t->prack_lo, t->prack_hi);
                /* an exchange can also change who leads: a leader slot may shed a leader, a follower slot may gain one */
                const int dvl = d_band((int)(co >> 16), kl == 0 ? -1 : +1, t->lead_lo, t->lead_hi);
                const int sc = (type == 0) ? dvo + (dvr < 0 ? dvr : 0) : (dv7 < 0 ? dv7 : 0) + (dvl < 0 ? dvl : 0);
                const uint32_t key = make_key(lam, S, sc, -(type == 0 ? slot_w(t, pl, old, kl == 0 ? 0 : 1) : role_w(t, pl, old, kl == 0 ? 0 : 1)), l, type == 0 ? p_out((int)(co & 0xFFFF), t->rep_lo, t->rep_hi, PAx(s, old)) + (kl == 0 ? p_out((int)(co >> 16), t->lead_lo, t->lead_hi, PLx(s, old)) : 0) : 0);
                if (key < keyA) { keyA = key; p = pl; k = kl; }
                if (ga == 0 || key < lkey[l]) { lkey[l] = key; lp[l] = pl; lk[l] = kl; }
              }
            /* REPLACE scan: the slot of the best lane, then the slot of the best OTHER lane (none when no other lane takes part) */
            int n_slots = 1, sp[2], sk[2];
            sp[0] = p; sk[0] = k;
            if (type == 0 && P * RF <= SCAN_TWO_SLOTS) {
                uint32_t k2 = KEY_NULL; const uint32_t wA = keyA & 63u;
                for (uint32_t l = 0; l < LANES; ++l)
                    if (l != wA && lkey[l] < k2) { k2 = lkey[l]; sp[1] = lp[l]; sk[1] = lk[l]; }
                if (k2 != KEY_NULL) n_slots = 2;
            }
            for (int si = 0; si < n_slots; ++si) {
            p = sp[si]; k = sk[si];
            const uint16_t *a = s->A + p * RFP;
            const unsigned old = a[k];
            const int nr = k == 0 ? 0 : 1;
            if (type == 0) { /* ---- phase B: scan every target broker for slot (p,k) ---- */
                const int ro = rack_of_x(t, old);
                const int g_old = slot_w(t, p, old, nr);
                const uint32_t co = s->C[old];
                int dV_old = d_band((int)(co & 0xFFFF), -1, t->rep_lo, t->rep_hi);
                if (k == 0) dV_old += d_band((int)(co >> 16), -1, t->lead_lo, t->lead_hi);
                const int dV_rack_old = d_band(s->K[ro], -1, t->rack_lo, t->rack_hi) + d_band(rack_count(t, a, ro), -1, t->prack_lo, t->prack_hi);
                int RT[256], RTP[256];
                int dP_old = p_out((int)(co & 0xFFFF), t->rep_lo, t->rep_hi, PAx(s, old));
                if (k == 0) dP_old += p_out((int)(co >> 16), t->lead_lo, t->lead_hi, PLx(s, old));
                for (int r = 0; r < t->R; ++r) {
                    RT[r] = (r == ro) ? 0 : dV_rack_old + d_band(s->K[r], +1, t->rack_lo, t->rack_hi) + d_band(rack_count(t, a, r), +1, t->prack_lo, t->prack_hi);
                    RTP[r] = (r == ro) ? 0 : p_out(s->K[ro], t->rack_lo, t->rack_hi, PGx(s, ro)) + p_in(s->K[r], t->rack_lo, t->rack_hi, PGx(s, r));
                }
                uint32_t lane_key[LANES]; unsigned lane_x[LANES]; int lane_dV[LANES], lane_dO[LANES];
                for (uint32_t l = 0; l < LANES; ++l) lane_key[l] = KEY_NULL;
                for (int base = 0; base < t->Bx; base += LANES)
                    for (uint32_t l = 0; l < LANES; ++l) {
                        const uint32_t tie = lcg24(&rng[l]) >> 8;
                        const unsigned x = (unsigned)base + l;
                        if ((int)x >= t->Bx || !valid_x(t, x)) continue;
                        s->n_eval += 1;
                        if (in_part(a, x)) continue;
                        s->n_valid += 1;
                        const uint32_t cn = s->C[x];
                        int dV = dV_old + d_band((int)(cn & 0xFFFF), +1, t->rep_lo, t->rep_hi) + RT[rack_of_x(t, x)];
                        if (k == 0) dV += d_band((int)(cn >> 16), +1, t->lead_lo, t->lead_hi);
                        const int dObj = slot_w(t, p, x, nr) - g_old;
                        int dP = dP_old + p_in((int)(cn & 0xFFFF), t->rep_lo, t->rep_hi, PAx(s, x)) + RTP[rack_of_x(t, x)];
                        if (k == 0) dP += p_in((int)(cn >> 16), t->lead_lo, t->lead_hi, PLx(s, x));
                        const uint32_t key = make_key_tie(lam, S, dV, dObj, tie, dP);
                        if (key < lane_key[l]) { lane_key[l] = key; lane_x[l] = x; lane_dV[l] = dV; lane_dO[l] = dObj; }
                    }
                for (uint32_t l = 0; l < LANES; ++l)
                    if (lane_key[l] < best_key) { best_key = lane_key[l]; bp.type = 0; bp.p = p; bp.k = k; bp.x = lane_x[l]; bp.dV = lane_dV[l]; bp.dObj = lane_dO[l]; }
            } else { /* ---- phase B: scan every partner slot (q,j) for an exchange with (p,k) ---- */
                const unsigned u = old;
                const int ru = rack_of_x(t, u);
                const int gu_p = role_w(t, p, u, nr);
                uint32_t lane_key[LANES]; int lane_q[LANES], lane_j[LANES], lane_dV[LANES], lane_dO[LANES];
                for (uint32_t l = 0; l < LANES; ++l) lane_key[l] = KEY_NULL;
                int nrounds = (P + LANES - 1) / LANES, q0 = 0;
                const int windowed = nrounds > XW;
                for (uint32_t l = 0; l < LANES; ++l) { /* every lane draws; lane 0's value places the window */
                    const int d = (int)rnd24(&rng[l], (uint32_t)P);
                    if (l == 0 && windowed) q0 = d;
                }
                if (windowed) nrounds = XW;
                for (int rd = 0; rd < nrounds; ++rd)
                    for (uint32_t l = 0; l < LANES; ++l) {
                        const uint32_t tie0 = lcg24(&rng[l]) >> 8;
                        int q = q0 + rd * LANES + (int)l;
                        if (q >= P) { if (!windowed) continue; q -= P; }
                        const uint16_t *b = s->A + q * RFP;
                        for (int j = 0; j < RF; ++j) {
                            s->n_eval += 1;
                            const unsigned v = b[j];
                            if (q == p || u == v || in_part(a, v) || in_part(b, u)) continue;
                            s->n_valid += 1;
                            const int nrq = j == 0 ? 0 : 1;
                            int dObj = role_w(t, p, v, nr) + role_w(t, q, u, nrq) - gu_p - role_w(t, q, v, nrq);
                            int dV = 0, dP = 0;
                            if ((k == 0) != (j == 0)) {
                                const unsigned lose = (k == 0) ? u : v, gain = (k == 0) ? v : u;
                                dObj += t->bwl[gain] - t->bwl[lose];   /* the leader (and its broker weight) moves */
                                dV += d_band((int)(s->C[lose] >> 16), -1, t->lead_lo, t->lead_hi) + d_band((int)(s->C[gain] >> 16), +1, t->lead_lo, t->lead_hi);
                                dP = p_out((int)(s->C[lose] >> 16), t->lead_lo, t->lead_hi, PLx(s, lose)) + p_in((int)(s->C[gain] >> 16), t->lead_lo, t->lead_hi, PLx(s, gain));
                            }
                            const int rv = rack_of_x(t, v);
                            if (ru != rv)
                                dV += d_band(rack_count(t, a, ru), -1, t->prack_lo, t->prack_hi) + d_band(rack_count(t, a, rv), +1, t->prack_lo, t->prack_hi)
                                    + d_band(rack_count(t, b, rv), -1, t->prack_lo, t->prack_hi) + d_band(rack_count(t, b, ru), +1, t->prack_lo, t->prack_hi);
                            const uint32_t key = make_key_tie(lam, S, dV, dObj, tie0 + (uint32_t)j * 0x55u, dP);
                            if (key < lane_key[l]) { lane_key[l] = key; lane_q[l] = q; lane_j[l] = j; lane_dV[l] = dV; lane_dO[l] = dObj; }
                        }
                    }
                for (uint32_t l = 0; l < LANES; ++l)
                    if (lane_key[l] < best_key) { best_key = lane_key[l]; bp.type = 1; bp.p = p; bp.k = k; bp.q = lane_q[l]; bp.j = lane_j[l]; bp.dV = lane_dV[l]; bp.dObj = lane_dO[l]; }
            }
            }
        }
        *out = bp;
        return best_key;
    }
}

/* One launch.  team = W wavefronts search the restart together (k_team; W = 1: k_search, one wavefront): lane l of wavefront
 * w draws from stream 64 w + l; in every iteration each wavefront makes its own proposal (ls_propose) against the SAME state;
 * a proposal is acceptable when its key is not null and its cost is <= 0; it is APPLIED iff it is acceptable and shares no
 * resource with an acceptable proposal of a lower-numbered wavefront.  Resources of a move: the partitions it rewrites (p, and q
 * of an EXCHANGE), the two brokers (leaving and entering the slot; an EXCHANGE: the two brokers that trade places), and -- only
 * for a REPLACE whose brokers sit in different racks -- the two racks.  Moves with disjoint resources commute and their
 * violation / objective deltas add up; the best-snapshot rule is evaluated once per iteration, after the applied moves. */
static void ls_run(const ls_topic *t, ls_state *s, const port_params *pp, uint32_t rho, uint32_t launch, uint32_t iters) {
    const uint32_t slo = (uint32_t)pp->seed, shi = (uint32_t)(pp->seed >> 32);
    const int S = pp->obj_scale;
    enum { TEAM_MAX = 16 };
    const int W = pp->team < 1 ? 1 : (pp->team > TEAM_MAX ? TEAM_MAX : pp->team);
    static __thread uint32_t rng[TEAM_MAX][LANES];
    for (int w = 0; w < W; ++w)
        for (uint32_t l = 0; l < LANES; ++l)
            rng[w][l] = fmix32(slo ^ fmix32(shi + rho * 0x9E3779B1u + launch * 0x85EBCA77u + ((uint32_t)w * LANES + l) * 0xC2B2AE3Du));
    ls_recount(t, s);
    if (s->V == 0 && s->obj > s->best_obj) { s->best_obj = s->obj; ls_snapshot(t, s); }
    const uint32_t plog = (uint32_t)pp->period_log2 + (rho & 3u);
    const uint32_t pmask = (1u << plog) - 1u;
    for (uint32_t i = 0; i < iters; ++i) {
        const uint32_t it = launch * iters + i;
        const int type = move_type(it);
        const uint32_t ph = it & pmask;
        int lam = pp->lam_min + (int)((2u * ph * (uint32_t)(pp->lam_max - pp->lam_min + 1)) >> plog);
        if (lam > pp->lam_max) lam = pp->lam_max;
        if (s->best_obj < 0) lam = pp->lam_max; /* no oscillation before the restart has been feasible once */
        proposal bp[TEAM_MAX];
        int ok[TEAM_MAX], rp[TEAM_MAX], rq[TEAM_MAX], rb0[TEAM_MAX], rb1[TEAM_MAX], rr0[TEAM_MAX], rr1[TEAM_MAX], applied[TEAM_MAX];
        for (int w = 0; w < W; ++w) {
            const uint32_t key = ls_propose(t, s, type, it, rng[w], lam, S, &bp[w]);
            ok[w] = key != KEY_NULL && (int)(key >> 8) - DBIAS <= 0;
            if (!ok[w]) continue;
            const uint16_t *a = s->A + bp[w].p * RFP;
            rp[w] = bp[w].p; rq[w] = bp[w].type == 1 ? bp[w].q : bp[w].p;
            rr0[w] = rr1[w] = -1;
            if (bp[w].type == 0) {
                rb0[w] = a[bp[w].k]; rb1[w] = (int)bp[w].x;
                const int ro = rack_of_x(t, (unsigned)rb0[w]), rn = rack_of_x(t, (unsigned)rb1[w]);
                if (ro != rn) { rr0[w] = ro; rr1[w] = rn; }
            } else if (bp[w].type == 1) { rb0[w] = a[bp[w].k]; rb1[w] = s->A[bp[w].q * RFP + bp[w].j]; }
            else { rb0[w] = a[0]; rb1[w] = a[bp[w].k]; }
        }
        for (int w = 0; w < W; ++w) {
            applied[w] = ok[w];
            for (int v = 0; v < w && applied[w]; ++v) {
                if (!ok[v]) continue;
                int clash = rp[v] == rp[w] || rp[v] == rq[w] || rq[v] == rp[w] || rq[v] == rq[w] ||
                            rb0[v] == rb0[w] || rb0[v] == rb1[w] || rb1[v] == rb0[w] || rb1[v] == rb1[w];
                if (rr0[v] >= 0 && rr0[w] >= 0) clash = clash || rr0[v] == rr0[w] || rr0[v] == rr1[w] || rr1[v] == rr0[w] || rr1[v] == rr1[w];
                if (clash) applied[w] = 0;
            }
        }
        for (int w = 0; w < W; ++w)
            if (applied[w]) { ls_apply(t, s, &bp[w]); s->n_accept++; }
        if (s->V == 0 && s->obj > s->best_obj) { s->best_obj = s->obj; ls_snapshot(t, s); }
    }
}

/* ---- launch-by-launch replay of one restart (sessions with search prices and elite launches) ----
 * port_extra carries what the device reads at the start of a launch besides the restart's own state:
 *   search prices (K-bound's multipliers on the quarter grid, or host-set): dense index, fixed point 4096; key units are
 *   (obj_scale * v + 2048) >> 12 clamped to 16 bits -- replica price a[b], leader price l[b], rack price g[r]; a price
 *   enters the cost of a move only where the row's count leaves or re-enters its band (p_in / p_out above);
 *   the elite: the topic's best feasible assignment as of the previous step, its objective and its restart.  Elite rule:
 *   a restart other than the elite's whose best feasible objective is below the elite's re-seeds its state from it when
 *   bit 0 of fmix32(seed_lo ^ rho * 0x9E3779B1 ^ launch * 0x85EBCA77 ^ 0xE117E) is set. */
typedef struct {
    const int32_t *pa, *pl, *pg;
    const uint16_t *elite;
    int32_t elite_obj, elite_rho;
    int32_t gen;   /* > 0: this launch starts generation `gen`: the restart is re-initialised (ls_init with the generation salt) */
} port_extra;
typedef struct { const ls_topic *t; ls_state s; port_params pp; uint32_t rho; int *PA, *PL; int PG[256]; } ls_runner;

static inline int price_units(int v, int S) {
    int u = (S * v + (1 << 15)) >> 16; /* fixed point 65536 = 1 (DB_SCALE); arithmetic shift: floor */
    if (u < -32767) u = -32767;
    if (u > 32767) u = 32767;
    return u;
}

void *kao_port_run_create(void *h, const port_params *pp, uint32_t rho) {
    const ls_topic *t = (const ls_topic *)h;
    ls_runner *r = (ls_runner *)calloc(1, sizeof(ls_runner));
    r->t = t; r->pp = *pp; r->rho = rho;
    r->s.A = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)t->P * RFP);
    r->s.C = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)t->Bx);
    r->s.best = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)t->P * t->RF);
    memset(r->s.best, 0xFF, sizeof(uint16_t) * (size_t)t->P * t->RF);
    r->PA = (int *)calloc((size_t)t->Bx, sizeof(int));
    r->PL = (int *)calloc((size_t)t->Bx, sizeof(int));
    return r;
}

void kao_port_run_destroy(void *run) {
    ls_runner *r = (ls_runner *)run;
    if (!r) return;
    free(r->s.A); free(r->s.C); free(r->s.best); free(r->PA); free(r->PL); free(r);
}

/* One launch (launch 0 builds the initial state first).  n_brokers = dense broker count (length of pa / pl). */
int kao_port_run_launch(void *run, uint32_t launch, uint32_t iters, const port_extra *ex, int32_t n_brokers) {
    ls_runner *r = (ls_runner *)run;
    const ls_topic *t = r->t;
    ls_state *s = &r->s;
    s->PA = NULL; s->PL = NULL; s->PG = NULL;
    if (ex && ex->pa && ex->pl && ex->pg) {
        for (int b = 0; b < n_brokers; ++b) {
            const unsigned x = t->int_of[b];
            r->PA[x] = price_units(ex->pa[b], r->pp.obj_scale);
            r->PL[x] = price_units(ex->pl[b], r->pp.obj_scale);
        }
        for (int q = 0; q < 256; ++q) r->PG[q] = q < t->R ? price_units(ex->pg[q], r->pp.obj_scale) : 0;
        s->PA = r->PA; s->PL = r->PL; s->PG = r->PG;
    }
    if (launch == 0) ls_init(t, s, &r->pp, r->rho, 0);
    else if (ex && ex->gen > 0) {
        ls_init(t, s, &r->pp, r->rho, (uint32_t)ex->gen);
        memset(s->best, 0xFF, sizeof(uint16_t) * (size_t)t->P * t->RF);   /* the snapshots of the old generation are dropped */
    }
    else if (ex && ex->elite) {
        const uint32_t slo = (uint32_t)r->pp.seed;
        const int go = (uint32_t)ex->elite_rho != r->rho && s->best_obj < ex->elite_obj &&
                       (fmix32(slo ^ (r->rho * 0x9E3779B1u) ^ (launch * 0x85EBCA77u) ^ 0xE117Eu) & 1u);
        if (go)
            for (int p = 0; p < t->P; ++p)
                for (int k = 0; k < RFP; ++k) s->A[p * RFP + k] = k < t->RF ? t->int_of[ex->elite[p * t->RF + k]] : NONE16;
    }
    ls_run(t, s, &r->pp, r->rho, launch, iters);
    return 0;
}

/* final state (dense, [P*RF]), best snapshot (dense), stats[6] = {best_obj, V, obj, n_eval lo, n_eval hi, n_accept} */
int kao_port_run_read(void *run, uint16_t *final_dense, uint16_t *best_dense, int64_t stats[6]) {
    ls_runner *r = (ls_runner *)run;
    const ls_topic *t = r->t;
    ls_state *s = &r->s;
    ls_recount(t, s);
    for (int p = 0; p < t->P; ++p)
        for (int k = 0; k < t->RF; ++k) final_dense[p * t->RF + k] = t->ext_of[s->A[p * RFP + k]];
    memcpy(best_dense, s->best, sizeof(uint16_t) * (size_t)t->P * t->RF);
    stats[0] = s->best_obj; stats[1] = s->V; stats[2] = s->obj;
    stats[3] = (int64_t)(s->n_eval & 0xFFFFFFFFu); stats[4] = (int64_t)(s->n_eval >> 32); stats[5] = (int64_t)s->n_accept;
    return 0;
}

/* Search one restart from scratch: `launches` launches of `iters` iterations each.
 * Outputs: final state (dense, [P*RF]), best snapshot (dense, [P*RF]), stats[6] =
 * {best_obj, V, obj, n_eval lo, n_eval hi, n_accept}. */
int kao_port_search(void *h, const port_params *pp, uint32_t rho, uint32_t launches, uint32_t iters,
                    uint16_t *final_dense, uint16_t *best_dense, int64_t stats[6]) {
    const ls_topic *t = (const ls_topic *)h;
    ls_state s; memset(&s, 0, sizeof s);
    s.A = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)t->P * RFP);
    s.C = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)t->Bx);
    s.best = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)t->P * t->RF);
    memset(s.best, 0xFF, sizeof(uint16_t) * (size_t)t->P * t->RF);
    ls_init(t, &s, pp, rho, 0);
    for (uint32_t L = 0; L < launches; ++L) ls_run(t, &s, pp, rho, L, iters);
    ls_recount(t, &s);
    for (int p = 0; p < t->P; ++p)
        for (int k = 0; k < t->RF; ++k) final_dense[p * t->RF + k] = t->ext_of[s.A[p * RFP + k]];
    memcpy(best_dense, s.best, sizeof(uint16_t) * (size_t)t->P * t->RF);
    stats[0] = s.best_obj; stats[1] = s.V; stats[2] = s.obj;
    stats[3] = (int64_t)(s.n_eval & 0xFFFFFFFFu); stats[4] = (int64_t)(s.n_eval >> 32); stats[5] = (int64_t)s.n_accept;
    free(s.A); free(s.C); free(s.best);
    return 0;
}

/* Fraction of the delta-evaluated neighbours of one restart that were real (non-null) proposals: bench.py multiplies the
 * device's neighbour count by it (the device evaluates the same proposals, null ones included, and does not count them apart). */
double kao_port_valid_fraction(void *h, const port_params *pp, uint32_t rho, uint32_t launches, uint32_t iters) {
    const ls_topic *t = (const ls_topic *)h;
    ls_state s; memset(&s, 0, sizeof s);
    s.A = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)t->P * RFP);
    s.C = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)t->Bx);
    s.best = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)t->P * t->RF);
    ls_init(t, &s, pp, rho, 0);
    for (uint32_t L = 0; L < launches; ++L) ls_run(t, &s, pp, rho, L, iters);
    const double f = s.n_eval ? (double)s.n_valid / (double)s.n_eval : 0.0;
    free(s.A); free(s.C); free(s.best);
    return f;
}

/* Throughput driver for bench.py's cpu_baseline leg: restarts rho0 .. rho0+n-1 of one topic replayed on `threads`
 * native threads (restarts are independent; no Python in the loop).  Returns the number of neighbours evaluated. */
#include <pthread.h>
typedef struct { void *h; const port_params *pp; uint32_t rho0, n, stride, first, launches, iters; uint64_t n_eval; } many_job;
static void *many_worker(void *arg) {
    many_job *j = (many_job *)arg;
    const ls_topic *t = (const ls_topic *)j->h;
    uint16_t *fin = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)t->P * t->RF), *best = (uint16_t *)malloc(sizeof(uint16_t) * (size_t)t->P * t->RF);
    for (uint32_t r = j->first; r < j->n; r += j->stride) {
        int64_t st[6];
        kao_port_search(j->h, j->pp, j->rho0 + r, j->launches, j->iters, fin, best, st);
        j->n_eval += (uint64_t)st[3] | ((uint64_t)st[4] << 32);
    }
    free(fin); free(best);
    return NULL;
}
uint64_t kao_port_search_many(void *h, const port_params *pp, uint32_t rho0, uint32_t n, uint32_t launches, uint32_t iters, uint32_t threads) {
    if (threads < 1) threads = 1;
    if (threads > n) threads = n;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * threads);
    many_job *jobs = (many_job *)calloc(threads, sizeof(many_job));
    for (uint32_t i = 0; i < threads; ++i) {
        jobs[i].h = h; jobs[i].pp = pp; jobs[i].rho0 = rho0; jobs[i].n = n; jobs[i].stride = threads; jobs[i].first = i;
        jobs[i].launches = launches; jobs[i].iters = iters;
        pthread_create(&th[i], NULL, many_worker, &jobs[i]);
    }
    uint64_t total = 0;
    for (uint32_t i = 0; i < threads; ++i) { pthread_join(th[i], NULL); total += jobs[i].n_eval; }
    free(th); free(jobs);
    return total;
}

/* ------------------------------------------------------------------ KAO-DB: Lagrangian dual bound
 *
 * Scalar replay of the device's dual-bound kernel (k_bound).  lp_solve proves optimality by
 * branch-and-bound on the LP relaxation (README.md:135-136); here the certificate is a Lagrangian dual of the same
 * 0-1 model (README.md:144-185): the coupling rows -- replicas per broker (C3, README.md:158-161), leaders per
 * broker (C4, README.md:163-166), replicas per rack (C6, README.md:173-176) -- are priced with integer multipliers
 * a[b], l[b], g[r] (fixed point, DB_SCALE = 1), the rows local to a partition (C1, C2, C5, C7) stay in a
 * per-partition subproblem that is solved exactly, and
 *     L(a,l,g) = sum_p max{priced value of one partition's leader + followers} + sum (multiplier x band end)
 * is an upper bound on the optimum for ANY multipliers; floor(min L / DB_SCALE) is the certificate.  The multipliers
 * move along a deflected subgradient d = 16*s + 3/4 d_prev with the Polyak step (L - level) / |d|^2, where the level is
 * the incumbent `target` (a known feasible objective) while that keeps working and moves up towards the record dual
 * value when it does not (level control below), all in integers so that the device and this replay agree bit for
 * bit. */
#define DB_LOG2 16
#define DB_SCALE (1 << DB_LOG2)
#define DB_CLAMP (1 << 26)
#define DB_STAGE 100
/* Deflection d = 2^(6-k) s + floor((2^k - 1) d_prev / 2^k), a memory of about 2^k iterations: k = 2 (16 s + 3/4 d_prev) up to
 * DB_DEFL_P partitions, k = 4 (4 s + 15/16 d_prev) beyond.  With thousands of partitions a subproblem solution is bang-bang
 * (|s|^2 ~ 2e5 on the drifted 400 x 3000 topic: a price change of a thousandth flips hundreds of partitions), one subgradient
 * says little, and the average over 16 of them -- the residual of an averaged, nearly LP-feasible assignment -- is a far
 * better direction: that topic (LP optimum 22586) reaches 22588.8 after 4,200 iterations instead of stalling at 22601.4.
 * Small topics close faster with the short memory (wide family: 174 of 175 within 1,500 iterations, 166 with k = 4). */
#define DB_DEFL_P 2048
static inline int db_defl(int n_partitions) { return n_partitions > DB_DEFL_P ? 4 : 2; }
#define DB_QUARTER_LOG2 (DB_LOG2 - 2) /* DB_SCALE / 4: the quarter grid of the rounding probes and of the search prices */
static inline int32_t db_round(int32_t v, int sh) { return (int32_t)(((v + (1 << (sh - 1))) >> sh) << sh); } /* nearest multiple, half up */

typedef struct { int b[RFP]; int f[RFP]; int r[RFP]; int n; } db_set;

static inline int db_wcur(const port_topic *t, const uint16_t *cur, unsigned b, int new_role) {
    for (int j = 0; j < t->rf_cur; ++j)
        if (cur[j] == b) return t->w[j == 0 ? 0 : 1][new_role];
    return 0;
}

/* One partition's priced subproblem.  Returns 0 and fills S (the chosen brokers, leader first) and *val, or -1 if
 * no set of RF brokers satisfies the per-partition rack band. */
static int db_partition_g(const port_topic *t, int p, const int32_t *a, const int32_t *l, const int32_t *g, int *S, int32_t *val, int *Gout) {
    const int B = t->n_brokers, R = t->n_racks, RF = t->rf, plo = t->prack_lo, phi = t->prack_hi;
    const uint16_t *cur = t->current + (size_t)p * t->rf_cur;
    db_set G; G.n = 0;
    int cnt[256]; memset(cnt, 0, sizeof(int) * (size_t)R);
    /* greedy: plo best followers of every rack first, then the best remaining under the cap; ties -> lowest b */
    for (int round = 0; round < RF; ++round) {
        const int forced_rack = round < R * plo ? round / plo : -1;
        int bb = -1; int32_t bv = 0;
        for (int b = 0; b < B; ++b) {
            const int rb = t->rack_of[b];
            if (forced_rack >= 0 ? rb != forced_rack : cnt[rb] >= phi) continue;
            int in = 0;
            for (int j = 0; j < G.n; ++j) in |= G.b[j] == b;
            if (in) continue;
            const int32_t fv = (db_wcur(t, cur, (unsigned)b, 1) + (t->broker_w ? t->broker_w[b] : 0)) * DB_SCALE - a[b] - g[rb];
            if (bb < 0 || fv > bv) { bb = b; bv = fv; }
        }
        if (bb < 0) return -1;
        G.b[G.n] = bb; G.f[G.n] = bv; G.r[G.n] = t->rack_of[bb]; G.n++;
        cnt[t->rack_of[bb]]++;
    }
    int32_t fG = 0;
    for (int j = 0; j < RF; ++j) fG += G.f[j];
    if (Gout) for (int j = 0; j < RF; ++j) Gout[j] = G.b[j];
    /* leader: every broker b0; outside G it displaces the cheapest element whose removal keeps the rack band */
    int best_b0 = -1, best_e = -1; int32_t best_v = 0;
    for (int b0 = 0; b0 < B; ++b0) {
        const int r0 = t->rack_of[b0];
        const int32_t lv = (db_wcur(t, cur, (unsigned)b0, 0) + (t->broker_w ? t->broker_w[b0] : 0) + (t->broker_wl ? t->broker_wl[b0] : 0)) * DB_SCALE
                           - a[b0] - g[r0] - l[b0];
        int pos = -1;
        for (int j = 0; j < RF; ++j) if (G.b[j] == b0) pos = j;
        int e = -1; int32_t v;
        if (pos >= 0) { v = fG - G.f[pos] + lv; e = pos; }
        else {
            for (int j = 0; j < RF; ++j) {
                const int ok = cnt[r0] >= phi ? G.r[j] == r0 : (G.r[j] == r0 || cnt[G.r[j]] > plo);
                if (ok && (e < 0 || G.f[j] <= G.f[e])) e = j;   /* cheapest; ties -> the latest picked */
            }
            if (e < 0) continue;
            v = fG - G.f[e] + lv;
        }
        if (best_b0 < 0 || v > best_v) { best_b0 = b0; best_v = v; best_e = e; }
    }
    if (best_b0 < 0) return -1;
    int n = 0;
    S[n++] = best_b0;
    for (int j = 0; j < RF; ++j) if (j != best_e) S[n++] = G.b[j];
    *val = best_v;
    return 0;
}

/* Test hook: the brute-force subproblem solution of one partition (S: leader first; G: the greedy follower set the
 * leader was then fitted into) under given multipliers. */
int kao_port_dual_partition(const port_topic *t, int p, const int32_t *a, const int32_t *l, const int32_t *g, int32_t *S, int32_t *G,
                            int32_t *val) {
    int s[RFP], gg[RFP];
    const int rc = db_partition_g(t, p, a, l, g, s, val, gg);
    if (!rc) for (int j = 0; j < t->rf; ++j) { S[j] = s[j]; G[j] = gg[j]; }
    return rc;
}

static int db_partition(const port_topic *t, int p, const int32_t *a, const int32_t *l, const int32_t *g, int *S, int32_t *val) {
    return db_partition_g(t, p, a, l, g, S, val, NULL);
}

static inline int32_t db_sub(int32_t m, int n, int lo, int hi) {   /* element of the subdifferential closest to 0 */
    if (m > 0) return hi - n;
    if (m < 0) return lo - n;
    return n < lo ? lo - n : n > hi ? hi - n : 0;
}
static inline int32_t db_dir(int32_t d_prev, int32_t s, int k) {  /* d = 2^(6-k) s + floor((2^k - 1) d_prev / 2^k) */
    return s * (1 << (6 - k)) + (int32_t)(((int64_t)d_prev * ((1 << k) - 1)) >> k);
}
/* the 16 fractional bits of a move are rounded up with probability equal to the fraction (16 bits hashed from the step
 * number and the multiplier's index): truncation froze the iterate on large topics, see kao_bound.hip */
static inline uint32_t db_dither(uint32_t seq, uint32_t idx) {
    uint32_t h = seq * 0x9E3779B1u + idx * 0x85EBCA77u + 0x68E31DA4u;
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h >> 16;
}
/* Exact line search along the COMMON SHIFT of one family of multipliers (round 3).  Adding c to every multiplier of a family
 * leaves every subproblem solution alone (each partition pays c per replica / per leader), so
 *     L(m + c) = const + sum_i (m_i + c) * (m_i + c > 0 ? hi : lo) - c * total
 * is a convex piecewise-linear function of c alone, minimal where k = (total - n * lo) / (hi - lo) multipliers are positive:
 * the (k+1)-th largest becomes 0.  With hi == lo the shift changes nothing (tight bands: nothing to do).  Subgradient steps
 * are poor at this direction -- its kinks sit at every multiplier's zero crossing -- and on slack bands (P * RF not a multiple
 * of the broker count) the record stalled 12..20 units above the LP value (drifted 270 x 2200, LP optimum 16459: 16474.6
 * after 20,000 iterations; with the shift taken once per launch 16459 after 2,000).  Done at the start of every launch. */
static int db_cmp_desc(const void *x, const void *y) { const int32_t a = *(const int32_t *)x, b = *(const int32_t *)y; return a < b ? 1 : (a > b ? -1 : 0); }
static void db_center(int32_t *m, int n, int lo, int hi, int64_t total) {
    if (hi <= lo || n <= 1) return;
    int64_t k = (total - (int64_t)n * lo) / (hi - lo);
    if (k < 0) k = 0;
    if (k >= n) return;
    int32_t *tmp = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    memcpy(tmp, m, sizeof(int32_t) * (size_t)n);
    qsort(tmp, (size_t)n, sizeof(int32_t), db_cmp_desc);
    const int64_t pivot = tmp[k];
    free(tmp);
    for (int i = 0; i < n; ++i) {
        int64_t v = (int64_t)m[i] - pivot;
        if (v > DB_CLAMP) v = DB_CLAMP;
        if (v < -DB_CLAMP) v = -DB_CLAMP;
        m[i] = (int32_t)v;
    }
}
static inline int32_t db_move(int32_t m, int64_t step, int sh, int k, int32_t d, uint32_t h) {
    const int64_t mag = (step * (d < 0 ? -(int64_t)d : (int64_t)d) + ((int64_t)h << (sh - 22 + k))) >> (sh - 6 + k);
    int64_t v = (int64_t)m - (d < 0 ? -mag : mag);
    if (v > DB_CLAMP) v = DB_CLAMP;
    if (v < -DB_CLAMP) v = -DB_CLAMP;
    return (int32_t)v;
}

/* Runs up to `iters` dual iterations from the state (a[B], l[B], g[R] multipliers; da[B], dl[B], dg[R] previous
 * direction; lv[4] level-control state and step counter, zeros to start; *best_L), all in/out (zeros and INT64_MAX to start).  Needs P*RF <= 2^20, 4096 (P*RF)^2 (2B+R) < 2^62 and
 * P*RF*max(w) <= 2^25 (round 4; rounds 1-3: P*RF <= 2^17 -- the arithmetic below never needed that).  flags: 1 = closed (best_L < (target+1)*DB_SCALE), 2 = zero subgradient (dual optimum reached),
 * 4 = a partition subproblem is infeasible (no bound).  Returns the number of iterations performed. */
int kao_port_dual_bound_rec(const port_topic *t, int64_t target, int32_t iters, int32_t *a, int32_t *l, int32_t *g,
                            int32_t *da, int32_t *dl, int32_t *dg, int64_t *lv, int64_t *best_L, int32_t *flags,
                            int32_t *ra, int32_t *rl_, int32_t *rg_);
int kao_port_dual_bound(const port_topic *t, int64_t target, int32_t iters, int32_t *a, int32_t *l, int32_t *g,
                        int32_t *da, int32_t *dl, int32_t *dg, int64_t *lv, int64_t *best_L, int32_t *flags) {
    return kao_port_dual_bound_rec(t, target, iters, a, l, g, da, dl, dg, lv, best_L, flags, NULL, NULL, NULL);
}
/* ra / rl_ / rg_ (may be NULL): the multipliers at the record (smallest) dual value, in/out -- what the device exports,
 * rounded to the quarter grid, as search prices. */
int kao_port_dual_bound_rec(const port_topic *t, int64_t target, int32_t iters, int32_t *a, int32_t *l, int32_t *g,
                            int32_t *da, int32_t *dl, int32_t *dg, int64_t *lv, int64_t *best_L, int32_t *flags,
                            int32_t *ra, int32_t *rl_, int32_t *rg_) {
    const int B = t->n_brokers, R = t->n_racks, P = t->n_partitions, RF = t->rf, dk = db_defl(t->n_partitions);
    int32_t *nrep = (int32_t *)malloc(sizeof(int32_t) * (size_t)B), *nlead = (int32_t *)malloc(sizeof(int32_t) * (size_t)B);
    int32_t nrack[256];
    *flags = 0;
    int it = 0;
    if (iters > 0) {   /* the common shifts, once per launch (k_bound_center on the device) */
        db_center(a, B, t->rep_lo, t->rep_hi, (int64_t)P * RF);
        db_center(l, B, t->lead_lo, t->lead_hi, (int64_t)P);
        db_center(g, R, t->rack_lo, t->rack_hi, (int64_t)P * RF);
    }
    for (; it < iters; ++it) {
        memset(nrep, 0, sizeof(int32_t) * (size_t)B); memset(nlead, 0, sizeof(int32_t) * (size_t)B);
        memset(nrack, 0, sizeof nrack);
        int64_t L = 0;
        for (int p = 0; p < P; ++p) {
            int S[RFP]; int32_t v;
            if (db_partition(t, p, a, l, g, S, &v)) { *flags |= 4; goto done; }
            L += v;
            nlead[S[0]]++;
            for (int j = 0; j < RF; ++j) { nrep[S[j]]++; nrack[t->rack_of[S[j]]]++; }
        }
        /* band terms of L, subgradient s, new direction d = 16 s + floor(3 d_prev / 4) (kept even when this iteration stops) */
        int64_t nrm = 0, dn = 0;
        for (int b = 0; b < B; ++b) {
            L += (int64_t)a[b] * (a[b] > 0 ? t->rep_hi : t->rep_lo) + (int64_t)l[b] * (l[b] > 0 ? t->lead_hi : t->lead_lo);
            const int32_t sa = db_sub(a[b], nrep[b], t->rep_lo, t->rep_hi), sl = db_sub(l[b], nlead[b], t->lead_lo, t->lead_hi);
            nrm += (int64_t)sa * sa + (int64_t)sl * sl;
            da[b] = db_dir(da[b], sa, dk);
            dl[b] = db_dir(dl[b], sl, dk);
            dn += (int64_t)da[b] * da[b] + (int64_t)dl[b] * dl[b];
        }
        for (int r = 0; r < R; ++r) {
            L += (int64_t)g[r] * (g[r] > 0 ? t->rack_hi : t->rack_lo);
            const int32_t sg = db_sub(g[r], nrack[r], t->rack_lo, t->rack_hi);
            nrm += (int64_t)sg * sg;
            dg[r] = db_dir(dg[r], sg, dk);
            dn += (int64_t)dg[r] * dg[r];
        }
        if (L < *best_L) {
            *best_L = L;
            if (ra) { memcpy(ra, a, sizeof(int32_t) * (size_t)B); memcpy(rl_, l, sizeof(int32_t) * (size_t)B); memcpy(rg_, g, sizeof(int32_t) * (size_t)R); }
        }
        if (*best_L < (target + 1) * DB_SCALE) { *flags |= 1; ++it; break; }
        if (nrm == 0) { *flags |= 2; ++it; break; }
        if (dn == 0) {                                    /* the memory cancelled the subgradient: restart from it */
            for (int b = 0; b < B; ++b) {
                da[b] = db_sub(a[b], nrep[b], t->rep_lo, t->rep_hi) * (1 << (6 - dk));
                dl[b] = db_sub(l[b], nlead[b], t->lead_lo, t->lead_hi) * (1 << (6 - dk));
            }
            for (int r = 0; r < R; ++r) dg[r] = db_sub(g[r], nrack[r], t->rack_lo, t->rack_hi) * (1 << (6 - dk));
            dn = nrm << (2 * (6 - dk));
        }
        /* Level control: the Polyak step aims at `level` = record - delta, never below the incumbent `target`; delta starts
         * as the whole distance record -> incumbent (an incumbent below the optimum is an unreachable level: steps too long,
         * the record stalls far above the optimum).  Per stage of DB_STAGE iterations the record's gain is held against delta:
         * less than delta / 32 AND less than half a unit halves delta (floor 1/16), at least delta / 8 doubles it (never beyond
         * the incumbent): absolute while delta is large, relative once delta < 16.  The
         * thresholds are relative because the gain per stage is itself proportional to delta; an absolute one (half a unit, the
         * first version) fails every stage once delta < ~2.5 and delta collapses wherever the record stands.
         * The record that steers the level is the best value among the ITERATES (lv[3]), not *best_L, which the rounding probes
         * also lower (a probe value the iterate cannot reach soon would read as "no progress").
         * lv = {delta (0 = not started), record at the start of the stage, iterations in the stage | steps taken << 8, best iterate value}. */
        int64_t level = target * DB_SCALE;
        if (lv[0] <= 0) { lv[3] = L; lv[0] = lv[3] - level; lv[1] = lv[3]; lv[2] &= ~(int64_t)0xFF; }
        if (L < lv[3]) lv[3] = L;
        if (((++lv[2]) & 0xFF) >= DB_STAGE) {
            const int64_t prog = lv[1] - lv[3];
            if (prog < lv[0] / 32 && prog < DB_SCALE / 2) { lv[0] /= 2; if (lv[0] < DB_SCALE / 16) lv[0] = DB_SCALE / 16; }
            else if (prog >= lv[0] / 8 && lv[3] - 2 * lv[0] >= level) lv[0] *= 2;
            lv[1] = lv[3]; lv[2] &= ~(int64_t)0xFF;
        }
        if (lv[3] - lv[0] > level) level = lv[3] - lv[0];
        int64_t gap = L - level;
        if (gap < 1) gap = 1;
        /* multiplier change = gap * 2^(6-k) d / |d|^2; step = (gap << sh) / |d|^2 with as many bits as 62 allow, at most 40
         * (20 fixed bits gave a ZERO quotient once |d|^2 > 2^32 at the smallest gap; gap < 2^42 by the limits above) */
        int sh = __builtin_clzll((unsigned long long)gap) - 2;
        sh = sh > 40 ? 40 : (sh < 20 ? 20 : sh);
        const int64_t step = (gap << sh) / dn;
        const uint32_t seq = (uint32_t)(lv[2] >> 8);      /* steps taken so far, over all launches */
        lv[2] += 256;
        for (int b = 0; b < B; ++b) {
            a[b] = db_move(a[b], step, sh, dk, da[b], db_dither(seq, (uint32_t)b));
            l[b] = db_move(l[b], step, sh, dk, dl[b], db_dither(seq, (uint32_t)(B + b)));
        }
        for (int r = 0; r < R; ++r) g[r] = db_move(g[r], step, sh, dk, dg[r], db_dither(seq, (uint32_t)(2 * B + r)));
    }
    /* Rounding probes (only when the launch ran all its iterations): the dual function is also evaluated at the
     * multipliers rounded to the quarter grid and to the half grid -- optimal multipliers of this model tend to be small
     * fractions, and the rounded point hits them exactly while the subgradient iterate hovers around them.  A probe only
     * lowers *best_L (any multipliers give a valid bound); the iterate and the directions are left alone. */
    if (it == iters && !(*flags & 7)) {
        int32_t *qa = (int32_t *)malloc(sizeof(int32_t) * (size_t)B), *ql = (int32_t *)malloc(sizeof(int32_t) * (size_t)B);
        int32_t rg[256];
        for (int sh = DB_QUARTER_LOG2; sh <= DB_QUARTER_LOG2 + 1; ++sh) {
            for (int b = 0; b < B; ++b) { qa[b] = db_round(a[b], sh); ql[b] = db_round(l[b], sh); }
            for (int r = 0; r < R; ++r) rg[r] = db_round(g[r], sh);
            int64_t L = 0;
            int bad = 0;
            for (int p = 0; p < P && !bad; ++p) {
                int S[RFP]; int32_t v;
                if (db_partition(t, p, qa, ql, rg, S, &v)) bad = 1; else L += v;
            }
            if (bad) break;
            for (int b = 0; b < B; ++b)
                L += (int64_t)qa[b] * (qa[b] > 0 ? t->rep_hi : t->rep_lo) + (int64_t)ql[b] * (ql[b] > 0 ? t->lead_hi : t->lead_lo);
            for (int r = 0; r < R; ++r) L += (int64_t)rg[r] * (rg[r] > 0 ? t->rack_hi : t->rack_lo);
            if (L < *best_L) {
                *best_L = L;
                if (ra) { memcpy(ra, qa, sizeof(int32_t) * (size_t)B); memcpy(rl_, ql, sizeof(int32_t) * (size_t)B); memcpy(rg_, rg, sizeof(int32_t) * (size_t)R); }
            }
            if (*best_L < (target + 1) * DB_SCALE) *flags |= 1;
        }
        free(qa); free(ql);
    }
done:
    free(nrep); free(nlead);
    return it;
}
