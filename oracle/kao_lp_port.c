/* kao_lp_port.c -- CPU restatement (plain C) of KAO-LP, the device's interior-point solve of the compact LP relaxation --
 * TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg; never the product library).
 *
 * What it restates.  lp_solve proves the reference's optimum on the generated model (README.md:135-136, README.md:144-185).
 * Here the certificate is the exact Lagrangian dual value (oracle/kao_port.c::kao_port_dual_bound) at multipliers that come
 * from the LP relaxation of the same model in COMPACT form (rows and variables: oracle/kao_lp.py, which also holds the
 * generic-sparse restatement of the iteration and the HiGHS reference).  This file is the BLOCK-STRUCTURED iteration the
 * kernels of kafka_assignment_optimizer_amd/csrc/kao_lp.hip run: Mehrotra predictor-corrector, normal equations solved by
 *   (1) per partition: row C5[p,j] folded into a 2x2 weight of (f_j, l_j); rows C7[p,r] are then mutually orthogonal
 *       (diagonal d_r); the two dense rows C1[p], C2[p] leave a 2x2 system T;
 *   (2) the coupling rows NF[r], NL[r], C6[r], C3[b], C4[b] (3R + 2B of them) get the Schur complement S: per partition a
 *       block-diagonal part (per rack) minus a rank-2 term V' T^-1 V; dense Cholesky of S.
 * Same starting point, step rule and stopping rule as oracle/kao_lp.py::ipm; agreement is to rounding (different summation
 * order), not bit for bit.
 *
 * PARITY STATUS: parity unpinned beyond KAT-1 -- lp_solve 5.5 is absent; the LP values are checked against HiGHS.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define NONE16 0xFFFFu

typedef struct {
    int32_t n_brokers, n_racks, n_partitions, rf, rf_cur;
    const uint8_t *rack_of;
    const uint16_t *current;
    int32_t w[2][2];
    int32_t rep_lo, rep_hi, lead_lo, lead_hi, rack_lo, rack_hi, prack_lo, prack_hi;
    const int32_t *broker_w;
    const int32_t *broker_wl;
} port_topic;

typedef struct {
    int P, B, R, RF, NJ, NV, GV, mc;
    int has_c5, has_t, t_ub, has_n, has_m, has_k, n_ub, m_ub, k_ub;
    const uint16_t *cur;
    const uint8_t *rack;
    int *rsz;
    double *c, *cg;            /* costs (min form) [NV*P], [GV] */
    unsigned char *pres, *presg, *ub, *ubg;   /* variable present / has a finite upper bound */
    double *uu, *uug;          /* the upper bounds (0 where none) */
    double *bl_c1, *bl_c7, *bc; /* right-hand sides: C1 (RF), C7 (phi) scalars below; coupling rows [mc] */
    unsigned char *rowc;       /* coupling row present [mc] */
    /* iterate */
    double *x, *s, *v, *xg, *sg, *vg;
    double *y1, *y2, *y7, *y5, *yc;
    /* factor */
    double *th, *thg;
    double *fj;                /* per (j, p): sig11 sig12 sig22 e5 k1 k2 -> [6][NJ][P] */
    double *fr;                /* per (r, p): d e1 e2 -> [3][R][P] */
    double *ti;                /* T^-1: [3][P] */
    double *S;                 /* [mc*mc] lower Cholesky factor after lp_factor */
    double reg;
} lp_t;

#define XV(a, vv, p) ((a)[(size_t)(vv) * L->P + (p)])
#define VF(j) (3 * (j))
#define VL(j) (3 * (j) + 1)
#define VQ(j) (3 * (j) + 2)
#define VYF(r) (3 * L->NJ + 3 * (r))
#define VYL(r) (3 * L->NJ + 3 * (r) + 1)
#define VT(r) (3 * L->NJ + 3 * (r) + 2)
#define GZF(b) (b)
#define GZL(b) (L->B + (b))
#define GN(b) (2 * L->B + (b))
#define GM(b) (3 * L->B + (b))
#define GK(r) (4 * L->B + (r))
#define RNF(r) (r)
#define RNL(r) (L->R + (r))
#define RC6(r) (2 * L->R + (r))
#define RC3(b) (3 * L->R + 2 * (b))
#define RC4(b) (3 * L->R + 2 * (b) + 1)

static inline int cur_b(const lp_t *L, int p, int j) {
    const unsigned b = L->cur[(size_t)p * L->NJ + j];
    return (b == NONE16 || (int)b >= L->B) ? -1 : (int)b;
}

static void *zalloc(size_t n) { void *p = calloc(n ? n : 1, 1); if (!p) abort(); return p; }

/* Cost perturbation (round 5, KAO-LP's primal side).  The optimal face of the model's LP is huge -- ties everywhere -- and the
 * interior-point iterate converges to its analytic centre: fractional in most partitions.  Adding eps * h(i) to the cost of every
 * variable i, h in [0, 1) a hash of the variable's index and a salt, leaves (generically) ONE optimal vertex, the iterate converges
 * to it, and on every topic tried that vertex is integral in all but a handful of partitions: rounding it (oracle/kao_lp.py
 * round_primal) gives an assignment whose objective equals the LP value of the unperturbed model -- the optimum -- on 300 x 2000,
 * 450 x 3500, 500 x 5000, 500 x 10000 (docs/notes_r05.md section 6).  Index: v * P + p for the partition variables, 0x80000000 + g for
 * the global ones.  The stopping rule keeps the norm of the UNPERTURBED costs. */
static double pert_hash(uint32_t i, uint32_t salt) {
    uint32_t h = (i ^ salt) * 0x9E3779B1u + 0x85EBCA6Bu;
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return (double)(h >> 8) / 16777216.0;
}
static lp_t *lp_create(const port_topic *t) {
    lp_t *L = (lp_t *)zalloc(sizeof(lp_t));
    const int P = L->P = t->n_partitions, B = L->B = t->n_brokers, R = L->R = t->n_racks;
    L->RF = t->rf; const int NJ = L->NJ = t->rf_cur;
    L->NV = 3 * NJ + 3 * R; L->GV = 4 * B + R; L->mc = 3 * R + 2 * B;
    L->cur = t->current; L->rack = t->rack_of;
    L->has_c5 = t->prack_hi >= 2;
    L->has_t = t->prack_hi > t->prack_lo;
    L->t_ub = (L->has_t && t->prack_lo > 0) ? t->prack_hi - t->prack_lo : 0;
    /* the bands with their implied ends (oracle/kao_lp.py lp_bands; kao_lp.hip lp_open): B rep_lo = P RF pins every broker at rep_lo, ... */
    int32_t rep_lo = t->rep_lo, rep_hi = t->rep_hi, lead_lo = t->lead_lo, lead_hi = t->lead_hi, rack_lo = t->rack_lo, rack_hi = t->rack_hi;
    {
        const long long tot = (long long)P * t->rf;
        if ((long long)B * rep_lo == tot) rep_hi = rep_lo; else if ((long long)B * rep_hi == tot) rep_lo = rep_hi;
        if ((long long)B * lead_lo == P) lead_hi = lead_lo; else if ((long long)B * lead_hi == P) lead_lo = lead_hi;
        if ((long long)R * rack_lo == tot) rack_hi = rack_lo; else if ((long long)R * rack_hi == tot) rack_lo = rack_hi;
    }
    L->has_n = rep_hi > rep_lo; L->n_ub = rep_hi - rep_lo;
    L->has_m = lead_hi > lead_lo; L->m_ub = lead_hi - lead_lo;
    L->has_k = L->has_n && rack_hi > rack_lo; L->k_ub = rack_hi - rack_lo;
    L->rsz = (int *)zalloc(sizeof(int) * (size_t)R);
    for (int b = 0; b < B; ++b) L->rsz[t->rack_of[b]]++;
    const size_t nv = (size_t)L->NV * P;
    L->c = (double *)zalloc(8 * nv); L->cg = (double *)zalloc(8 * (size_t)L->GV);
    L->pres = (unsigned char *)zalloc(nv); L->presg = (unsigned char *)zalloc((size_t)L->GV);
    L->ub = (unsigned char *)zalloc(nv); L->ubg = (unsigned char *)zalloc((size_t)L->GV);
    L->uu = (double *)zalloc(8 * nv); L->uug = (double *)zalloc(8 * (size_t)L->GV);
    L->bc = (double *)zalloc(8 * (size_t)L->mc); L->rowc = (unsigned char *)zalloc((size_t)L->mc);
    for (int p = 0; p < P; ++p) {
        for (int j = 0; j < NJ; ++j) {
            const int b = cur_b(L, p, j);
            if (b < 0) continue;
            const int cr = j == 0 ? 0 : 1;
            const int bw = t->broker_w ? t->broker_w[b] : 0, bwl = t->broker_wl ? t->broker_wl[b] : 0;
            XV(L->pres, VF(j), p) = 1; XV(L->c, VF(j), p) = -(double)(t->w[cr][1] + bw);
            XV(L->pres, VL(j), p) = 1; XV(L->c, VL(j), p) = -(double)(t->w[cr][0] + bw + bwl);
            if (L->has_c5) XV(L->pres, VQ(j), p) = 1;
        }
        for (int r = 0; r < R; ++r) {
            XV(L->pres, VYF(r), p) = 1; XV(L->pres, VYL(r), p) = 1;
            if (L->has_t) { XV(L->pres, VT(r), p) = 1; if (L->t_ub) { XV(L->ub, VT(r), p) = 1; XV(L->uu, VT(r), p) = L->t_ub; } }
        }
    }
    for (int b = 0; b < B; ++b) {
        const int bw = t->broker_w ? t->broker_w[b] : 0, bwl = t->broker_wl ? t->broker_wl[b] : 0;
        L->presg[GZF(b)] = 1; L->cg[GZF(b)] = -(double)bw;
        L->presg[GZL(b)] = 1; L->cg[GZL(b)] = -(double)(bw + bwl);
        if (L->has_n) { L->presg[GN(b)] = 1; L->ubg[GN(b)] = 1; L->uug[GN(b)] = L->n_ub; }
        if (L->has_m) { L->presg[GM(b)] = 1; L->ubg[GM(b)] = 1; L->uug[GM(b)] = L->m_ub; }
        L->rowc[RC3(b)] = 1; L->bc[RC3(b)] = rep_lo;
        L->rowc[RC4(b)] = 1; L->bc[RC4(b)] = lead_lo;
    }
    for (int r = 0; r < R; ++r) {
        if (L->has_k) { L->presg[GK(r)] = 1; L->ubg[GK(r)] = 1; L->uug[GK(r)] = L->k_ub; }
        L->rowc[RNF(r)] = 1; L->rowc[RNL(r)] = 1;
        if (L->has_n) { L->rowc[RC6(r)] = 1; L->bc[RC6(r)] = (double)rack_lo - (double)L->rsz[r] * rep_lo; }
    }
    /* Exact row dependencies: without n (replicas per broker fixed) sum C1 = sum C3 + sum NF + sum NL, without m (leaders per
     * broker fixed) sum C2 = sum C4 + sum NL.  One row of each is redundant: NF[0] / NL[0] are pinned (dy = 0). */
    if (!L->has_n) L->rowc[RNF(0)] = 2;
    if (!L->has_m) L->rowc[RNL(0)] = 2;
    L->x = (double *)zalloc(8 * nv); L->s = (double *)zalloc(8 * nv); L->v = (double *)zalloc(8 * nv);
    L->xg = (double *)zalloc(8 * (size_t)L->GV); L->sg = (double *)zalloc(8 * (size_t)L->GV); L->vg = (double *)zalloc(8 * (size_t)L->GV);
    L->y1 = (double *)zalloc(8 * (size_t)P); L->y2 = (double *)zalloc(8 * (size_t)P);
    L->y7 = (double *)zalloc(8 * (size_t)P * R); L->y5 = (double *)zalloc(8 * (size_t)P * NJ);
    L->yc = (double *)zalloc(8 * (size_t)L->mc);
    L->th = (double *)zalloc(8 * nv); L->thg = (double *)zalloc(8 * (size_t)L->GV);
    L->fj = (double *)zalloc(8 * 6 * (size_t)NJ * P); L->fr = (double *)zalloc(8 * 3 * (size_t)R * P); L->ti = (double *)zalloc(8 * 3 * (size_t)P);
    L->S = (double *)zalloc(8 * (size_t)L->mc * L->mc);
    L->reg = 1e-10;
    return L;
}

static void lp_destroy(lp_t *L) {
    free(L->rsz); free(L->c); free(L->cg); free(L->pres); free(L->presg); free(L->ub); free(L->ubg); free(L->uu); free(L->uug);
    free(L->bc); free(L->rowc); free(L->x); free(L->s); free(L->v); free(L->xg); free(L->sg); free(L->vg);
    free(L->y1); free(L->y2); free(L->y7); free(L->y5); free(L->yc); free(L->th); free(L->thg); free(L->fj); free(L->fr); free(L->ti); free(L->S);
    free(L);
}

/* rows of A x: local rows into r1[P], r2[P], r7[R*P], r5[NJ*P]; coupling rows into rc[mc] */
static void lp_A(const lp_t *L, const double *x, const double *xg, double *r1, double *r2, double *r7, double *r5, double *rc) {
    const int P = L->P, R = L->R, NJ = L->NJ, B = L->B;
    memset(rc, 0, 8 * (size_t)L->mc);
    for (int p = 0; p < P; ++p) {
        double a1 = 0, a2 = 0;
        for (int r = 0; r < R; ++r) {
            const double yf = XV(x, VYF(r), p), yl = XV(x, VYL(r), p), tt = L->has_t ? XV(x, VT(r), p) : 0.0;
            a1 += yf + yl; a2 += yl;
            r7[(size_t)r * P + p] = yf + yl + tt;
            rc[RNF(r)] += yf; rc[RNL(r)] += yl;
        }
        for (int j = 0; j < NJ; ++j) {
            r5[(size_t)j * P + p] = 0;
            const int b = cur_b(L, p, j);
            if (b < 0) continue;
            const double f = XV(x, VF(j), p), l = XV(x, VL(j), p);
            a1 += f + l; a2 += l;
            r7[(size_t)L->rack[b] * P + p] += f + l;
            if (L->has_c5) r5[(size_t)j * P + p] = f + l + XV(x, VQ(j), p);
            rc[RC3(b)] += f + l; rc[RC4(b)] += l;
        }
        r1[p] = a1; r2[p] = a2;
    }
    for (int b = 0; b < B; ++b) {
        const int r = L->rack[b];
        rc[RC3(b)] += xg[GZF(b)] + xg[GZL(b)] - (L->has_n ? xg[GN(b)] : 0.0);
        rc[RC4(b)] += xg[GZL(b)] - (L->has_m ? xg[GM(b)] : 0.0);
        rc[RNF(r)] -= xg[GZF(b)]; rc[RNL(r)] -= xg[GZL(b)];
        if (L->has_n) rc[RC6(r)] += xg[GN(b)];
    }
    if (L->has_k) for (int r = 0; r < R; ++r) rc[RC6(r)] -= xg[GK(r)];
}

/* A^T y per variable (absent variables get 0) */
static void lp_AT(const lp_t *L, const double *y1, const double *y2, const double *y7, const double *y5, const double *yc, double *z, double *zg) {
    const int P = L->P, R = L->R, NJ = L->NJ, B = L->B;
    for (int p = 0; p < P; ++p) {
        for (int r = 0; r < R; ++r) {
            const double c7 = y7[(size_t)r * P + p];
            XV(z, VYF(r), p) = y1[p] + c7 + yc[RNF(r)];
            XV(z, VYL(r), p) = y1[p] + y2[p] + c7 + yc[RNL(r)];
            XV(z, VT(r), p) = L->has_t ? c7 : 0.0;
        }
        for (int j = 0; j < NJ; ++j) {
            const int b = cur_b(L, p, j);
            if (b < 0) { XV(z, VF(j), p) = XV(z, VL(j), p) = XV(z, VQ(j), p) = 0; continue; }
            const double c7 = y7[(size_t)L->rack[b] * P + p], c5 = L->has_c5 ? y5[(size_t)j * P + p] : 0.0;
            XV(z, VF(j), p) = y1[p] + c7 + c5 + yc[RC3(b)];
            XV(z, VL(j), p) = y1[p] + y2[p] + c7 + c5 + yc[RC3(b)] + yc[RC4(b)];
            XV(z, VQ(j), p) = c5;
        }
    }
    for (int b = 0; b < B; ++b) {
        const int r = L->rack[b];
        zg[GZF(b)] = yc[RC3(b)] - yc[RNF(r)];
        zg[GZL(b)] = yc[RC3(b)] + yc[RC4(b)] - yc[RNL(r)];
        zg[GN(b)] = L->has_n ? -yc[RC3(b)] + yc[RC6(r)] : 0.0;
        zg[GM(b)] = L->has_m ? -yc[RC4(b)] : 0.0;
    }
    for (int r = 0; r < R; ++r) zg[GK(r)] = L->has_k ? -yc[RC6(r)] : 0.0;
}

/* column list of one partition: coupling column, its rack, (m1, m2, eps) of the text above */
typedef struct { int col, rk, mate; double m1, m2, eps, dg, v0, v1; } pcol;   /* dg = the column's own diagonal entry, mate = index of the
                                                                                 other row of the same replica (its cross entry is sig12) */

static int lp_cols(const lp_t *L, int p, pcol *q) {
    const int P = L->P, R = L->R, NJ = L->NJ;
    int n = 0;
    for (int j = 0; j < NJ; ++j) {
        const int b = cur_b(L, p, j);
        if (b < 0) continue;
        const double s11 = L->fj[((size_t)0 * NJ + j) * P + p], s12 = L->fj[((size_t)1 * NJ + j) * P + p], s22 = L->fj[((size_t)2 * NJ + j) * P + p];
        q[n].col = RC3(b); q[n].rk = L->rack[b]; q[n].m1 = s11; q[n].m2 = s12; q[n].eps = s11; q[n].dg = s11; q[n].mate = n + 1; n++;
        q[n].col = RC4(b); q[n].rk = L->rack[b]; q[n].m1 = s12; q[n].m2 = s22; q[n].eps = s12; q[n].dg = s22; q[n].mate = n - 1; n++;
    }
    for (int r = 0; r < R; ++r) {
        const double cyf = XV(L->th, VYF(r), p), cyl = XV(L->th, VYL(r), p);
        q[n].col = RNF(r); q[n].rk = r; q[n].m1 = cyf; q[n].m2 = 0; q[n].eps = cyf; q[n].dg = cyf; q[n].mate = -1; n++;
        q[n].col = RNL(r); q[n].rk = r; q[n].m1 = cyl; q[n].m2 = cyl; q[n].eps = cyl; q[n].dg = cyl; q[n].mate = -1; n++;
    }
    for (int i = 0; i < n; ++i) {
        const int r = q[i].rk;
        const double d = L->fr[((size_t)0 * R + r) * P + p], e1 = L->fr[((size_t)1 * R + r) * P + p], e2 = L->fr[((size_t)2 * R + r) * P + p];
        q[i].v0 = q[i].m1 - e1 * q[i].eps / d;
        q[i].v1 = q[i].m2 - e2 * q[i].eps / d;
    }
    return n;
}

/* Lower Cholesky factor in place, row-oriented.  The coupling rows are linearly dependent when every band is tight (the C1 rows
 * add up to the C3 + NF + NL rows), and near the optimum S is numerically singular along further directions: a pivot that has
 * lost all but a 1e-12th of its diagonal entry marks a dependent row -- it gets a huge pivot, i.e. its dy component is held
 * at 0 (the usual treatment of rank-deficient normal equations in interior-point codes). */
#define LP_PIVOT_REL 1e-12
#define LP_PIVOT_BIG 1e64
static int cholesky(double *S, int n) {
    for (int i = 0; i < n; ++i) {
        double *Si = S + (size_t)i * n;
        const double orig = Si[i];
        for (int j = 0; j <= i; ++j) {
            const double *Sj = S + (size_t)j * n;
            double a = Si[j];
            for (int k = 0; k < j; ++k) a -= Si[k] * Sj[k];
            if (j < i) Si[j] = a / Sj[j];
            else Si[i] = (a > LP_PIVOT_REL * orig) ? sqrt(a) : LP_PIVOT_BIG;
        }
    }
    return 0;
}
static void chol_solve(const double *S, int n, double *b) {
    for (int i = 0; i < n; ++i) { const double *Si = S + (size_t)i * n; double a = b[i]; for (int k = 0; k < i; ++k) a -= Si[k] * b[k]; b[i] = a / Si[i]; }
    for (int i = n - 1; i >= 0; --i) { double a = b[i]; for (int k = i + 1; k < n; ++k) a -= S[(size_t)k * n + i] * b[k]; b[i] = a / S[(size_t)i * n + i]; }
}

/* theta (L->th, L->thg) -> per-partition factors, Schur complement of the coupling rows, its Cholesky factor */
static int lp_factor(lp_t *L) {
    const int P = L->P, R = L->R, NJ = L->NJ, B = L->B, mc = L->mc;
    double *S = L->S;
    memset(S, 0, 8 * (size_t)mc * mc);
    pcol *q = (pcol *)zalloc(sizeof(pcol) * (size_t)(2 * NJ + 2 * R));
    for (int p = 0; p < P; ++p) {
        double m11 = 0, m12 = 0, m22 = 0;
        for (int r = 0; r < R; ++r) {
            const double cyf = XV(L->th, VYF(r), p), cyl = XV(L->th, VYL(r), p), ct = L->has_t ? XV(L->th, VT(r), p) : 0.0;
            L->fr[((size_t)0 * R + r) * P + p] = cyf + cyl + ct + L->reg;
            L->fr[((size_t)1 * R + r) * P + p] = cyf + cyl;
            L->fr[((size_t)2 * R + r) * P + p] = cyl;
            m11 += cyf + cyl; m12 += cyl; m22 += cyl;
        }
        for (int j = 0; j < NJ; ++j) {
            const int b = cur_b(L, p, j);
            double s11 = 0, s12 = 0, s22 = 0, e5 = 1, k1 = 0, k2 = 0;
            if (b >= 0) {
                const double tf = XV(L->th, VF(j), p), tl = XV(L->th, VL(j), p);
                double a11 = tf, a12 = 0, a22 = tl;
                if (L->has_c5) {
                    e5 = tf + tl + XV(L->th, VQ(j), p) + L->reg; k1 = tf + tl; k2 = tl;
                    a11 = tf - tf * tf / e5; a12 = -tf * tl / e5; a22 = tl - tl * tl / e5;
                }
                s11 = a11 + 2 * a12 + a22; s12 = a12 + a22; s22 = a22;
                const int r = L->rack[b];
                L->fr[((size_t)0 * R + r) * P + p] += s11;
                L->fr[((size_t)1 * R + r) * P + p] += s11;
                L->fr[((size_t)2 * R + r) * P + p] += s12;
                m11 += s11; m12 += s12; m22 += s22;
            }
            L->fj[((size_t)0 * NJ + j) * P + p] = s11; L->fj[((size_t)1 * NJ + j) * P + p] = s12; L->fj[((size_t)2 * NJ + j) * P + p] = s22;
            L->fj[((size_t)3 * NJ + j) * P + p] = e5; L->fj[((size_t)4 * NJ + j) * P + p] = k1; L->fj[((size_t)5 * NJ + j) * P + p] = k2;
        }
        m11 += L->reg; m22 += L->reg;
        const double o11 = m11, o22 = m22;
        for (int r = 0; r < R; ++r) {
            const double d = L->fr[((size_t)0 * R + r) * P + p], e1 = L->fr[((size_t)1 * R + r) * P + p], e2 = L->fr[((size_t)2 * R + r) * P + p];
            m11 -= e1 * e1 / d; m12 -= e1 * e2 / d; m22 -= e2 * e2 / d;
        }
        /* T^-1 with guarded pivots: without slack in the C7 rows (plo == phi) they add up to C1, which is then a dependent row:
         * a pivot that lost all but 1e-12 of its entry pins that row's dy to 0 (as in the Cholesky of S) */
        double i11, i12, i22;
        if (!(m11 > LP_PIVOT_REL * o11)) { i11 = 0; i12 = 0; i22 = m22 > LP_PIVOT_REL * o22 ? 1.0 / m22 : 0.0; }
        else {
            const double l21 = m12 / m11, p2 = m22 - l21 * m12;
            if (!(p2 > LP_PIVOT_REL * o22)) { i11 = 1.0 / m11; i12 = 0; i22 = 0; }
            else { i22 = 1.0 / p2; i12 = -l21 * i22; i11 = 1.0 / m11 + l21 * l21 * i22; }
        }
        L->ti[(size_t)0 * P + p] = i11; L->ti[(size_t)1 * P + p] = i12; L->ti[(size_t)2 * P + p] = i22;
        const int n = lp_cols(L, p, q);
        for (int a = 0; a < n; ++a) {
            const double w0 = i11 * q[a].v0 + i12 * q[a].v1, w1 = i12 * q[a].v0 + i22 * q[a].v1;
            for (int c = 0; c < n; ++c) {
                if (q[c].col > q[a].col) continue;     /* lower triangle */
                double val = -(w0 * q[c].v0 + w1 * q[c].v1);
                if (q[a].rk == q[c].rk) {
                    const double d = L->fr[((size_t)0 * R + q[a].rk) * P + p];
                    val -= q[a].eps * q[c].eps / d;
                    if (a == c) val += q[a].dg;
                    else if (q[a].mate == c) val += q[a].col > q[c].col ? q[a].m1 : q[c].m1;   /* (C4j, C3j) = sig12 = m1 of the C4 column */
                }
                S[(size_t)q[a].col * mc + q[c].col] += val;
            }
        }
    }
    free(q);
    for (int b = 0; b < B; ++b) {
        const int r = L->rack[b];
        const double zf = L->thg[GZF(b)], zl = L->thg[GZL(b)], tn = L->has_n ? L->thg[GN(b)] : 0.0, tm = L->has_m ? L->thg[GM(b)] : 0.0;
        S[(size_t)RC3(b) * mc + RC3(b)] += zf + zl + tn;
        S[(size_t)RC4(b) * mc + RC3(b)] += zl;
        S[(size_t)RC4(b) * mc + RC4(b)] += zl + tm;
        S[(size_t)RC3(b) * mc + RNF(r)] -= zf;
        S[(size_t)RC3(b) * mc + RNL(r)] -= zl;
        S[(size_t)RC4(b) * mc + RNL(r)] -= zl;
        S[(size_t)RNF(r) * mc + RNF(r)] += zf;
        S[(size_t)RNL(r) * mc + RNL(r)] += zl;
        if (L->has_n) { S[(size_t)RC3(b) * mc + RC6(r)] -= tn; S[(size_t)RC6(r) * mc + RC6(r)] += tn; }
    }
    for (int r = 0; r < R; ++r) if (L->has_k) S[(size_t)RC6(r) * mc + RC6(r)] += L->thg[GK(r)];
    for (int i = 0; i < mc; ++i) {
        if (L->rowc[i] == 1) { S[(size_t)i * mc + i] += L->reg; continue; }
        for (int k = 0; k < i; ++k) S[(size_t)i * mc + k] = 0;          /* absent or pinned row: identity */
        for (int k = i + 1; k < mc; ++k) S[(size_t)k * mc + i] = 0;
        S[(size_t)i * mc + i] = 1.0;
    }
    return cholesky(S, mc);
}

/* N dy = rho: rho given as local rows (r1, r2, r7, r5) and coupling rows rc, all overwritten by dy */
static void lp_solve_normal(lp_t *L, double *r1, double *r2, double *r7, double *r5, double *rc) {
    const int P = L->P, R = L->R, NJ = L->NJ;
    pcol *q = (pcol *)zalloc(sizeof(pcol) * (size_t)(2 * NJ + 2 * R));
    for (int p = 0; p < P; ++p) {
        if (L->has_c5)
            for (int j = 0; j < NJ; ++j) {
                const int b = cur_b(L, p, j);
                if (b < 0) continue;
                const double g5 = r5[(size_t)j * P + p] / L->fj[((size_t)3 * NJ + j) * P + p];
                const double k1 = L->fj[((size_t)4 * NJ + j) * P + p], k2 = L->fj[((size_t)5 * NJ + j) * P + p];
                r1[p] -= k1 * g5; r7[(size_t)L->rack[b] * P + p] -= k1 * g5; rc[RC3(b)] -= k1 * g5;
                r2[p] -= k2 * g5; rc[RC4(b)] -= k2 * g5;
            }
        const int n = lp_cols(L, p, q);
        for (int r = 0; r < R; ++r) {
            const double d = L->fr[((size_t)0 * R + r) * P + p], e1 = L->fr[((size_t)1 * R + r) * P + p], e2 = L->fr[((size_t)2 * R + r) * P + p];
            const double g7 = r7[(size_t)r * P + p] / d;
            r1[p] -= e1 * g7; r2[p] -= e2 * g7;
            for (int a = 0; a < n; ++a) if (q[a].rk == r) rc[q[a].col] -= q[a].eps * g7;
        }
        const double i11 = L->ti[(size_t)0 * P + p], i12 = L->ti[(size_t)1 * P + p], i22 = L->ti[(size_t)2 * P + p];
        const double g1 = i11 * r1[p] + i12 * r2[p], g2 = i12 * r1[p] + i22 * r2[p];
        for (int a = 0; a < n; ++a) rc[q[a].col] -= q[a].v0 * g1 + q[a].v1 * g2;
    }
    for (int i = 0; i < L->mc; ++i) if (L->rowc[i] != 1) rc[i] = 0;
    chol_solve(L->S, L->mc, rc);
    for (int p = 0; p < P; ++p) {
        const int n = lp_cols(L, p, q);
        double t1 = r1[p], t2 = r2[p];
        for (int a = 0; a < n; ++a) { t1 -= q[a].v0 * rc[q[a].col]; t2 -= q[a].v1 * rc[q[a].col]; }
        const double i11 = L->ti[(size_t)0 * P + p], i12 = L->ti[(size_t)1 * P + p], i22 = L->ti[(size_t)2 * P + p];
        const double d1 = i11 * t1 + i12 * t2, d2 = i12 * t1 + i22 * t2;
        r1[p] = d1; r2[p] = d2;
        for (int r = 0; r < R; ++r) {
            const double d = L->fr[((size_t)0 * R + r) * P + p], e1 = L->fr[((size_t)1 * R + r) * P + p], e2 = L->fr[((size_t)2 * R + r) * P + p];
            double a7 = r7[(size_t)r * P + p] - e1 * d1 - e2 * d2;
            for (int a = 0; a < n; ++a) if (q[a].rk == r) a7 -= q[a].eps * rc[q[a].col];
            r7[(size_t)r * P + p] = a7 / d;
        }
        if (L->has_c5)
            for (int j = 0; j < NJ; ++j) {
                const int b = cur_b(L, p, j);
                if (b < 0) { r5[(size_t)j * P + p] = 0; continue; }
                const double k1 = L->fj[((size_t)4 * NJ + j) * P + p], k2 = L->fj[((size_t)5 * NJ + j) * P + p];
                r5[(size_t)j * P + p] = (r5[(size_t)j * P + p] - k1 * (d1 + r7[(size_t)L->rack[b] * P + p] + rc[RC3(b)]) - k2 * (d2 + rc[RC4(b)]))
                                        / L->fj[((size_t)3 * NJ + j) * P + p];
            }
    }
    free(q);
}

typedef struct { double *z, *zg, *r1, *r2, *r7, *r5, *rc; } lp_vec;   /* one variable-space + one row-space vector */
static lp_vec vec_new(const lp_t *L) {
    lp_vec v;
    v.z = (double *)zalloc(8 * (size_t)L->NV * L->P); v.zg = (double *)zalloc(8 * (size_t)L->GV);
    v.r1 = (double *)zalloc(8 * (size_t)L->P); v.r2 = (double *)zalloc(8 * (size_t)L->P);
    v.r7 = (double *)zalloc(8 * (size_t)L->P * L->R); v.r5 = (double *)zalloc(8 * (size_t)L->P * L->NJ); v.rc = (double *)zalloc(8 * (size_t)L->mc);
    return v;
}
static void vec_free(lp_vec *v) { free(v->z); free(v->zg); free(v->r1); free(v->r2); free(v->r7); free(v->r5); free(v->rc); }

/* Interior-point solve.  out_y[mc] = coupling-row duals (order NF[R] NL[R] C6[R] then C3[b], C4[b] interleaved),
 * trace[5 * (iterations + 1)] = (mu, pobj, dobj, pinf, dinf) per iteration (may be NULL), stats = {iterations, README
 * objective of the primal iterate, of the dual iterate, status (0 converged, 1 iteration limit, 2 Cholesky failed, 3 stalled)}. */
/* ---- multiple centrality correctors (Gondzio): after the predictor-corrector direction, the step lengths are enlarged by `MCC_DELTA`,
 * the complementarity products of that trial point are projected onto [MCC_BMIN, MCC_BMAX] x (sigma mu), and the difference is the
 * right-hand side of one more solve with the same factor; the corrected direction is kept when it lengthens a step. ---- */
#define SIGMA_EXP 10
#define SIGMA_EXP_HUGE 24          /* topics of more than SIGMA_HUGE_SLOTS replica slots (kao_lp.hip lp_sigexp) */
#define SIGMA_HUGE_SLOTS 131072
#define START_X_FLOOR 0.1
#define STEP_FRACTION 0.9
#define STEP_FRACTION_MAX 0.9995
#define MCC_DELTA 0.3
#define MCC_BMIN 0.1
#define MCC_BMAX 10.0
static void mcc_build(size_t n, const unsigned char *pres, const unsigned char *ub, const double *uu, const double *x, const double *s, const double *v,
                      const double *th, const double *dx, const double *ds, const double *dv, double apt, double adt, double mut,
                      double *h, double *g, double *rxs_out, double *rwv_out) {
    for (size_t i = 0; i < n; ++i) {
        if (!pres[i]) { h[i] = 0; g[i] = 0; rxs_out[i] = 0; rwv_out[i] = 0; continue; }
        double pr = (x[i] + apt * dx[i]) * (s[i] + adt * ds[i]);
        double tg = pr < MCC_BMIN * mut ? MCC_BMIN * mut : (pr > MCC_BMAX * mut ? MCC_BMAX * mut : pr);
        double rxs = tg - pr; if (rxs < -MCC_BMAX * mut) rxs = -MCC_BMAX * mut;
        double rwv = 0, w = 1;
        if (ub[i]) {
            w = uu[i] - x[i];
            pr = (w - apt * dx[i]) * (v[i] + adt * dv[i]);
            tg = pr < MCC_BMIN * mut ? MCC_BMIN * mut : (pr > MCC_BMAX * mut ? MCC_BMAX * mut : pr);
            rwv = tg - pr; if (rwv < -MCC_BMAX * mut) rwv = -MCC_BMAX * mut;
        }
        h[i] = -rxs / x[i] + (ub[i] ? rwv / w : 0.0);
        g[i] = th[i] * h[i];
        rxs_out[i] = rxs; rwv_out[i] = rwv;
    }
}
/* correction direction from A^T dy (in z) and the parked right-hand sides; then the step lengths of direction + correction */
static void mcc_finish(size_t n, const unsigned char *pres, const unsigned char *ub, const double *uu, const double *x, const double *s, const double *v,
                       const double *th, const double *h, double *z, double *dsc, double *dvc, const double *dx, const double *ds, const double *dv,
                       double *ap, double *ad) {
    for (size_t i = 0; i < n; ++i) {
        if (!pres[i]) { z[i] = 0; dsc[i] = 0; dvc[i] = 0; continue; }
        const double dxc = th[i] * (z[i] - h[i]);
        const double dss = (dsc[i] - s[i] * dxc) / x[i];
        z[i] = dxc; dsc[i] = dss;
        const double tx = dx[i] + dxc, ts = ds[i] + dss;
        if (tx < 0) { const double a = -x[i] / tx; if (a < *ap) *ap = a; }
        if (ts < 0) { const double a = -s[i] / ts; if (a < *ad) *ad = a; }
        if (ub[i]) {
            const double w = uu[i] - x[i], dvv = (dvc[i] + v[i] * dxc) / w;
            dvc[i] = dvv;
            const double tv = dv[i] + dvv;
            if (tx > 0) { const double a = w / tx; if (a < *ap) *ap = a; }
            if (tv < 0) { const double a = -v[i] / tv; if (a < *ad) *ad = a; }
        } else dvc[i] = 0;
    }
}

/* the same, also returning the primal iterate: out_x[(3 NJ + 3 R) * P] (variable-major: f_j l_j q_j per current replica, then yf_r yl_r t_r
 * per rack) and out_xg[4 B + R] (zf zl n m per broker, k per rack); either may be NULL */
int kao_lp_port_solve_p(const port_topic *t, double tol, int maxit, double eps, uint32_t salt, double *out_y, double *trace, double stats[4], double *out_x, double *out_xg);
int kao_lp_port_solve_x(const port_topic *t, double tol, int maxit, double *out_y, double *trace, double stats[4], double *out_x, double *out_xg) {
    return kao_lp_port_solve_p(t, tol, maxit, 0.0, 0u, out_y, trace, stats, out_x, out_xg);
}
int kao_lp_port_solve(const port_topic *t, double tol, int maxit, double *out_y, double *trace, double stats[4]) {
    return kao_lp_port_solve_p(t, tol, maxit, 0.0, 0u, out_y, trace, stats, NULL, NULL);
}
/* the same with perturbed costs: c_i + eps * pert_hash(i, salt) on every present variable (eps = 0: the model's own LP) */
int kao_lp_port_solve_p(const port_topic *t, double tol, int maxit, double eps, uint32_t salt, double *out_y, double *trace, double stats[4], double *out_x, double *out_xg) {
    lp_t *L = lp_create(t);
    const int P = L->P, R = L->R, NJ = L->NJ, mc = L->mc, GV = L->GV;
    const size_t nv = (size_t)L->NV * P;
    lp_vec rp = vec_new(L), rd = vec_new(L), h = vec_new(L), d1 = vec_new(L), d2 = vec_new(L), tmp = vec_new(L), dc = vec_new(L);
    double *dsc = (double *)zalloc(8 * nv), *dscg = (double *)zalloc(8 * (size_t)L->GV), *dvc = (double *)zalloc(8 * nv), *dvcg = (double *)zalloc(8 * (size_t)L->GV);
    int n_mcc = 0;
    double *dsa = (double *)zalloc(8 * nv), *dsag = (double *)zalloc(8 * (size_t)GV), *dva = (double *)zalloc(8 * nv), *dvag = (double *)zalloc(8 * (size_t)GV);
    double *ds = (double *)zalloc(8 * nv), *dsg = (double *)zalloc(8 * (size_t)GV), *dv = (double *)zalloc(8 * nv), *dvg = (double *)zalloc(8 * (size_t)GV);
    int status = 1, it = 0;
    double pobj = 0, dobj = 0, plast = 0, dlast = 0;
    double *ylast = (double *)zalloc(8 * (size_t)mc);
    long nvar = 0, nub = 0;
    for (size_t i = 0; i < nv; ++i) { nvar += L->pres[i]; nub += L->ub[i]; }
    for (int i = 0; i < GV; ++i) { nvar += L->presg[i]; nub += L->ubg[i]; }
    double nb = 0, ncn = 0;
    nb += (double)P * L->RF * L->RF + (double)P + (double)P * R * t->prack_hi * t->prack_hi + (L->has_c5 ? (double)P * 0 : 0);
    if (L->has_c5) for (int p = 0; p < P; ++p) for (int j = 0; j < NJ; ++j) if (cur_b(L, p, j) >= 0) nb += 1.0;
    for (int i = 0; i < mc; ++i) nb += L->bc[i] * L->bc[i];
    for (size_t i = 0; i < nv; ++i) ncn += L->c[i] * L->c[i];
    for (int i = 0; i < GV; ++i) ncn += L->cg[i] * L->cg[i];
    nb = 1.0 + sqrt(nb); ncn = 1.0 + sqrt(ncn);
    if (eps > 0) {
        for (size_t i = 0; i < nv; ++i) if (L->pres[i]) L->c[i] += eps * pert_hash((uint32_t)i, salt);
        for (int i = 0; i < GV; ++i) if (L->presg[i]) L->cg[i] += eps * pert_hash(0x80000000u + (uint32_t)i, salt);
    }
#define RHS_B(V) do { for (int p_ = 0; p_ < P; ++p_) { (V).r1[p_] = L->RF; (V).r2[p_] = 1; for (int r_ = 0; r_ < R; ++r_) (V).r7[(size_t)r_ * P + p_] = t->prack_hi; \
        for (int j_ = 0; j_ < NJ; ++j_) (V).r5[(size_t)j_ * P + p_] = (L->has_c5 && cur_b(L, p_, j_) >= 0) ? 1.0 : 0.0; } memcpy((V).rc, L->bc, 8 * (size_t)mc); } while (0)
    /* starting point: theta = 1; x = max(x~, START_X_FLOOR) capped at half the upper bound (kao_lp.hip k_lp_start; KAO_LP_XFLOOR: same hook) */
    double xfloor = getenv("KAO_LP_XFLOOR") ? atof(getenv("KAO_LP_XFLOOR")) : START_X_FLOOR;
    if (!(xfloor > 0.0 && xfloor <= 10.0)) xfloor = START_X_FLOOR;
    for (size_t i = 0; i < nv; ++i) L->th[i] = L->pres[i] ? 1.0 : 0.0;
    for (int i = 0; i < GV; ++i) L->thg[i] = L->presg[i] ? 1.0 : 0.0;
    if (lp_factor(L)) { status = 2; goto done; }
    RHS_B(tmp);
    lp_solve_normal(L, tmp.r1, tmp.r2, tmp.r7, tmp.r5, tmp.rc);
    lp_AT(L, tmp.r1, tmp.r2, tmp.r7, tmp.r5, tmp.rc, L->x, L->xg);            /* x~ = A^T (A A^T)^-1 b */
    lp_A(L, L->c, L->cg, tmp.r1, tmp.r2, tmp.r7, tmp.r5, tmp.rc);              /* A c */
    lp_solve_normal(L, tmp.r1, tmp.r2, tmp.r7, tmp.r5, tmp.rc);
    memcpy(L->y1, tmp.r1, 8 * (size_t)P); memcpy(L->y2, tmp.r2, 8 * (size_t)P); memcpy(L->y7, tmp.r7, 8 * (size_t)P * R);
    memcpy(L->y5, tmp.r5, 8 * (size_t)P * NJ); memcpy(L->yc, tmp.rc, 8 * (size_t)mc);
    lp_AT(L, L->y1, L->y2, L->y7, L->y5, L->yc, tmp.z, tmp.zg);
    for (size_t i = 0; i < nv; ++i) {
        if (!L->pres[i]) { L->x[i] = 1; L->s[i] = 1; L->v[i] = 0; continue; }
        double x = L->x[i] > xfloor ? L->x[i] : xfloor;
        if (L->ub[i]) { const double cap = L->uu[i] * 0.5 > 1e-2 ? L->uu[i] * 0.5 : 1e-2; if (x > cap) x = cap; }
        L->x[i] = x;
        const double s = L->c[i] - tmp.z[i];
        L->s[i] = s > 1.0 ? s : 1.0;
        L->v[i] = L->ub[i] ? 1.0 : 0.0;
    }
    for (int i = 0; i < GV; ++i) {
        if (!L->presg[i]) { L->xg[i] = 1; L->sg[i] = 1; L->vg[i] = 0; continue; }
        double x = L->xg[i] > xfloor ? L->xg[i] : xfloor;
        if (L->ubg[i]) { const double cap = L->uug[i] * 0.5 > 1e-2 ? L->uug[i] * 0.5 : 1e-2; if (x > cap) x = cap; }
        L->xg[i] = x;
        const double s = L->cg[i] - tmp.zg[i];
        L->sg[i] = s > 1.0 ? s : 1.0;
        L->vg[i] = L->ubg[i] ? 1.0 : 0.0;
    }
    double mu_ref = 1e300, pinf_best = 1e300;
    int it_ref = 0;
    for (it = 0;; ++it) {
        /* residuals */
        lp_A(L, L->x, L->xg, rp.r1, rp.r2, rp.r7, rp.r5, rp.rc);
        RHS_B(tmp);
        double pin = 0;
        for (int p = 0; p < P; ++p) {
            rp.r1[p] = tmp.r1[p] - rp.r1[p]; rp.r2[p] = tmp.r2[p] - rp.r2[p]; pin += rp.r1[p] * rp.r1[p] + rp.r2[p] * rp.r2[p];
            for (int r = 0; r < R; ++r) { const size_t k = (size_t)r * P + p; rp.r7[k] = tmp.r7[k] - rp.r7[k]; pin += rp.r7[k] * rp.r7[k]; }
            for (int j = 0; j < NJ; ++j) { const size_t k = (size_t)j * P + p; rp.r5[k] = tmp.r5[k] - rp.r5[k]; pin += rp.r5[k] * rp.r5[k]; }
        }
        for (int i = 0; i < mc; ++i) { rp.rc[i] = L->rowc[i] == 1 ? tmp.rc[i] - rp.rc[i] : 0.0; pin += rp.rc[i] * rp.rc[i]; }
        lp_AT(L, L->y1, L->y2, L->y7, L->y5, L->yc, tmp.z, tmp.zg);
        double din = 0, xs = 0, uv = 0;
        pobj = 0; dobj = 0;
        for (size_t i = 0; i < nv; ++i) {
            if (!L->pres[i]) { rd.z[i] = 0; continue; }
            rd.z[i] = L->c[i] - tmp.z[i] - L->s[i] + L->v[i]; din += rd.z[i] * rd.z[i];
            xs += L->x[i] * L->s[i]; pobj += L->c[i] * L->x[i];
            if (L->ub[i]) { xs += (L->uu[i] - L->x[i]) * L->v[i]; uv += L->uu[i] * L->v[i]; }
        }
        for (int i = 0; i < GV; ++i) {
            if (!L->presg[i]) { rd.zg[i] = 0; continue; }
            rd.zg[i] = L->cg[i] - tmp.zg[i] - L->sg[i] + L->vg[i]; din += rd.zg[i] * rd.zg[i];
            xs += L->xg[i] * L->sg[i]; pobj += L->cg[i] * L->xg[i];
            if (L->ubg[i]) { xs += (L->uug[i] - L->xg[i]) * L->vg[i]; uv += L->uug[i] * L->vg[i]; }
        }
        for (int p = 0; p < P; ++p) {
            dobj += L->RF * L->y1[p] + L->y2[p];
            for (int r = 0; r < R; ++r) dobj += t->prack_hi * L->y7[(size_t)r * P + p];
            if (L->has_c5) for (int j = 0; j < NJ; ++j) if (cur_b(L, p, j) >= 0) dobj += L->y5[(size_t)j * P + p];
        }
        for (int i = 0; i < mc; ++i) if (L->rowc[i]) dobj += L->bc[i] * L->yc[i];
        dobj -= uv;
        const double mu = xs / (double)(nvar + nub), pinf = sqrt(pin) / nb, dinf = sqrt(din) / ncn;
        if (trace) { trace[5 * it] = mu; trace[5 * it + 1] = pobj; trace[5 * it + 2] = dobj; trace[5 * it + 3] = pinf; trace[5 * it + 4] = dinf; }
        if (!(mu == mu) || !(pobj == pobj) || !(dobj == dobj)) { status = 3; break; }      /* the last finite iterate is what is returned */
        memcpy(ylast, L->yc, 8 * (size_t)mc); plast = pobj; dlast = dobj;
        const double gap = fabs(pobj - dobj) / (1.0 + fabs(pobj));
        if (gap < tol && pinf < 100 * tol && dinf < tol) { status = 0; break; }
        /* stalled at the numerical floor (kao_lp.hip k_lp_sc_resid, round 6): within 100 tolerances of the optimum and mu has not fallen
         * by a tenth in 8 iterations -> converged */
        /* past the floor (kao_lp.hip kLpFloorJump): it has been within 100 tolerances and its primal infeasibility is now 1000 x its smallest
         * value there -> converged */
        if (gap < 100.0 * tol && dinf < tol && pinf < pinf_best) pinf_best = pinf;
        if (pinf_best < 100 * tol && pinf > 100 * tol && pinf > 1e3 * pinf_best && dinf < tol) { status = 0; break; }
        if (it - it_ref >= 8) {
            if (gap < 100.0 * tol && pinf < 100 * tol && dinf < tol && mu > 0.9 * mu_ref) { status = 0; break; }
            mu_ref = mu; it_ref = it;
        }
        if (it >= maxit) { status = 1; break; }
        for (size_t i = 0; i < nv; ++i)
            L->th[i] = L->pres[i] ? 1.0 / (L->s[i] / L->x[i] + (L->ub[i] ? L->v[i] / (L->uu[i] - L->x[i]) : 0.0)) : 0.0;
        for (int i = 0; i < GV; ++i)
            L->thg[i] = L->presg[i] ? 1.0 / (L->sg[i] / L->xg[i] + (L->ubg[i] ? L->vg[i] / (L->uug[i] - L->xg[i]) : 0.0)) : 0.0;
        if (lp_factor(L)) { status = 2; break; }
        double ap = 1, ad = 1, sigma_mu = 0;
        for (int pass = 0; pass < 2; ++pass) {
            lp_vec *D = pass == 0 ? &d1 : &d2;
            double *pds = pass == 0 ? dsa : ds, *pdsg = pass == 0 ? dsag : dsg, *pdv = pass == 0 ? dva : dv, *pdvg = pass == 0 ? dvag : dvg;
            /* h = rd - rxs / x + rwv / w;  g = theta h */
            for (size_t i = 0; i < nv; ++i) {
                if (!L->pres[i]) { h.z[i] = 0; tmp.z[i] = 0; continue; }
                const double w = L->ub[i] ? L->uu[i] - L->x[i] : 1.0;
                double rxs = -L->x[i] * L->s[i], rwv = L->ub[i] ? -w * L->v[i] : 0.0;
                if (pass) { rxs += sigma_mu - d1.z[i] * dsa[i]; if (L->ub[i]) rwv += sigma_mu + d1.z[i] * dva[i]; }
                h.z[i] = rd.z[i] - rxs / L->x[i] + (L->ub[i] ? rwv / w : 0.0);
                tmp.z[i] = L->th[i] * h.z[i];
                pds[i] = rxs; pdv[i] = rwv;       /* parked: turned into ds, dv below */
            }
            for (int i = 0; i < GV; ++i) {
                if (!L->presg[i]) { h.zg[i] = 0; tmp.zg[i] = 0; continue; }
                const double w = L->ubg[i] ? L->uug[i] - L->xg[i] : 1.0;
                double rxs = -L->xg[i] * L->sg[i], rwv = L->ubg[i] ? -w * L->vg[i] : 0.0;
                if (pass) { rxs += sigma_mu - d1.zg[i] * dsag[i]; if (L->ubg[i]) rwv += sigma_mu + d1.zg[i] * dvag[i]; }
                h.zg[i] = rd.zg[i] - rxs / L->xg[i] + (L->ubg[i] ? rwv / w : 0.0);
                tmp.zg[i] = L->thg[i] * h.zg[i];
                pdsg[i] = rxs; pdvg[i] = rwv;
            }
            lp_A(L, tmp.z, tmp.zg, D->r1, D->r2, D->r7, D->r5, D->rc);
            for (int p = 0; p < P; ++p) {
                D->r1[p] += rp.r1[p]; D->r2[p] += rp.r2[p];
                for (int r = 0; r < R; ++r) D->r7[(size_t)r * P + p] += rp.r7[(size_t)r * P + p];
                for (int j = 0; j < NJ; ++j) D->r5[(size_t)j * P + p] += rp.r5[(size_t)j * P + p];
            }
            for (int i = 0; i < mc; ++i) D->rc[i] += rp.rc[i];
            lp_solve_normal(L, D->r1, D->r2, D->r7, D->r5, D->rc);       /* dy */
            lp_AT(L, D->r1, D->r2, D->r7, D->r5, D->rc, D->z, D->zg);
            ap = 1; ad = 1;
            long blk_p = -1, blk_d = -1; int blk_pk = 0, blk_dk = 0;
            for (size_t i = 0; i < nv; ++i) {
                if (!L->pres[i]) { D->z[i] = 0; pds[i] = 0; pdv[i] = 0; continue; }
                const double dx = L->th[i] * (D->z[i] - h.z[i]);
                const double dss = (pds[i] - L->s[i] * dx) / L->x[i];
                D->z[i] = dx; pds[i] = dss;
                if (dx < 0) { const double a = -L->x[i] / dx; if (a < ap) { ap = a; blk_p = (long)i; blk_pk = 0; } }
                if (dss < 0) { const double a = -L->s[i] / dss; if (a < ad) { ad = a; blk_d = (long)i; blk_dk = 0; } }
                if (L->ub[i]) {
                    const double w = L->uu[i] - L->x[i], dvv = (pdv[i] + L->v[i] * dx) / w;
                    pdv[i] = dvv;
                    if (dx > 0) { const double a = w / dx; if (a < ap) { ap = a; blk_p = (long)i; blk_pk = 1; } }
                    if (dvv < 0) { const double a = -L->v[i] / dvv; if (a < ad) { ad = a; blk_d = (long)i; blk_dk = 1; } }
                } else pdv[i] = 0;
            }
            if (getenv("KAO_LP_STEP_TRACE") && pass == 1) {
                fprintf(stderr, "[kao_lp_port]   local blockers: primal var class %ld (%s) x %.3e s %.3e ap %.4f | dual class %ld (%s) x %.3e s %.3e ad %.4f\n",
                        blk_p < 0 ? -1 : blk_p / P, blk_pk ? "w" : "x", blk_p < 0 ? 0 : L->x[blk_p], blk_p < 0 ? 0 : L->s[blk_p], ap,
                        blk_d < 0 ? -1 : blk_d / P, blk_dk ? "v" : "s", blk_d < 0 ? 0 : L->x[blk_d], blk_d < 0 ? 0 : L->s[blk_d], ad);
            }
            const double ap_loc = ap, ad_loc = ad;
            for (int i = 0; i < GV; ++i) {
                if (!L->presg[i]) { D->zg[i] = 0; pdsg[i] = 0; pdvg[i] = 0; continue; }
                const double dx = L->thg[i] * (D->zg[i] - h.zg[i]);
                const double dss = (pdsg[i] - L->sg[i] * dx) / L->xg[i];
                D->zg[i] = dx; pdsg[i] = dss;
                if (dx < 0) { const double a = -L->xg[i] / dx; if (a < ap) ap = a; }
                if (dss < 0) { const double a = -L->sg[i] / dss; if (a < ad) ad = a; }
                if (L->ubg[i]) {
                    const double w = L->uug[i] - L->xg[i], dvv = (pdvg[i] + L->vg[i] * dx) / w;
                    pdvg[i] = dvv;
                    if (dx > 0) { const double a = w / dx; if (a < ap) ap = a; }
                    if (dvv < 0) { const double a = -L->vg[i] / dvv; if (a < ad) ad = a; }
                } else pdvg[i] = 0;
            }
            if (getenv("KAO_LP_STEP_TRACE") && pass == 1 && (ap < ap_loc || ad < ad_loc)) fprintf(stderr, "[kao_lp_port]   global variables block: ap %.4f (local %.4f) ad %.4f (local %.4f)\n", ap, ap_loc, ad, ad_loc);
            (void)ap_loc; (void)ad_loc;
            if (pass == 0) {
                double xs2 = 0;
                for (size_t i = 0; i < nv; ++i) {
                    if (!L->pres[i]) continue;
                    xs2 += (L->x[i] + ap * d1.z[i]) * (L->s[i] + ad * dsa[i]);
                    if (L->ub[i]) xs2 += (L->uu[i] - L->x[i] - ap * d1.z[i]) * (L->v[i] + ad * dva[i]);
                }
                for (int i = 0; i < GV; ++i) {
                    if (!L->presg[i]) continue;
                    xs2 += (L->xg[i] + ap * d1.zg[i]) * (L->sg[i] + ad * dsag[i]);
                    if (L->ubg[i]) xs2 += (L->uug[i] - L->xg[i] - ap * d1.zg[i]) * (L->vg[i] + ad * dvag[i]);
                }
                const double mu_aff = xs2 / (double)(nvar + nub), ratio = mu_aff / mu;
                {   /* sigma = (mu_aff / mu)^SIGMA_EXP, the powers multiplied up one by one (kao_lp.hip k_lp_sc_sigma; KAO_LP_SIGEXP: same hook) */
                    const char *e = getenv("KAO_LP_SIGEXP");
                    const int dflt = (long long)t->n_partitions * t->rf > SIGMA_HUGE_SLOTS ? SIGMA_EXP_HUGE : SIGMA_EXP;
                    int ke = e ? atoi(e) : dflt;
                    if (ke < 1 || ke > 64) ke = dflt;
                    double sg = ratio;
                    for (int k = 1; k < ke; ++k) sg *= ratio;
                    sigma_mu = sg * mu;
                }
            }
        }
        {   /* up to two centrality correctors per iteration (the device enqueues exactly as many: KAO_LP_MCC=<k> overrides both sides, 0 = none) */
            const char *e = getenv("KAO_LP_MCC");
            const int kmax = e ? atoi(e) : 2;
            for (int k = 0; k < kmax && (ap < 1.0 || ad < 1.0); ++k) {
                const double apt = ap + MCC_DELTA < 1.0 ? ap + MCC_DELTA : 1.0, adt = ad + MCC_DELTA < 1.0 ? ad + MCC_DELTA : 1.0;
                mcc_build(nv, L->pres, L->ub, L->uu, L->x, L->s, L->v, L->th, d2.z, ds, dv, apt, adt, sigma_mu, h.z, tmp.z, dsc, dvc);
                mcc_build((size_t)GV, L->presg, L->ubg, L->uug, L->xg, L->sg, L->vg, L->thg, d2.zg, dsg, dvg, apt, adt, sigma_mu, h.zg, tmp.zg, dscg, dvcg);
                lp_A(L, tmp.z, tmp.zg, dc.r1, dc.r2, dc.r7, dc.r5, dc.rc);
                lp_solve_normal(L, dc.r1, dc.r2, dc.r7, dc.r5, dc.rc);
                lp_AT(L, dc.r1, dc.r2, dc.r7, dc.r5, dc.rc, dc.z, dc.zg);
                double ap2 = 1, ad2 = 1;
                mcc_finish(nv, L->pres, L->ub, L->uu, L->x, L->s, L->v, L->th, h.z, dc.z, dsc, dvc, d2.z, ds, dv, &ap2, &ad2);
                mcc_finish((size_t)GV, L->presg, L->ubg, L->uug, L->xg, L->sg, L->vg, L->thg, h.zg, dc.zg, dscg, dvcg, d2.zg, dsg, dvg, &ap2, &ad2);
                if (!(ap2 >= ap + 0.01 * MCC_DELTA || ad2 >= ad + 0.01 * MCC_DELTA) || ap2 < 0.9 * ap || ad2 < 0.9 * ad) break;
                for (size_t i = 0; i < nv; ++i) { d2.z[i] += dc.z[i]; ds[i] += dsc[i]; dv[i] += dvc[i]; }
                for (int i = 0; i < GV; ++i) { d2.zg[i] += dc.zg[i]; dsg[i] += dscg[i]; dvg[i] += dvcg[i]; }
                for (int p = 0; p < P; ++p) {
                    d2.r1[p] += dc.r1[p]; d2.r2[p] += dc.r2[p];
                    for (int r = 0; r < R; ++r) d2.r7[(size_t)r * P + p] += dc.r7[(size_t)r * P + p];
                    for (int j = 0; j < NJ; ++j) d2.r5[(size_t)j * P + p] += dc.r5[(size_t)j * P + p];
                }
                for (int i = 0; i < mc; ++i) d2.rc[i] += dc.rc[i];
                ap = ap2; ad = ad2;
                ++n_mcc;
            }
        }
        {   /* a blocked step goes a fraction min(STEP_FRACTION_MAX, max(STEP_FRACTION, alpha)) of the way to the boundary: short steps leave the
             * blocking pair a tenth of its value, nearly full ones stay nearly full (kao_lp.hip k_lp_sc_final; KAO_LP_GAMMA is the same
             * measurement hook: a fixed fraction) */
            const char *e = getenv("KAO_LP_GAMMA");
            const double gfix = e ? atof(e) : 0.0;
            if (ap < 1.0) ap *= (gfix > 0.5 && gfix < 1.0) ? gfix : (ap > STEP_FRACTION_MAX ? STEP_FRACTION_MAX : (ap < STEP_FRACTION ? STEP_FRACTION : ap));
            if (ad < 1.0) ad *= (gfix > 0.5 && gfix < 1.0) ? gfix : (ad > STEP_FRACTION_MAX ? STEP_FRACTION_MAX : (ad < STEP_FRACTION ? STEP_FRACTION : ad));
        }
        if (getenv("KAO_LP_STEP_TRACE")) fprintf(stderr, "[kao_lp_port] it %d mu %.3e sigma %.3e ap %.4f ad %.4f\n", it, mu, sigma_mu / mu, ap, ad);
        for (size_t i = 0; i < nv; ++i) { if (!L->pres[i]) continue; L->x[i] += ap * d2.z[i]; L->s[i] += ad * ds[i]; if (L->ub[i]) L->v[i] += ad * dv[i]; }
        for (int i = 0; i < GV; ++i) { if (!L->presg[i]) continue; L->xg[i] += ap * d2.zg[i]; L->sg[i] += ad * dsg[i]; if (L->ubg[i]) L->vg[i] += ad * dvg[i]; }
        for (int p = 0; p < P; ++p) {
            L->y1[p] += ad * d2.r1[p]; L->y2[p] += ad * d2.r2[p];
            for (int r = 0; r < R; ++r) L->y7[(size_t)r * P + p] += ad * d2.r7[(size_t)r * P + p];
            for (int j = 0; j < NJ; ++j) L->y5[(size_t)j * P + p] += ad * d2.r5[(size_t)j * P + p];
        }
        for (int i = 0; i < mc; ++i) L->yc[i] += ad * d2.rc[i];
    }
done:
    if (out_y) memcpy(out_y, ylast, 8 * (size_t)mc);
    if (out_x) memcpy(out_x, L->x, 8 * nv);
    if (out_xg) memcpy(out_xg, L->xg, 8 * (size_t)GV);
    if (stats) { stats[0] = it; stats[1] = -plast; stats[2] = -dlast; stats[3] = status; }
    free(ylast);
    free(dsa); free(dsag); free(dva); free(dvag); free(ds); free(dsg); free(dv); free(dvg);
    if (getenv("KAO_LP_MCC_TRACE")) fprintf(stderr, "[kao_lp_port] %d centrality correctors accepted in %d iterations\n", n_mcc, it);
    free(dsc); free(dscg); free(dvc); free(dvcg);
    vec_free(&rp); vec_free(&rd); vec_free(&h); vec_free(&d1); vec_free(&d2); vec_free(&tmp); vec_free(&dc);
    lp_destroy(L);
    return status;
}
