// kao_lp.hip -- KAO-LP: the LP relaxation of the Kafka partition-assignment model solved on gfx950 by a block-structured
// interior-point method; its row duals are the multipliers K-bound's exact dual value is taken at (round 5).
//
// lp_solve proves the reference's optimum on the generated model (README.md:135-136, README.md:144-185).  K-bound (kao_bound.hip)
// gives a certificate for ANY multipliers of the rows C3 / C4 / C6, but its subgradient iteration stalls above the LP value on
// slack-band and on large topics (450 x 3500: 26336 against an LP value of 26330 = the incumbent; 1000 x 30000: 231,562 against
// 231,532).  Here the multipliers come from the LP itself, in COMPACT form (oracle/kao_lp.py states it row by row): a variable
// of a broker that does not hold the partition today has objective coefficient 0 (README.md:145-146), so the new placements of
// a partition are pooled per rack (yf, yl) and handed to the brokers through per-broker inflow variables (zf, zl); the band
// rows are written on slack variables n, m, k.  What is reported never rests on floating point: the duals are rounded to
// K-bound's fixed point and the dual value there is computed by K-bound in integers.
//
// Iteration: Mehrotra predictor-corrector (oracle/kao_lp.py::ipm, restated on the block structure by oracle/kao_lp_port.c).
// Normal equations per iteration:
//   per partition (one lane each): row C5[p,j] is folded into a 2x2 weight of (f_j, l_j); the rows C7[p,r] are then mutually
//     orthogonal (a diagonal d_r); the dense rows C1[p], C2[p] leave a 2x2 system T (guarded pivots);
//   coupling rows NF[r] NL[r] C6[r] C3[b] C4[b] (mc = 3R + 2B): Schur complement S = per-partition block-diagonal parts minus
//     rank-2 terms, GATHERED in a fixed order (one wavefront per broker walks the broker's incidence list, the rack x rack
//     block is a tiled outer-product sum over fixed chunks): no floating-point atomics, the same bits on every run;
//   dense Cholesky of S and two triangular solves per right-hand side: kao_chol.hip (round 6: v_mfma_f64_16x16x4_f64, one launch per
//     64-row tile column; pivots that lost all but 1e-12 of their entry pin a dependent row).
// f64 throughout.  The dense contractions of the path -- the Cholesky and the rack x rack block of S -- run on the f64 matrix cores;
// everything else is gathers and per-partition streams.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <chrono>
#include <thread>
#include <vector>

#include "kao_host.h"

namespace kao {
namespace {

constexpr double kLpPivotRel = 1e-12;
constexpr double kLpReg = 1e-10;
constexpr int kNB = 64;            // Cholesky tile
constexpr int kRedVals = 8;        // values per reduction record
constexpr int kRedBlock = 256;
constexpr int kVarBlocks = 2048;   // workgroups (= records) of the reductions over the variables: every thread walks its elements with the grid's stride

struct LpDev {
    int P, B, R, NJ, RF, NV, GV, mc, mcp;    // mcp = mc rounded up to the Cholesky tile (padding rows are identity)
    // A SHARD of one LP (round 6, lp_open with an LpShard): the context holds the partitions p0 .. p0 + P - 1 of a topic of Pg partitions;
    // the coupling rows and the global variables are replicated on every shard, their own terms are counted by shard 0 only, and the
    // sums over the partitions meet through LpColl::allreduce.  A whole topic is the shard (p0 = 0, Pg = P, shard = 0).
    int Pg, p0, shard;
    int has_c5, has_t, t_ub, has_n, has_m, has_k, n_ub, m_ub, k_ub, phi;
    const uint16_t *cur;        // [P*NJ] dense broker or KAO_NONE
    const uint8_t *rack;        // [B]
    const int *inc_off, *inc;   // per broker: its incidences (p << 3 | j), ascending
    const int *rk_off, *rk_mem; // per rack: member brokers, ascending
    const double *c;            // [2*NJ][P]: cost (min form) of f_j, l_j
    const double *cg;           // [GV]
    const unsigned char *rowc;  // [mc]: 1 = row present, 0 = absent, 2 = pinned (exact dependency)
    const double *bc;           // [mc] right-hand sides of the coupling rows
    double *sc;                 // [kScN] the iteration's scalars, resident: no host round trip inside an iteration (see ScIdx)
};
// Device-resident scalars.  The kernels of an iteration read the step lengths, sigma mu and the stop flag from here, so any number
// of iterations can be enqueued without the host looking; once the stop flag is set every later kernel returns at its first line.
enum ScIdx { SC_STOP = 0 /* 0 run, 1 converged, 2 iteration limit, 3 stalled (non-finite iterate), 4 aborted by the host (lp_abort) */, SC_IT, SC_AP, SC_AD, SC_SIGMU, SC_MU, SC_POBJ, SC_DOBJ,
             SC_PINF, SC_DINF, SC_PLAST, SC_DLAST, SC_HAVE_LAST, SC_KEEP /* this iterate is finite: copy its duals */, SC_TOL, SC_MAXIT, SC_NVU /* variables + bounded variables */, SC_NB, SC_NCN,
             SC_PERT /* cost perturbation eps (0: the model's own LP) */, SC_SALT,
             SC_MCC_GO /* centrality correctors: the next one is still wanted */, SC_MCC_ACC /* the last one was accepted */,
             SC_MU_REF, SC_IT_REF /* the stall test's reference iterate */, SC_PINF_BEST /* smallest primal infeasibility among the near-optimal iterates */,
             SC_GAMMA /* the fraction of the way to the boundary a blocked step takes */, SC_SIGEXP /* sigma = (mu_aff / mu)^this */, kScN = 32 };
// Stalled at the numerical floor (round 6): some perturbed solves reach a relative gap of 5e-10 .. 1e-9 after ~110 iterations and then
// stand still -- step lengths ~0, mu unchanged for the remaining 90 iterations of their cap (profiles/r06_c09_stalled_solves.txt).  An
// iterate within kLpStallGap tolerances of the optimum whose mu has not fallen by a tenth in kLpStallWindow iterations counts as converged:
// its duals go through K-bound's integer evaluation like any others, its primal side through the rounding.
constexpr int kLpStallWindow = 8;
constexpr double kLpStallGap = 100.0, kLpStallMu = 0.9;
// Past the floor (round 6, later): on two racks (RF 3: every partition splits 1 + 2) the normal equations lose their conditioning once mu is
// ~1e-9 -- 600 x 50,000: relative gap 9e-10 and primal infeasibility 3e-9 at iteration 70, then the infeasibility jumps to 6e-5 and stays
// there for the remaining 130 iterations of the cap (the duals no longer move; 600 x 10,000 ends non-finite).  A solve that has been within
// kLpStallGap tolerances of the optimum and whose primal infeasibility is now kLpFloorJump times its smallest value there (and beyond what
// the stopping test accepts) stops as converged: later iterates are no better, and that one rounds like the one 130 iterations on.
constexpr double kLpFloorJump = 1e3;
#define LP_STOPPED(D) ((D).sc[SC_STOP] != 0.0)
// kernels of a centrality corrector carry gated = 1: once no further corrector is wanted they return at their first line
#define LP_GATED_OFF(D, gated) ((gated) && (D).sc[SC_MCC_GO] == 0.0)

// variable numbering inside a partition (SoA: element (v, p) at v*P + p) and among the global variables
__device__ __host__ __forceinline__ int VF(int j) { return 3 * j; }
__device__ __host__ __forceinline__ int VL(int j) { return 3 * j + 1; }
__device__ __host__ __forceinline__ int VQ(int j) { return 3 * j + 2; }
__device__ __host__ __forceinline__ int VYF(const LpDev &D, int r) { return 3 * D.NJ + 3 * r; }
__device__ __host__ __forceinline__ int VYL(const LpDev &D, int r) { return 3 * D.NJ + 3 * r + 1; }
__device__ __host__ __forceinline__ int VT(const LpDev &D, int r) { return 3 * D.NJ + 3 * r + 2; }
__device__ __host__ __forceinline__ int RNF(const LpDev &, int r) { return r; }
__device__ __host__ __forceinline__ int RNL(const LpDev &D, int r) { return D.R + r; }
__device__ __host__ __forceinline__ int RC6(const LpDev &D, int r) { return 2 * D.R + r; }
__device__ __host__ __forceinline__ int RC3(const LpDev &D, int b) { return 3 * D.R + 2 * b; }
__device__ __host__ __forceinline__ int RC4(const LpDev &D, int b) { return 3 * D.R + 2 * b + 1; }

__device__ __forceinline__ int cur_b(const LpDev &D, int p, int j) {
    const unsigned b = D.cur[(size_t)p * D.NJ + j];
    return (b == KAO_NONE || (int)b >= D.B) ? -1 : (int)b;
}
// presence / upper bound / cost of variable v of partition p
__device__ __forceinline__ bool var_present(const LpDev &D, int v, int p) {
    if (v < 3 * D.NJ) { const int k = v % 3; return cur_b(D, p, v / 3) >= 0 && (k < 2 || D.has_c5); }
    const int k = (v - 3 * D.NJ) % 3;
    return k < 2 || D.has_t;
}
__device__ __forceinline__ double var_ub(const LpDev &D, int v) {   // 0 = none
    return (v >= 3 * D.NJ && (v - 3 * D.NJ) % 3 == 2) ? (double)D.t_ub : 0.0;
}
// Cost perturbation (the primal side, round 5; oracle/kao_lp_port.c pert_hash): eps * h(i) on every present variable, h in [0, 1) a
// hash of the variable's index (v * P + p; 0x80000000 + g for the global ones) and a salt.  eps and salt live with the other scalars, so
// the captured iteration graph serves both the model's own LP (eps = 0) and the perturbed one.
__device__ __forceinline__ double pert_hash(uint32_t i, uint32_t salt) {
    uint32_t h = (i ^ salt) * 0x9E3779B1u + 0x85EBCA6Bu;
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return (double)(h >> 8) / 16777216.0;
}
__device__ __forceinline__ double pert_term(const LpDev &D, uint32_t i) {
    const double eps = D.sc[SC_PERT];
    return eps > 0.0 ? eps * pert_hash(i, (uint32_t)D.sc[SC_SALT]) : 0.0;
}
__device__ __forceinline__ double var_cost(const LpDev &D, int v, int p) {   // of a PRESENT variable
    const double pt = pert_term(D, (uint32_t)((size_t)v * D.Pg + D.p0 + p));
    if (v >= 3 * D.NJ) return pt;
    const int k = v % 3;
    return (k == 2 ? 0.0 : D.c[(size_t)(2 * (v / 3) + k) * D.P + p]) + pt;
}
__device__ __forceinline__ double gvar_cost(const LpDev &D, int g) { return D.cg[g] + pert_term(D, 0x80000000u + (uint32_t)g); }   // of a present one
__device__ __forceinline__ bool gvar_present(const LpDev &D, int g) {
    if (g < 2 * D.B) return true;
    if (g < 3 * D.B) return D.has_n;
    if (g < 4 * D.B) return D.has_m;
    return D.has_k;
}
__device__ __forceinline__ double gvar_ub(const LpDev &D, int g) {
    if (g < 2 * D.B) return 0.0;
    if (g < 3 * D.B) return D.n_ub;
    if (g < 4 * D.B) return D.m_ub;
    return D.k_ub;
}

struct RowVec { double *r1, *r2, *r7, *r5, *rc; };    // one vector over the rows: local C1[P] C2[P] C7[R][P] C5[NJ][P], coupling [mcp]
struct VarVec { double *z, *zg; };                    // one vector over the variables: [NV][P], [GV]

// (A^T y) of variable v of partition p / of global variable g
__device__ __forceinline__ double at_val(const LpDev &D, int v, int p, const RowVec &y) {
    const int P = D.P;
    if (v < 3 * D.NJ) {
        const int j = v / 3, k = v % 3, b = cur_b(D, p, j);
        if (b < 0) return 0.0;
        const double c5 = D.has_c5 ? y.r5[(size_t)j * P + p] : 0.0;
        if (k == 2) return c5;
        double a = y.r1[p] + y.r7[(size_t)D.rack[b] * P + p] + c5 + y.rc[RC3(D, b)];
        if (k == 1) a += y.r2[p] + y.rc[RC4(D, b)];
        return a;
    }
    const int r = (v - 3 * D.NJ) / 3, k = (v - 3 * D.NJ) % 3;
    const double c7 = y.r7[(size_t)r * P + p];
    if (k == 2) return D.has_t ? c7 : 0.0;
    if (k == 0) return y.r1[p] + c7 + y.rc[RNF(D, r)];
    return y.r1[p] + y.r2[p] + c7 + y.rc[RNL(D, r)];
}
__device__ __forceinline__ double at_val_g(const LpDev &D, int g, const double *yc) {
    const int B = D.B;
    if (g < B) return yc[RC3(D, g)] - yc[RNF(D, D.rack[g])];
    if (g < 2 * B) { const int b = g - B; return yc[RC3(D, b)] + yc[RC4(D, b)] - yc[RNL(D, D.rack[b])]; }
    if (g < 3 * B) { const int b = g - 2 * B; return D.has_n ? -yc[RC3(D, b)] + yc[RC6(D, D.rack[b])] : 0.0; }
    if (g < 4 * B) { const int b = g - 3 * B; return D.has_m ? -yc[RC4(D, b)] : 0.0; }
    return D.has_k ? -yc[RC6(D, g - 4 * B)] : 0.0;
}

// ---- deterministic block reductions: every block writes one record of kRedVals values, a second kernel adds the records in order
__device__ __forceinline__ void block_reduce(double *vals, int n, bool is_min, double *out_rec) {   // vals: this thread's n values
    __shared__ double sh[kRedBlock];
    for (int k = 0; k < n; ++k) {
        sh[threadIdx.x] = vals[k];
        __syncthreads();
        for (int s = kRedBlock / 2; s > 0; s >>= 1) {
            if ((int)threadIdx.x < s) sh[threadIdx.x] = is_min ? fmin(sh[threadIdx.x], sh[threadIdx.x + s]) : sh[threadIdx.x] + sh[threadIdx.x + s];
            __syncthreads();
        }
        if (threadIdx.x == 0) out_rec[k] = sh[0];
        __syncthreads();
    }
}
__global__ void __launch_bounds__(kRedBlock) k_lp_red_final(const double *sc, const double *rec, int nrec, int n, int is_min, double *out) {
    // one block: thread t folds records t, t + 256, ... in order, then the fixed tree of block_reduce -- the same bits on every run
    if (sc[SC_STOP] != 0.0) return;
    double a[kRedVals];
    for (int k = 0; k < n; ++k) a[k] = is_min ? 1.0 : 0.0;
    for (int i = threadIdx.x; i < nrec; i += kRedBlock)
        for (int k = 0; k < n; ++k) a[k] = is_min ? fmin(a[k], rec[(size_t)i * kRedVals + k]) : a[k] + rec[(size_t)i * kRedVals + k];
    block_reduce(a, n, is_min != 0, out);
}
// ---- the scalar steps of an iteration (one thread each) ------------------------------------------------------------------
// residual sums -> mu, objectives, infeasibilities, trace, the stopping test.  redA = {|rd|^2, x.s + w.v, c.x, u.v},
// redB = {|rp|^2, b.y} over the local rows, redC = the same over the coupling rows
__global__ void k_lp_sc_resid(double *sc, const double *redA, const double *redB, const double *redC, double *trace) {
    if (sc[SC_STOP] != 0.0) return;
    const double din = redA[0], xs = redA[1], cx = redA[2], uv = redA[3];
    const double pobj = cx, dobj = redB[1] + redC[1] - uv;
    const double mu = xs / sc[SC_NVU], pinf = sqrt(redB[0] + redC[0]) / sc[SC_NB], dinf = sqrt(din) / sc[SC_NCN];
    const int it = (int)sc[SC_IT];
    if (trace) { trace[5 * it] = mu; trace[5 * it + 1] = pobj; trace[5 * it + 2] = dobj; trace[5 * it + 3] = pinf; trace[5 * it + 4] = dinf; }
    sc[SC_MU] = mu; sc[SC_POBJ] = pobj; sc[SC_DOBJ] = dobj; sc[SC_PINF] = pinf; sc[SC_DINF] = dinf;
    const bool finite = isfinite(mu) && isfinite(pobj) && isfinite(dobj);
    sc[SC_KEEP] = finite ? 1.0 : 0.0;
    if (!finite) { sc[SC_STOP] = 3.0; return; }
    sc[SC_PLAST] = pobj; sc[SC_DLAST] = dobj; sc[SC_HAVE_LAST] = 1.0;     // k_lp_keep_last copies y next
    const double tol = sc[SC_TOL];
    const double gap = fabs(pobj - dobj) / (1.0 + fabs(pobj));
    if (gap < tol && pinf < 100 * tol && dinf < tol) { sc[SC_STOP] = 1.0; return; }
    if (gap < kLpStallGap * tol && dinf < tol && pinf < sc[SC_PINF_BEST]) sc[SC_PINF_BEST] = pinf;
    if (sc[SC_PINF_BEST] < 100 * tol && pinf > 100 * tol && pinf > kLpFloorJump * sc[SC_PINF_BEST] && dinf < tol) { sc[SC_STOP] = 1.0; return; }
    if (it - (int)sc[SC_IT_REF] >= kLpStallWindow) {
        if (gap < kLpStallGap * tol && pinf < 100 * tol && dinf < tol && mu > kLpStallMu * sc[SC_MU_REF]) { sc[SC_STOP] = 1.0; return; }
        sc[SC_MU_REF] = mu; sc[SC_IT_REF] = (double)it;
    }
    if (it >= (int)sc[SC_MAXIT]) sc[SC_STOP] = 2.0;
}
// the coupling-row duals of the last finite iterate (what the multipliers are read from); runs right after k_lp_sc_resid
__global__ void k_lp_keep_last(const double *sc, const double *yc, double *ylast, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && sc[SC_KEEP] == 1.0) ylast[i] = yc[i];
}
// step lengths to the boundary from the min-reduction; the last ones are damped (kLpGamma) and close the iteration count
__global__ void k_lp_sc_step(double *sc, const double *red, int pass) {
    if (sc[SC_STOP] != 0.0) return;
    const double ap = red[0], ad = red[1];
    sc[SC_AP] = ap; sc[SC_AD] = ad;
    if (pass) { sc[SC_MCC_GO] = (ap < 1.0 || ad < 1.0) ? 1.0 : 0.0; sc[SC_MCC_ACC] = 0.0; }
}
// after the correctors: the step stops short of the boundary, the iteration is counted
// A blocked step of length a goes min(kLpGammaMax, max(kLpGamma, a)) of the way to the boundary.  0.9995 throughout until late in round 6; a short
// step that stops a tenth short leaves the blocking pair a tenth of its value instead of a two-thousandth, and the next iteration is not
// blocked by the same pair again -- 16 drift seeds of the 1000 x 100,000 topic: fewer iterations of the perturbed solve on every one
// (profiles/r06_c30_step_fraction.txt); nearly full steps stay nearly full, so the last iterations converge as before.
// SC_GAMMA > 0 (KAO_LP_GAMMA, measurement hook): that fixed fraction instead.
constexpr double kLpGamma = 0.9, kLpGammaMax = 0.9995;
// sigma = (mu_aff / mu)^kLpSigmaExp.  Mehrotra's exponent 3 until late in round 6: on these LPs the predictor's step is short (0.1 .. 0.3: one pair blocks
// it) while the corrected step is not, so the cube asked for sigma = 0.5 .. 0.7 and mu fell by a tenth an iteration.  Exponent 10 keeps sigma
// near 1 only where the predictor achieves nothing: the perturbed solve of the drifted 1000 x 100,000 topic takes 66 iterations instead of
// 88, the certificate's LP 26 instead of 42 (profiles/r06_c31_sigma_exponent.txt; 6 / 8 / 10 / 12 / 16 / 24 / 32 tried: 24 stalled once).
constexpr int kLpSigmaExp = 10;
// The starting point's x = max(x~, kLpXFloor) (capped at half the upper bound).  1.0 until late in round 6: the partition variables live in [0, 1] and
// the unbounded ones (new-replica masses per rack, inflows) mostly far below 1; from 0.1 the perturbed solve of eight drift seeds takes 476
// iterations instead of 527 (0.3: 487, 0.03: 538; profiles/r06_c32_start_floor.txt).
constexpr double kLpXFloor = 0.1;
static double lp_xfloor() { const char *e = std::getenv("KAO_LP_XFLOOR"); const double f = e ? std::atof(e) : kLpXFloor; return f > 0.0 && f <= 10.0 ? f : kLpXFloor; }   // (measurement hook)
// Topics of more than kLpSigmaHugeSlots replica slots (the ones kao_solve gives their LP alone) take kLpSigmaExpHuge: with the starting point's floor at
// 0.1 the iteration count keeps falling up to ~24 there (13 drifted 100,000-partition topics: 761 iterations at 10, 663 at 14, 554 at 20, 508 at 24,
// 503-525 at 32-64; profiles/r06_c36_sigma_exponent_huge.txt) -- but on the small goldens 24 leaves 11 rounded iterates outside a band row instead of 4,
// so smaller topics keep 10.
constexpr int kLpSigmaExpHuge = 24;
constexpr long long kLpSigmaHugeSlots = 131072;
static double lp_sigexp(long long slots) {
    const int dflt = slots > kLpSigmaHugeSlots ? kLpSigmaExpHuge : kLpSigmaExp;
    const char *e = std::getenv("KAO_LP_SIGEXP"); const int k = e ? std::atoi(e) : dflt;      // (measurement hook)
    return k >= 1 && k <= 64 ? k : dflt;
}
static double lp_gamma() { const char *e = std::getenv("KAO_LP_GAMMA"); const double g = e ? std::atof(e) : 0.0; return g > 0.5 && g < 1.0 ? g : 0.0; }
__device__ __forceinline__ double lp_step_fraction(double a, double fixed) { return fixed > 0.0 ? fixed : (a > kLpGammaMax ? kLpGammaMax : (a < kLpGamma ? kLpGamma : a)); }
__global__ void k_lp_sc_final(double *sc) {
    if (sc[SC_STOP] != 0.0) return;
    if (sc[SC_AP] < 1.0) sc[SC_AP] *= lp_step_fraction(sc[SC_AP], sc[SC_GAMMA]);
    if (sc[SC_AD] < 1.0) sc[SC_AD] *= lp_step_fraction(sc[SC_AD], sc[SC_GAMMA]);
    sc[SC_IT] += 1.0;
}
// ---- Gondzio's multiple centrality correctors (round 5, last; oracle/kao_lp_port.c mcc_build / mcc_finish): the step lengths of the
// predictor-corrector direction are enlarged by kMccDelta, the complementarity products of that trial point are projected onto
// [kMccBmin, kMccBmax] x sigma mu, and the direction that moves them there is added when it lengthens the step.  The graph of an iteration
// is static: both correctors are always enqueued and turn into no-ops through SC_MCC_GO / SC_MCC_ACC.
constexpr double kMccDelta = 0.3, kMccBmin = 0.1, kMccBmax = 10.0;
__global__ void k_lp_sc_mcc(double *sc, const double *red) {
    if (sc[SC_STOP] != 0.0) return;
    if (sc[SC_MCC_GO] == 0.0) { sc[SC_MCC_ACC] = 0.0; return; }
    const double ap = sc[SC_AP], ad = sc[SC_AD], ap2 = red[0], ad2 = red[1];
    const bool bad = !(ap2 >= ap + 0.01 * kMccDelta || ad2 >= ad + 0.01 * kMccDelta) || ap2 < 0.9 * ap || ad2 < 0.9 * ad;
    if (bad) { sc[SC_MCC_ACC] = 0.0; sc[SC_MCC_GO] = 0.0; return; }
    sc[SC_MCC_ACC] = 1.0; sc[SC_AP] = ap2; sc[SC_AD] = ad2;
    if (!(ap2 < 1.0 || ad2 < 1.0)) sc[SC_MCC_GO] = 0.0;
}
__global__ void k_lp_sc_sigma(double *sc, const double *red) {
    if (sc[SC_STOP] != 0.0) return;
    const double ratio = red[0] / sc[SC_NVU] / sc[SC_MU];
    double sg = ratio;
    for (int k = 1; k < (int)sc[SC_SIGEXP]; ++k) sg *= ratio;
    sc[SC_SIGMU] = sg * sc[SC_MU];
}

// ---- elementwise over the variables ------------------------------------------------------------------------------
// theta = 1 / (s / x + v / w); init = 1: theta = 1 on present variables (the starting point's least-squares solves)
__global__ void k_lp_theta(LpDev D, const double *x, const double *s, const double *v, double *th, int init) {
    if (LP_STOPPED(D)) return;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)D.NV * D.P) return;
    const int vv = (int)(i / D.P), p = (int)(i % D.P);
    if (!var_present(D, vv, p)) { th[i] = 0.0; return; }
    if (init) { th[i] = 1.0; return; }
    const double u = var_ub(D, vv);
    th[i] = 1.0 / (s[i] / x[i] + (u > 0 ? v[i] / (u - x[i]) : 0.0));
}
__global__ void k_lp_theta_g(LpDev D, const double *x, const double *s, const double *v, double *th, int init) {
    if (LP_STOPPED(D)) return;
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= D.GV) return;
    if (!gvar_present(D, g)) { th[g] = 0.0; return; }
    if (init) { th[g] = 1.0; return; }
    const double u = gvar_ub(D, g);
    th[g] = 1.0 / (s[g] / x[g] + (u > 0 ? v[g] / (u - x[g]) : 0.0));
}

// ---- per-partition factor: sig11 sig12 sig22 e5 k1 k2 per replica, d e1 e2 per rack, T^-1 ------------------------------
// qd[3][P][ncp], qc[P][ncp], wr[4][NJ][P] (ncp = 0: not wanted): every coupling column of the partition as k_lp_schur_broker needs it
// (round 6) -- per column c (2j = C3 of replica j, 2j + 1 = C4, then NF[r], NL[r]): v0, v1, eps / d and row | rack << 16 (-1: the replica's
// broker is not in the target set) -- PARTITION-major, so that the 2 NJ + 2 R lanes of a broker's wavefront read one contiguous run per
// incidence (the kernel used to derive them per incidence through four dependent round trips of 8-byte gathers: 2.5 GB per launch at
// 100,000 partitions, each partition three times); and the two rows of replica j premultiplied by T^-1.
__global__ void k_lp_factor_local(LpDev D, const double *th, double *fj, double *fr, double *ti, double *__restrict__ qd, int *__restrict__ qc, double *__restrict__ wr, int ncp) {
    if (LP_STOPPED(D)) return;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= D.P) return;
    const int P = D.P, R = D.R, NJ = D.NJ;
    double m11 = 0, m12 = 0, m22 = 0;
    for (int r = 0; r < R; ++r) {
        const double cyf = th[(size_t)VYF(D, r) * P + p], cyl = th[(size_t)VYL(D, r) * P + p], ct = D.has_t ? th[(size_t)VT(D, r) * P + p] : 0.0;
        fr[((size_t)0 * R + r) * P + p] = cyf + cyl + ct + kLpReg;
        fr[((size_t)1 * R + r) * P + p] = cyf + cyl;
        fr[((size_t)2 * R + r) * P + p] = cyl;
        m11 += cyf + cyl; m12 += cyl; m22 += cyl;
    }
    for (int j = 0; j < NJ; ++j) {
        const int b = cur_b(D, p, j);
        double s11 = 0, s12 = 0, s22 = 0, e5 = 1, k1 = 0, k2 = 0;
        if (b >= 0) {
            const double tf = th[(size_t)VF(j) * P + p], tl = th[(size_t)VL(j) * P + p];
            double a11 = tf, a12 = 0, a22 = tl;
            if (D.has_c5) {
                e5 = tf + tl + th[(size_t)VQ(j) * P + p] + kLpReg; k1 = tf + tl; k2 = tl;
                a11 = tf - tf * tf / e5; a12 = -tf * tl / e5; a22 = tl - tl * tl / e5;
            }
            s11 = a11 + 2 * a12 + a22; s12 = a12 + a22; s22 = a22;
            const int r = D.rack[b];
            fr[((size_t)0 * R + r) * P + p] += s11;
            fr[((size_t)1 * R + r) * P + p] += s11;
            fr[((size_t)2 * R + r) * P + p] += s12;
            m11 += s11; m12 += s12; m22 += s22;
        }
        fj[((size_t)0 * NJ + j) * P + p] = s11; fj[((size_t)1 * NJ + j) * P + p] = s12; fj[((size_t)2 * NJ + j) * P + p] = s22;
        fj[((size_t)3 * NJ + j) * P + p] = e5; fj[((size_t)4 * NJ + j) * P + p] = k1; fj[((size_t)5 * NJ + j) * P + p] = k2;
    }
    m11 += kLpReg; m22 += kLpReg;
    const double o11 = m11, o22 = m22;
    for (int r = 0; r < R; ++r) {
        const double d = fr[((size_t)0 * R + r) * P + p], e1 = fr[((size_t)1 * R + r) * P + p], e2 = fr[((size_t)2 * R + r) * P + p];
        m11 -= e1 * e1 / d; m12 -= e1 * e2 / d; m22 -= e2 * e2 / d;
        // 1 / d, e1 / d, e2 / d for the eliminations and back substitutions of the iteration's solves (two to four of each: no division there)
        const double id = 1.0 / d;
        fr[((size_t)3 * R + r) * P + p] = id; fr[((size_t)4 * R + r) * P + p] = e1 * id; fr[((size_t)5 * R + r) * P + p] = e2 * id;
        if (ncp) {   // the rack's two columns NF[r], NL[r]
            const double cyf = th[(size_t)VYF(D, r) * P + p], cyl = th[(size_t)VYL(D, r) * P + p];
            const size_t pn = (size_t)P * ncp, kf = (size_t)p * ncp + 2 * NJ + 2 * r, kl = kf + 1;
            qd[kf] = cyf - cyf * (e1 * id); qd[pn + kf] = 0.0 - cyf * (e2 * id); qd[2 * pn + kf] = cyf * id; qc[kf] = RNF(D, r) | (r << 16);
            qd[kl] = cyl - cyl * (e1 * id); qd[pn + kl] = cyl - cyl * (e2 * id); qd[2 * pn + kl] = cyl * id; qc[kl] = RNL(D, r) | (r << 16);
        }
    }
    double i11, i12, i22;   // guarded pivots: C1 is a dependent row when the C7 rows carry no slack (oracle/kao_lp_port.c)
    if (!(m11 > kLpPivotRel * o11)) { i11 = 0; i12 = 0; i22 = m22 > kLpPivotRel * o22 ? 1.0 / m22 : 0.0; }
    else {
        const double l21 = m12 / m11, p2 = m22 - l21 * m12;
        if (!(p2 > kLpPivotRel * o22)) { i11 = 1.0 / m11; i12 = 0; i22 = 0; }
        else { i22 = 1.0 / p2; i12 = -l21 * i22; i11 = 1.0 / m11 + l21 * l21 * i22; }
    }
    ti[(size_t)0 * P + p] = i11; ti[(size_t)1 * P + p] = i12; ti[(size_t)2 * P + p] = i22;
    if (!ncp) return;
    const size_t cs = (size_t)P * ncp;      // stride between the three planes of qd
    for (int j = 0; j < NJ; ++j) {
        const int b = cur_b(D, p, j);
        const size_t k3 = (size_t)p * ncp + 2 * j, k4 = k3 + 1;
        if (b < 0) {
            qc[k3] = -1; qc[k4] = -1;
            qd[k3] = 0.0; qd[cs + k3] = 0.0; qd[2 * cs + k3] = 0.0; qd[k4] = 0.0; qd[cs + k4] = 0.0; qd[2 * cs + k4] = 0.0;
            for (int m = 0; m < 4; ++m) wr[((size_t)m * NJ + j) * P + p] = 0.0;
            continue;
        }
        const int rk = D.rack[b];
        const double s11 = fj[((size_t)0 * NJ + j) * P + p], s12 = fj[((size_t)1 * NJ + j) * P + p], s22 = fj[((size_t)2 * NJ + j) * P + p];
        const double d = fr[((size_t)0 * R + rk) * P + p], e1 = fr[((size_t)1 * R + rk) * P + p], e2 = fr[((size_t)2 * R + rk) * P + p];
        const double a3v0 = s11 - e1 * s11 / d, a3v1 = s12 - e2 * s11 / d, a4v0 = s12 - e1 * s12 / d, a4v1 = s22 - e2 * s12 / d;
        qd[k3] = a3v0; qd[cs + k3] = a3v1; qd[2 * cs + k3] = s11 / d; qc[k3] = RC3(D, b) | (rk << 16);
        qd[k4] = a4v0; qd[cs + k4] = a4v1; qd[2 * cs + k4] = s12 / d; qc[k4] = RC4(D, b) | (rk << 16);
        wr[((size_t)0 * NJ + j) * P + p] = i11 * a3v0 + i12 * a3v1; wr[((size_t)1 * NJ + j) * P + p] = i12 * a3v0 + i22 * a3v1;
        wr[((size_t)2 * NJ + j) * P + p] = i11 * a4v0 + i12 * a4v1; wr[((size_t)3 * NJ + j) * P + p] = i12 * a4v0 + i22 * a4v1;
    }
}

// One coupling column of a partition: which row of S, its rack, and (m1, m2, eps, dg) as in oracle/kao_lp_port.c::lp_cols.
// Column numbering inside a partition: 2j, 2j+1 = C3 / C4 of replica j (j < NJ), then 2NJ + 2r, 2NJ + 2r + 1 = NF[r] / NL[r].
struct PCol { int col, rk; double m1, m2, eps, dg, v0, v1; };
__device__ __forceinline__ bool lp_col(const LpDev &D, const double *th, const double *fj, const double *fr, int p, int c, PCol &q) {
    const int P = D.P, R = D.R, NJ = D.NJ;
    if (c < 2 * NJ) {
        const int j = c >> 1, b = cur_b(D, p, j);
        if (b < 0) return false;
        const double s11 = fj[((size_t)0 * NJ + j) * P + p], s12 = fj[((size_t)1 * NJ + j) * P + p], s22 = fj[((size_t)2 * NJ + j) * P + p];
        q.rk = D.rack[b];
        if (!(c & 1)) { q.col = RC3(D, b); q.m1 = s11; q.m2 = s12; q.eps = s11; q.dg = s11; }
        else { q.col = RC4(D, b); q.m1 = s12; q.m2 = s22; q.eps = s12; q.dg = s22; }
    } else {
        const int r = (c - 2 * NJ) >> 1;
        if (r >= R) return false;
        q.rk = r;
        if (!(c & 1)) { const double cyf = th[(size_t)VYF(D, r) * P + p]; q.col = RNF(D, r); q.m1 = cyf; q.m2 = 0; q.eps = cyf; q.dg = cyf; }
        else { const double cyl = th[(size_t)VYL(D, r) * P + p]; q.col = RNL(D, r); q.m1 = cyl; q.m2 = cyl; q.eps = cyl; q.dg = cyl; }
    }
    const double d = fr[((size_t)0 * R + q.rk) * P + p], e1 = fr[((size_t)1 * R + q.rk) * P + p], e2 = fr[((size_t)2 * R + q.rk) * P + p];
    q.v0 = q.m1 - e1 * q.eps / d;
    q.v1 = q.m2 - e2 * q.eps / d;
    return true;
}

// ---- Schur complement, broker rows: one wavefront per broker walks the broker's incidences in order -----------------------
// rows C3[b] and C4[b] are accumulated in LDS (2 x mc doubles per wavefront) and written once (lower triangle)
// NCL = columns per lane: 1 (up to 64 coupling columns per partition, 2 RF + 2 R: 29 racks at RF 3) or 2 (up to 128: 61 racks)
template <int U, int NCL = 1>
__global__ void __launch_bounds__(256) k_lp_schur_broker(LpDev D, const double *__restrict__ th, const double *__restrict__ thg, const double *__restrict__ fj,
                                                          const double *__restrict__ fr, const double *__restrict__ ti, const double *__restrict__ qd, const int *__restrict__ qc,
                                                          const double *__restrict__ wr, double *__restrict__ S) {
    if (LP_STOPPED(D)) return;
    extern __shared__ double lds_rows[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    // (the broker is the same for the whole wavefront: said so, the incidence words, the partition's factors and the incidence's own rows travel
    // through the scalar cache -- nine of the seventeen loads of an incidence)
    const int b = __builtin_amdgcn_readfirstlane(blockIdx.x * nw + wave);
    if (b >= D.B) return;
    const int P = D.P, R = D.R, NJ = D.NJ, mc = D.mc;
    double *rowA = lds_rows + (size_t)wave * 2 * mc, *rowB = rowA + mc;
    for (int i = lane; i < 2 * mc; i += 64) rowA[i] = 0.0;
    const int r0 = D.rack[b], nc = 2 * NJ + 2 * R;
    if (nc <= 64 * NCL) {
        // Every lane owns one column (NCL = 2: two) of an incidence; U incidences are in flight.  An incidence is TWO dependent round trips (round 6): the
        // incidence word (scalar), then everything else, which depends on the partition only -- the incidence's own rows premultiplied by
        // T^-1 and the replica columns come precomputed from k_lp_factor_local (wr, qd, qc), the rack columns are two products away from
        // theta and the reciprocal rack factors.  (Round 5 walked incidence word -> partition's factors -> rack of the column's broker ->
        // rack factors: four trips, 1.07 ms at 100,000 partitions with one wavefront per SIMD.)  The adds reach every column in incidence
        // order: the same bits on every run.
        const int e1 = D.inc_off[b + 1];
        bool has_col[NCL]; int cl[NCL];
#pragma unroll
        for (int n = 0; n < NCL; ++n) { has_col[n] = lane + 64 * n < nc; cl[n] = has_col[n] ? lane + 64 * n : 0; }
        const int ncp = (nc + 7) & ~7;
        const size_t cs = (size_t)P * ncp;
        for (int e = D.inc_off[b]; e < e1; e += U) {
            int pp[U], j0[U]; bool live[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { live[u] = e + u < e1; const int idx = D.inc[min(e + u, e1 - 1)]; pp[u] = idx >> 3; j0[u] = idx & 7; }
            double w30[U], w31[U], w40[U], w41[U], s11[U], s12[U], s22[U], q0[U][NCL], q1[U][NCL], qe[U][NCL];
            int col[U][NCL], rk[U][NCL];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int p = pp[u];
                w30[u] = wr[((size_t)0 * NJ + j0[u]) * P + p]; w31[u] = wr[((size_t)1 * NJ + j0[u]) * P + p];
                w40[u] = wr[((size_t)2 * NJ + j0[u]) * P + p]; w41[u] = wr[((size_t)3 * NJ + j0[u]) * P + p];
                s11[u] = fj[((size_t)0 * NJ + j0[u]) * P + p]; s12[u] = fj[((size_t)1 * NJ + j0[u]) * P + p]; s22[u] = fj[((size_t)2 * NJ + j0[u]) * P + p];
#pragma unroll
                for (int n = 0; n < NCL; ++n) {
                    const size_t k = (size_t)p * ncp + cl[n];      // consecutive lanes, consecutive columns of the partition
                    q0[u][n] = qd[k]; q1[u][n] = qd[cs + k]; qe[u][n] = qd[2 * cs + k];
                    const int w = qc[k];
                    col[u][n] = (w < 0 || !has_col[n]) ? -1 : (w & 0xFFFF); rk[u][n] = w < 0 ? -1 : (w >> 16);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int n = 0; n < NCL; ++n) {
                    if (!live[u] || col[u][n] < 0) continue;
                    const int c = lane + 64 * n;
                    double v3 = -(w30[u] * q0[u][n] + w31[u] * q1[u][n]), v4 = -(w40[u] * q0[u][n] + w41[u] * q1[u][n]);
                    if (rk[u][n] == r0) {
                        v3 -= s11[u] * qe[u][n]; v4 -= s12[u] * qe[u][n];                   // a3.eps = s11, a4.eps = s12
                        if (c == 2 * j0[u]) { v3 += s11[u]; v4 += s12[u]; }          // (C3, C3) = sig11, (C4, C3) = sig12
                        else if (c == 2 * j0[u] + 1) { v3 += s12[u]; v4 += s22[u]; } // (C3, C4) = sig12, (C4, C4) = sig22
                    }
                    rowA[col[u][n]] += v3; rowB[col[u][n]] += v4;
                }
        }
    } else
    for (int e = D.inc_off[b]; e < D.inc_off[b + 1]; ++e) {
        const int idx = D.inc[e], p = idx >> 3, j0 = idx & 7;
        const double i11 = ti[(size_t)0 * P + p], i12 = ti[(size_t)1 * P + p], i22 = ti[(size_t)2 * P + p];
        PCol a3, a4;
        lp_col(D, th, fj, fr, p, 2 * j0, a3);
        lp_col(D, th, fj, fr, p, 2 * j0 + 1, a4);
        const double w30 = i11 * a3.v0 + i12 * a3.v1, w31 = i12 * a3.v0 + i22 * a3.v1;
        const double w40 = i11 * a4.v0 + i12 * a4.v1, w41 = i12 * a4.v0 + i22 * a4.v1;
        const double d0 = fr[((size_t)0 * R + r0) * P + p];
        for (int c = lane; c < nc; c += 64) {
            PCol q;
            if (!lp_col(D, th, fj, fr, p, c, q)) continue;
            double v3 = -(w30 * q.v0 + w31 * q.v1), v4 = -(w40 * q.v0 + w41 * q.v1);
            if (q.rk == r0) {
                v3 -= a3.eps * q.eps / d0; v4 -= a4.eps * q.eps / d0;
                if (c == 2 * j0) { v3 += a3.dg; v4 += a4.m1; }            // (C3, C3) = sig11, (C4, C3) = sig12
                else if (c == 2 * j0 + 1) { v3 += a4.m1; v4 += a4.dg; }   // (C3, C4) = sig12, (C4, C4) = sig22
            }
            rowA[q.col] += v3; rowB[q.col] += v4;
        }
    }
    if (lane == 0 && D.shard == 0) {   // the broker's own global variables
        const double zf = thg[b], zl = thg[D.B + b], tn = D.has_n ? thg[2 * D.B + b] : 0.0, tm = D.has_m ? thg[3 * D.B + b] : 0.0;
        rowA[RC3(D, b)] += zf + zl + tn;
        rowB[RC3(D, b)] += zl;
        rowB[RC4(D, b)] += zl + tm;
        rowA[RNF(D, r0)] -= zf; rowA[RNL(D, r0)] -= zl; rowB[RNL(D, r0)] -= zl;
        if (D.has_n) rowA[RC6(D, r0)] -= tn;
    }
    const int ra = RC3(D, b), rb = RC4(D, b);
    for (int i = lane; i <= ra; i += 64) S[(size_t)ra * D.mcp + i] = rowA[i];
    for (int i = lane; i <= rb; i += 64) S[(size_t)rb * D.mcp + i] = rowB[i];
}

// ---- Schur complement, rack x rack block (rows NF, NL against columns NF, NL): tiled outer-product sum over a fixed chunk of
// partitions per block; partial sums per block, added in block order by k_lp_schur_rack_sum
__global__ void k_lp_schur_rack(LpDev D, const double *th, const double *fj, const double *fr, const double *ti, int chunk, int tile, double *part) {
    if (LP_STOPPED(D)) return;
    extern __shared__ double lds_t[];   // per partition of the tile: v0[2R] v1[2R] w0[2R] w1[2R] eps[2R] dg[2R] dinv[R]
    const int R = D.R, P = D.P, n2 = 2 * R, per = 6 * n2 + R;
    const int p0 = blockIdx.x * chunk, p1 = min(P, p0 + chunk);
    const int ne = n2 * n2;
    // every thread owns entries e = threadIdx.x + k * blockDim.x; at most 8 per thread are kept in registers per sweep
    for (int e0 = 0; e0 < ne; e0 += blockDim.x * 8) {
        double acc[8];
        for (int k = 0; k < 8; ++k) acc[k] = 0.0;
        for (int t0 = p0; t0 < p1; t0 += tile) {
            const int nt = min(tile, p1 - t0);
            __syncthreads();
            for (int i = threadIdx.x; i < nt * n2; i += blockDim.x) {
                const int tp = i / n2, c = i % n2, p = t0 + tp;      // column c: r = c % R, NF if c < R else NL
                const int r = c % R;
                const double cy = th[(size_t)(c < R ? VYF(D, r) : VYL(D, r)) * P + p];
                const double d = fr[((size_t)0 * R + r) * P + p], e1 = fr[((size_t)1 * R + r) * P + p], e2 = fr[((size_t)2 * R + r) * P + p];
                const double v0 = cy - e1 * cy / d, v1 = (c < R ? 0.0 : cy) - e2 * cy / d;
                const double i11 = ti[(size_t)0 * P + p], i12 = ti[(size_t)1 * P + p], i22 = ti[(size_t)2 * P + p];
                double *T = lds_t + (size_t)tp * per;
                T[c] = v0; T[n2 + c] = v1; T[2 * n2 + c] = i11 * v0 + i12 * v1; T[3 * n2 + c] = i12 * v0 + i22 * v1; T[4 * n2 + c] = cy; T[5 * n2 + c] = cy;
                if (c < R) T[6 * n2 + c] = 1.0 / d;
            }
            __syncthreads();
            for (int k = 0; k < 8; ++k) {
                const int e = e0 + threadIdx.x + k * blockDim.x;
                if (e >= ne) break;
                const int a = e / n2, c = e % n2;
                if (c > a) continue;
                const int ra = a % R, rc = c % R;
                double s = 0.0;
                for (int tp = 0; tp < nt; ++tp) {
                    const double *T = lds_t + (size_t)tp * per;
                    double v = -(T[2 * n2 + a] * T[c] + T[3 * n2 + a] * T[n2 + c]);
                    if (ra == rc) { v -= T[4 * n2 + a] * T[4 * n2 + c] * T[6 * n2 + ra]; if (a == c) v += T[5 * n2 + a]; }
                    s += v;
                }
                acc[k] += s;
            }
        }
        for (int k = 0; k < 8; ++k) {
            const int e = e0 + threadIdx.x + k * blockDim.x;
            if (e < ne) part[(size_t)blockIdx.x * ne + e] = acc[k];
        }
    }
}
__global__ void k_lp_schur_rack_sum(LpDev D, const double *part, int nblk, const double *thg, double *S) {
    if (LP_STOPPED(D)) return;
    const int n2 = 2 * D.R, ne = n2 * n2;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= ne) return;
    const int a = e / n2, c = e % n2;
    if (c > a) return;
    double s = 0.0;
    for (int i = 0; i < nblk; ++i) s += part[(size_t)i * ne + e];
    if (a == c && D.shard == 0) {   // the racks' own inflow variables: NF[r] += sum zf, NL[r] += sum zl over the rack's brokers (in order)
        const int r = a % D.R;
        for (int i = D.rk_off[r]; i < D.rk_off[r + 1]; ++i) s += thg[(a < D.R ? 0 : D.B) + D.rk_mem[i]];
    }
    S[(size_t)a * D.mcp + c] = s;    // rows NF[r] = r, NL[r] = R + r: exactly the numbering of a
}
// The same block on the matrix cores (round 6; 2R <= 128).  The block is a sum over the partitions of outer products,
//   S[a][c] = sum_p -(w0_a v0_c + w1_a v1_c) - [rack a == rack c] (cy_a / d) cy_c + [a == c] cy_a,      a, c in NF[0..R) NL[0..R),
// i.e. two (three with the same-rack term, masked afterwards) 2R x P x 2R products: v_mfma_f64_16x16x4_f64 with the partitions as the
// k index.  Lane (m, q) of a wavefront holds column 16 t + m of partition p0 + q for every 16-column tile t, which is at once the A operand
// of tile row t and the B operand of tile column t.  Every wavefront walks a fixed slice of the partitions, the four wavefronts of a
// workgroup are added in order through LDS, the workgroups' partial blocks by k_lp_schur_rack_sum2 in order: the same bits on every run.
typedef double lp_v4d __attribute__((ext_vector_type(4)));
constexpr int kRackMfmaBlocks = 256;
constexpr int kRackMfmaT16Max = 8;                          // 2R <= 128
constexpr int rack_mfma_record(int t16) { return 2 * (t16 * (t16 + 1) / 2) * 256 + t16 * 16; }   // doubles of one partial record: C tiles, same-rack tiles, column sums
// One workgroup's share: the tile rows TA0 .. TA1-1 (against the tile columns 0 .. ta) of the block.  Up to 64 columns (T16 <= 4) that is the
// whole lower triangle, ten tile pairs in 160 accumulator registers; up to 128 columns (33 .. 64 racks) the rows are dealt to four groups
// of workgroups (blockIdx.y: rows 0-3, 4-5, 6, 7 = 10, 11, 7, 8 pairs), every group walking all the partitions: the operands of the
// columns left of a group's rows are formed again by it (25 tile operands instead of 8 per four partitions; they are a quarter of a GB).
template <int T16, int TA0, int TA1>
__device__ __forceinline__ void rack_mfma_rows(const LpDev &D, const double *__restrict__ th, const double *__restrict__ fr, const double *__restrict__ ti, double *__restrict__ part, double *comb) {
    constexpr int NPALL = T16 * (T16 + 1) / 2;              // tile pairs (row tile >= column tile) of the whole block: the record's numbering
    constexpr int NE = rack_mfma_record(T16);
    constexpr int K0 = TA0 * (TA0 + 1) / 2, NP = TA1 * (TA1 + 1) / 2 - K0;   // this share's pairs: K0 .. K0 + NP - 1
    const int R = D.R, P = D.P, n2 = 2 * R;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, lm = lane & 15, lq = lane >> 4;
    const int nwaves = gridDim.x * 4, gw = blockIdx.x * 4 + w;
    const int chunk = ((P + nwaves - 1) / nwaves + 3) / 4 * 4;
    const int pb = min(P, gw * chunk), pe = min(P, pb + chunk);
    int vy[TA1], rr[TA1]; bool on[TA1], nl[TA1];
#pragma unroll
    for (int t = 0; t < TA1; ++t) {
        const int a = 16 * t + lm;
        on[t] = a < n2; nl[t] = a >= R;
        rr[t] = on[t] ? a % R : 0;
        vy[t] = nl[t] ? VYL(D, rr[t]) : VYF(D, rr[t]);
    }
    lp_v4d C[NP], G[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) { C[k] = lp_v4d{0.0, 0.0, 0.0, 0.0}; G[k] = lp_v4d{0.0, 0.0, 0.0, 0.0}; }
    double dsum[TA1 - TA0];
#pragma unroll
    for (int t = 0; t < TA1 - TA0; ++t) dsum[t] = 0.0;
    for (int p0 = pb; p0 < pe; p0 += 4) {
        const int p = p0 + lq;
        const bool live = p < pe;
        const int pc = live ? p : pb;
        const double i11 = ti[(size_t)0 * P + pc], i12 = ti[(size_t)1 * P + pc], i22 = ti[(size_t)2 * P + pc];
        double v0[TA1], v1[TA1], w0[TA1 - TA0], w1[TA1 - TA0], cy[TA1], g[TA1 - TA0];
#pragma unroll
        for (int t = 0; t < TA1; ++t) {
            const double c = th[(size_t)vy[t] * P + pc], d = fr[((size_t)0 * R + rr[t]) * P + pc], e1 = fr[((size_t)1 * R + rr[t]) * P + pc], e2 = fr[((size_t)2 * R + rr[t]) * P + pc];
            const bool use = live && on[t];
            cy[t] = use ? c : 0.0;
            v0[t] = use ? c - e1 * c / d : 0.0;
            v1[t] = use ? (nl[t] ? c : 0.0) - e2 * c / d : 0.0;
            if (t >= TA0) {
                g[t - TA0] = use ? c / d : 0.0;
                w0[t - TA0] = i11 * v0[t] + i12 * v1[t]; w1[t - TA0] = i12 * v0[t] + i22 * v1[t];
                dsum[t - TA0] += cy[t];
            }
        }
#pragma unroll
        for (int ta = TA0; ta < TA1; ++ta)
#pragma unroll
            for (int tc = 0; tc <= ta; ++tc) {
                const int k = ta * (ta + 1) / 2 + tc - K0;
                C[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(-w0[ta - TA0], v0[tc], C[k], 0, 0, 0);
                C[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(-w1[ta - TA0], v1[tc], C[k], 0, 0, 0);
                G[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(-g[ta - TA0], cy[tc], G[k], 0, 0, 0);
            }
    }
    // column sums: the four q-lanes of a column, in order
#pragma unroll
    for (int t = 0; t < TA1 - TA0; ++t) {
        const double s1 = __shfl(dsum[t], lm + 16, 64), s2 = __shfl(dsum[t], lm + 32, 64), s3 = __shfl(dsum[t], lm + 48, 64);
        dsum[t] = ((dsum[t] + s1) + s2) + s3;       // (meaningful on lanes 0..15)
    }
    // comb: [NP pairs x 256] C, [NP x 256] G, [(TA1 - TA0) x 16] column sums
    for (int ww = 0; ww < 4; ++ww) {               // the workgroup's four wavefronts, added in order
        if (w == ww) {
#pragma unroll
            for (int k = 0; k < NP; ++k)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int e = (k * 4 + r) * 64 + lane;
                    comb[e] = ww ? comb[e] + C[k][r] : C[k][r];
                    comb[NP * 256 + e] = ww ? comb[NP * 256 + e] + G[k][r] : G[k][r];
                }
            if (lane < 16)
#pragma unroll
                for (int t = 0; t < TA1 - TA0; ++t) comb[2 * NP * 256 + 16 * t + lane] = ww ? comb[2 * NP * 256 + 16 * t + lane] + dsum[t] : dsum[t];
        }
        __syncthreads();
    }
    double *rec = part + (size_t)blockIdx.x * NE;   // the record keeps the whole block's numbering: this share fills its own entries
    for (int e = threadIdx.x; e < NP * 256; e += 256) { rec[K0 * 256 + e] = comb[e]; rec[NPALL * 256 + K0 * 256 + e] = comb[NP * 256 + e]; }
    for (int e = threadIdx.x; e < (TA1 - TA0) * 16; e += 256) rec[2 * NPALL * 256 + TA0 * 16 + e] = comb[2 * NP * 256 + e];
}
constexpr int kRackMfmaComb = 2 * 11 * 256 + 4 * 16;        // the largest share: 11 pairs (rows 4-5 of eight), four tile rows of column sums
template <int T16>
__global__ void __launch_bounds__(256) k_lp_schur_rack_mfma(LpDev D, const double *__restrict__ th, const double *__restrict__ fr, const double *__restrict__ ti, double *__restrict__ part) {
    if (LP_STOPPED(D)) return;
    __shared__ double comb[T16 <= 4 ? 2 * (T16 * (T16 + 1) / 2) * 256 + T16 * 16 : kRackMfmaComb];
    if constexpr (T16 <= 4) rack_mfma_rows<T16, 0, T16>(D, th, fr, ti, part, comb);
    else {   // (blockIdx.y is uniform: one share per workgroup)
        if (blockIdx.y == 0) rack_mfma_rows<T16, 0, 4>(D, th, fr, ti, part, comb);
        else if (blockIdx.y == 1) rack_mfma_rows<T16, 4, (T16 < 6 ? T16 : 6)>(D, th, fr, ti, part, comb);
        else if constexpr (T16 >= 7) {
            if (blockIdx.y == 2) rack_mfma_rows<T16, 6, 7>(D, th, fr, ti, part, comb);
            else if constexpr (T16 >= 8) rack_mfma_rows<T16, 7, 8>(D, th, fr, ti, part, comb);
        }
    }
}
constexpr int rack_mfma_shares(int t16) { return t16 <= 4 ? 1 : t16 <= 6 ? 2 : t16 == 7 ? 3 : 4; }
// 16 entries x 16 slices of the partial records per workgroup: a thread adds its slice in record order, the slices are added in order
// through LDS (one thread per entry walking all 256 records took 0.16 ms: 25 wavefronts of dependent loads)
template <int T16>
__global__ void __launch_bounds__(256) k_lp_schur_rack_sum2(LpDev D, const double *__restrict__ part, int nblk, const double *__restrict__ thg, double *__restrict__ S) {
    if (LP_STOPPED(D)) return;
    constexpr int NP = T16 * (T16 + 1) / 2, NE = rack_mfma_record(T16);
    __shared__ double sl[16][17];
    const int n2 = 2 * D.R, ne = n2 * n2;
    const int el = threadIdx.x & 15, slice = threadIdx.x >> 4;
    const int e = blockIdx.x * 16 + el;
    const bool live = e < ne;
    const int a = live ? e / n2 : 0, c = live ? e % n2 : 0;
    const bool lower = live && c <= a;
    const int ta = a >> 4, tc = c >> 4, k = ta * (ta + 1) / 2 + tc, row = a & 15, col = c & 15;
    const int idx = (k * 4 + (row >> 2)) * 64 + (row & 3) * 16 + col;      // accumulator layout: row = (lane >> 4) + 4 reg, col = lane & 15
    const bool same = a % D.R == c % D.R;
    const int per = (nblk + 15) / 16, i0 = slice * per, i1 = min(nblk, i0 + per);
    double s = 0.0;
    if (lower)
        for (int i = i0; i < i1; ++i) {
            const double *rec = part + (size_t)i * NE;
            double v = rec[idx];
            if (same) { v += rec[NP * 256 + idx]; if (a == c) v += rec[2 * NP * 256 + a]; }
            s += v;
        }
    sl[el][slice] = s;
    __syncthreads();
    if (slice || !lower) return;
    s = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) s += sl[el][q];
    if (a == c && D.shard == 0) {   // the racks' own inflow variables: NF[r] += sum zf, NL[r] += sum zl over the rack's brokers (in order)
        const int r = a % D.R;
        for (int i = D.rk_off[r]; i < D.rk_off[r + 1]; ++i) s += thg[(a < D.R ? 0 : D.B) + D.rk_mem[i]];
    }
    S[(size_t)a * D.mcp + c] = s;
}
// A shard's part of S travels PACKED (round 6): the rows 0 .. mc-1 of the lower triangle, row i at offset i (i + 1) / 2 -- half the bytes of the
// square the all-reduce would otherwise carry (17 instead of 34 MB at 2,060 rows).  multigpu.py tri_pack / tri_unpack restate the index map.
__global__ void k_lp_tri_pack(LpDev D, const double *S, double *tri, int unpack, double *S_out) {
    if (LP_STOPPED(D)) return;
    const size_t n = (size_t)D.mc * (D.mc + 1) / 2;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        size_t i = (size_t)((sqrt(8.0 * (double)e + 1.0) - 1.0) * 0.5);
        while ((i + 1) * (i + 2) / 2 <= e) ++i;
        while (i * (i + 1) / 2 > e) --i;
        const size_t j = e - i * (i + 1) / 2;
        if (unpack) S_out[i * D.mcp + j] = tri[e]; else tri[e] = S[i * D.mcp + j];
    }
}
// rows C6[r] (columns NF, NL: none; own diagonal: sum n + k), regularisation, absent / pinned rows, padding; saves the diagonal.
// One wavefront per row: the rows that are cleared left of the diagonal (padding, C6, absent / pinned) are cleared by its 64 lanes -- one
// thread per row walked up to 2,111 entries of each of the 52 padding rows by itself, 71 us of every iteration at 2,060 coupling rows.
__global__ void __launch_bounds__(64) k_lp_schur_fix(LpDev D, const double *thg, double *S, double *diag0) {
    if (LP_STOPPED(D)) return;
    const int i = blockIdx.x, t = threadIdx.x;
    if (i >= D.mcp) return;
    const int R = D.R;
    double *row = S + (size_t)i * D.mcp;
    const bool c6 = i < D.mc && i >= 2 * R && i < 3 * R, live = i < D.mc && D.rowc[i] == 1;
    if (c6 || !live)
        for (int k = t; k < i; k += 64) row[k] = 0.0;
    if (t != 0) return;
    if (i >= D.mc) { row[i] = 1.0; diag0[i] = 1.0; return; }
    if (c6) {
        double s = 0.0;
        if (D.has_n) { const int r = i - 2 * R; for (int e = D.rk_off[r]; e < D.rk_off[r + 1]; ++e) s += thg[2 * D.B + D.rk_mem[e]]; if (D.has_k) s += thg[4 * D.B + r]; }
        row[i] = s;
    }
    if (live) row[i] += kLpReg;
    else row[i] = 1.0;
    diag0[i] = row[i];
}
// column of an absent / pinned row below the diagonal (only rack rows can be absent or pinned: i < 3R)
__global__ void k_lp_schur_fix_cols(LpDev D, double *S) {
    if (LP_STOPPED(D)) return;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= D.mc) return;
    for (int i = 0; i < 3 * D.R && i < k; ++i)
        if (D.rowc[i] != 1) S[(size_t)k * D.mcp + i] = 0.0;
}

// ---- rows of A z ------------------------------------------------------------------------------------------------------
// local rows; mode 0: out = A z, 1: out = b - A z, 2: out = A z + add
__global__ void k_lp_A_local(LpDev D, const double *__restrict__ z, RowVec out, int mode, RowVec add, int gated) {
    if (LP_STOPPED(D) || LP_GATED_OFF(D, gated)) return;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= D.P) return;
    const int P = D.P, R = D.R, NJ = D.NJ;
    double a1 = 0, a2 = 0;
    for (int r = 0; r < R; ++r) {
        const double yf = z[(size_t)VYF(D, r) * P + p], yl = z[(size_t)VYL(D, r) * P + p], tt = D.has_t ? z[(size_t)VT(D, r) * P + p] : 0.0;
        a1 += yf + yl; a2 += yl;
        out.r7[(size_t)r * P + p] = yf + yl + tt;
    }
    for (int j = 0; j < NJ; ++j) {
        const int b = cur_b(D, p, j);
        double v5 = 0.0;
        if (b >= 0) {
            const double f = z[(size_t)VF(j) * P + p], l = z[(size_t)VL(j) * P + p];
            a1 += f + l; a2 += l;
            out.r7[(size_t)D.rack[b] * P + p] += f + l;
            if (D.has_c5) v5 = f + l + z[(size_t)VQ(j) * P + p];
        }
        const bool row5 = D.has_c5 && b >= 0;
        out.r5[(size_t)j * P + p] = mode == 1 ? (row5 ? 1.0 - v5 : 0.0) : (mode == 2 ? v5 + add.r5[(size_t)j * P + p] : v5);
    }
    if (mode == 1) { a1 = D.RF - a1; a2 = 1.0 - a2; }
    if (mode == 2) { a1 += add.r1[p]; a2 += add.r2[p]; }
    out.r1[p] = a1; out.r2[p] = a2;
    if (mode)
        for (int r = 0; r < R; ++r) {
            const size_t k = (size_t)r * P + p;
            out.r7[k] = mode == 1 ? D.phi - out.r7[k] : out.r7[k] + add.r7[k];
        }
}
// coupling rows C3[b], C4[b]: one wavefront per broker; `cb` (may be null): extra per-incidence terms [2 NJ][P] of the eliminations
__global__ void k_lp_A_broker(LpDev D, const double *z, const double *zg, const double *cb, double *rc, int mode, const double *addc, int gated) {
    if (LP_STOPPED(D) || LP_GATED_OFF(D, gated)) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    const int b = blockIdx.x * nw + wave;
    if (b >= D.B) return;
    const int P = D.P;
    double s3 = 0, s4 = 0;
    for (int e = D.inc_off[b] + lane; e < D.inc_off[b + 1]; e += 64) {
        const int idx = D.inc[e], p = idx >> 3, j = idx & 7;
        const double f = z[(size_t)VF(j) * P + p], l = z[(size_t)VL(j) * P + p];
        s3 += f + l; s4 += l;
        if (cb) { s3 += cb[(size_t)(2 * j) * P + p]; s4 += cb[(size_t)(2 * j + 1) * P + p]; }
    }
    for (int o = 32; o > 0; o >>= 1) { s3 += __shfl_xor(s3, o, 64); s4 += __shfl_xor(s4, o, 64); }
    if (lane) return;
    const int r3 = RC3(D, b), r4 = RC4(D, b);
    if (D.shard != 0) {   // a further shard contributes its partitions' sums only (sign of mode 1); shard 0 carries the global variables and the affine parts
        rc[r3] = mode == 1 ? -s3 : s3; rc[r4] = mode == 1 ? -s4 : s4;
        return;
    }
    s3 += zg[b] + zg[D.B + b] - (D.has_n ? zg[2 * D.B + b] : 0.0);
    s4 += zg[D.B + b] - (D.has_m ? zg[3 * D.B + b] : 0.0);
    rc[r3] = mode == 1 ? D.bc[r3] - s3 : (mode == 2 ? s3 + addc[r3] : s3);
    rc[r4] = mode == 1 ? D.bc[r4] - s4 : (mode == 2 ? s4 + addc[r4] : s4);
}
// coupling rows NF[r], NL[r] and C6[r]; `cr`: extra terms [2 R][P].  Two stages (round 6): kRackChunks workgroups per row sum a fixed
// slice of the partitions each (one workgroup per row walked 100,000 partitions with 256 threads: 0.12 ms, five times an iteration), one
// thread per row adds the slices in order and the row's global variables.
constexpr int kRackChunks = 16;
__global__ void __launch_bounds__(kRedBlock) k_lp_A_rack_part(LpDev D, const double *z, const double *cr, double *part, int gated) {
    if (LP_STOPPED(D) || LP_GATED_OFF(D, gated)) return;
    __shared__ double sh[kRedBlock];
    const int R = D.R, P = D.P, row = blockIdx.x / kRackChunks, chunk = blockIdx.x % kRackChunks;     // row: 0..R-1 NF, R..2R-1 NL
    const int r = row % R, kind = row / R;
    const int per = (P + kRackChunks - 1) / kRackChunks, p0 = chunk * per, p1 = min(P, p0 + per);
    const double *zz = z + (size_t)(kind == 0 ? VYF(D, r) : VYL(D, r)) * P;
    const double *cc = cr ? cr + (size_t)(2 * r + kind) * P : nullptr;
    double s = 0.0;
    for (int p = p0 + threadIdx.x; p < p1; p += kRedBlock) s += zz[p] + (cc ? cc[p] : 0.0);
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = kRedBlock / 2; o > 0; o >>= 1) { if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) part[blockIdx.x] = sh[0];
}
__global__ void __launch_bounds__(64) k_lp_A_rack_fin(LpDev D, const double *part, const double *zg, double *rc, int mode, const double *addc, int gated) {
    if (LP_STOPPED(D) || LP_GATED_OFF(D, gated)) return;
    // one wavefront per row (0..R-1 NF, R..2R-1 NL, 2R..3R-1 C6): the lanes share the slices and the rack's brokers, fixed butterfly
    const int R = D.R, row = blockIdx.x, lane = threadIdx.x;
    const int r = row % R, kind = row / R;
    double s = 0.0;
    const bool own = D.shard == 0;
    if (kind < 2) {
        if (lane < kRackChunks) s = part[row * kRackChunks + lane];
        if (own) for (int e = D.rk_off[r] + lane; e < D.rk_off[r + 1]; e += 64) s -= zg[(kind == 0 ? 0 : D.B) + D.rk_mem[e]];
    } else if (D.has_n && own) {
        for (int e = D.rk_off[r] + lane; e < D.rk_off[r + 1]; e += 64) s += zg[2 * D.B + D.rk_mem[e]];
        if (D.has_k && lane == 0) s -= zg[4 * D.B + r];
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane) return;
    double out = own ? (mode == 1 ? D.bc[row] - s : (mode == 2 ? s + addc[row] : s)) : (mode == 1 ? -s : s);
    if (D.rowc[row] != 1) out = 0.0;      // absent / pinned rows carry no residual and no right-hand side
    rc[row] = out;
}

// ---- the normal equations' local eliminations (oracle/kao_lp_port.c::lp_solve_normal, first loop): local right-hand sides in
// place, the terms they send to the coupling rows into cb [2 NJ][P] (C3 / C4 of replica j) and cr [2 R][P] (NF / NL of rack r)
__global__ void k_lp_elim_local(LpDev D, const double *__restrict__ th, const double *__restrict__ fj, const double *__restrict__ fr, const double *__restrict__ ti, RowVec v,
                                double *__restrict__ cb, double *__restrict__ cr, int gated) {
    if (LP_STOPPED(D) || LP_GATED_OFF(D, gated)) return;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= D.P) return;
    const int P = D.P, R = D.R, NJ = D.NJ;
    const double *__restrict__ ID = fr + (size_t)3 * R * P, *__restrict__ F1 = fr + (size_t)4 * R * P, *__restrict__ F2 = fr + (size_t)5 * R * P, *__restrict__ E1 = fr + (size_t)1 * R * P, *__restrict__ E2 = fr + (size_t)2 * R * P;
    double r1 = v.r1[p], r2 = v.r2[p];
    if (D.has_c5)
        for (int j = 0; j < NJ; ++j) {
            const int b = cur_b(D, p, j);
            double c3 = 0.0, c4 = 0.0;
            if (b >= 0) {
                const double g5 = v.r5[(size_t)j * P + p] / fj[((size_t)3 * NJ + j) * P + p];
                const double k1 = fj[((size_t)4 * NJ + j) * P + p], k2 = fj[((size_t)5 * NJ + j) * P + p];
                r1 -= k1 * g5; v.r7[(size_t)D.rack[b] * P + p] -= k1 * g5; c3 = -k1 * g5;
                r2 -= k2 * g5; c4 = -k2 * g5;
            }
            cb[(size_t)(2 * j) * P + p] = c3; cb[(size_t)(2 * j + 1) * P + p] = c4;
        }
    // (round 6) With q_r = r7_r / d_r - (e1_r g1 + e2_r g2) / d_r the terms a partition sends to the coupling rows are
    //   NF[r]: -cyf (q_r + g1)    NL[r]: -cyl (q_r + g1 + g2)    C3_j: -sig11 (q_rk + g1) - sig12 g2    C4_j: -sig12 (q_rk + g1) - sig22 g2
    // (rk the rack of replica j's broker) -- the same sums as lp_col's columns give, regrouped: two passes over the racks with
    // independent loads, no division, no read-modify-write of cb / cr.
#pragma unroll 4
    for (int r = 0; r < R; ++r) {
        const double g7 = v.r7[(size_t)r * P + p] * ID[(size_t)r * P + p];
        r1 -= E1[(size_t)r * P + p] * g7; r2 -= E2[(size_t)r * P + p] * g7;
    }
    const double i11 = ti[(size_t)0 * P + p], i12 = ti[(size_t)1 * P + p], i22 = ti[(size_t)2 * P + p];
    const double g1 = i11 * r1 + i12 * r2, g2 = i12 * r1 + i22 * r2;
#pragma unroll 4
    for (int r = 0; r < R; ++r) {
        const size_t k = (size_t)r * P + p;
        const double q = v.r7[k] * ID[k] - (F1[k] * g1 + F2[k] * g2);
        cr[(size_t)(2 * r) * P + p] = -th[(size_t)VYF(D, r) * P + p] * (q + g1);
        cr[(size_t)(2 * r + 1) * P + p] = -th[(size_t)VYL(D, r) * P + p] * (q + g1 + g2);
    }
    for (int j = 0; j < NJ; ++j) {
        const int b = cur_b(D, p, j);
        double c3 = 0.0, c4 = 0.0;
        if (b >= 0) {
            const size_t k = (size_t)D.rack[b] * P + p;
            const double q = v.r7[k] * ID[k] - (F1[k] * g1 + F2[k] * g2);
            const double s11 = fj[((size_t)0 * NJ + j) * P + p], s12 = fj[((size_t)1 * NJ + j) * P + p], s22 = fj[((size_t)2 * NJ + j) * P + p];
            c3 = -s11 * (q + g1) - s12 * g2; c4 = -s12 * (q + g1) - s22 * g2;
            if (D.has_c5) { c3 += cb[(size_t)(2 * j) * P + p]; c4 += cb[(size_t)(2 * j + 1) * P + p]; }
        }
        cb[(size_t)(2 * j) * P + p] = c3; cb[(size_t)(2 * j + 1) * P + p] = c4;
    }
    v.r1[p] = r1; v.r2[p] = r2;
}
// back substitution (second loop): dy of the local rows in place, given dy of the coupling rows
__global__ void k_lp_back_local(LpDev D, const double *__restrict__ th, const double *__restrict__ fj, const double *__restrict__ fr, const double *__restrict__ ti, RowVec v, int gated) {
    if (LP_STOPPED(D) || LP_GATED_OFF(D, gated)) return;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= D.P) return;
    const int P = D.P, R = D.R, NJ = D.NJ;
    const double *__restrict__ ID = fr + (size_t)3 * R * P, *__restrict__ F1 = fr + (size_t)4 * R * P, *__restrict__ F2 = fr + (size_t)5 * R * P;
    // (round 6) E = sum of eps_c y_c over a rack's columns (NF, NL: cyf y_NF + cyl y_NL; a replica's C3 / C4: sig11 y3 + sig12 y4):
    //   t1 = r1 - sum E (1 - e1 / d),   t2 = r2 - sum (m2 . y) + sum E e2 / d,   dy7_r = (r7_r - E_r) / d_r - (e1_r d1 + e2_r d2) / d_r
    // -- lp_col's columns regrouped by rack: independent loads, no division.
    double t1 = v.r1[p], t2 = v.r2[p];
#pragma unroll 4
    for (int r = 0; r < R; ++r) {
        const size_t k = (size_t)r * P + p;
        const double yl = th[(size_t)VYL(D, r) * P + p] * v.rc[RNL(D, r)], E = th[(size_t)VYF(D, r) * P + p] * v.rc[RNF(D, r)] + yl;
        t1 -= E - F1[k] * E; t2 -= yl - F2[k] * E;
    }
    for (int j = 0; j < NJ; ++j) {
        const int b = cur_b(D, p, j);
        if (b < 0) continue;
        const size_t k = (size_t)D.rack[b] * P + p;
        const double s11 = fj[((size_t)0 * NJ + j) * P + p], s12 = fj[((size_t)1 * NJ + j) * P + p], s22 = fj[((size_t)2 * NJ + j) * P + p];
        const double y3 = v.rc[RC3(D, b)], y4 = v.rc[RC4(D, b)], E = s11 * y3 + s12 * y4;
        t1 -= E - F1[k] * E; t2 -= (s12 * y3 + s22 * y4) - F2[k] * E;
    }
    const double i11 = ti[(size_t)0 * P + p], i12 = ti[(size_t)1 * P + p], i22 = ti[(size_t)2 * P + p];
    const double d1 = i11 * t1 + i12 * t2, d2 = i12 * t1 + i22 * t2;
    v.r1[p] = d1; v.r2[p] = d2;
#pragma unroll 4
    for (int r = 0; r < R; ++r) {
        const size_t k = (size_t)r * P + p;
        const double E = th[(size_t)VYF(D, r) * P + p] * v.rc[RNF(D, r)] + th[(size_t)VYL(D, r) * P + p] * v.rc[RNL(D, r)];
        v.r7[k] = (v.r7[k] - E) * ID[k] - (F1[k] * d1 + F2[k] * d2);
    }
    for (int j = 0; j < NJ; ++j) {
        const int b = cur_b(D, p, j);
        if (b < 0) continue;
        const size_t k = (size_t)D.rack[b] * P + p;
        v.r7[k] -= (fj[((size_t)0 * NJ + j) * P + p] * v.rc[RC3(D, b)] + fj[((size_t)1 * NJ + j) * P + p] * v.rc[RC4(D, b)]) * ID[k];
    }
    for (int j = 0; j < NJ; ++j) {
        const int b = cur_b(D, p, j);
        if (!D.has_c5 || b < 0) { v.r5[(size_t)j * P + p] = 0.0; continue; }
        const double k1 = fj[((size_t)4 * NJ + j) * P + p], k2 = fj[((size_t)5 * NJ + j) * P + p];
        v.r5[(size_t)j * P + p] = (v.r5[(size_t)j * P + p] - k1 * (d1 + v.r7[(size_t)D.rack[b] * P + p] + v.rc[RC3(D, b)]) - k2 * (d2 + v.rc[RC4(D, b)]))
                                  / fj[((size_t)3 * NJ + j) * P + p];
    }
}

// ---- interior-point vector kernels ------------------------------------------------------------------------------------
// z = A^T y (only the starting point needs it as a vector)
__global__ void k_lp_AT(LpDev D, RowVec y, double *z, double *zg) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t nv = (size_t)D.NV * D.P;
    if (i < nv) z[i] = var_present(D, (int)(i / D.P), (int)(i % D.P)) ? at_val(D, (int)(i / D.P), (int)(i % D.P), y) : 0.0;
    else if (i < nv + D.GV) { const int g = (int)(i - nv); zg[g] = gvar_present(D, g) ? at_val_g(D, g, y.rc) : 0.0; }
}
// starting point: x = max(x~, xf) capped at half the upper bound, s = max(c - A^T y, 1), v = 1 where bounded
__global__ void k_lp_start(LpDev D, RowVec y, double *x, double *xg, double *s, double *sg, double *v, double *vg, double xf) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t nv = (size_t)D.NV * D.P;
    if (i < nv) {
        const int vv = (int)(i / D.P), p = (int)(i % D.P);
        if (!var_present(D, vv, p)) { x[i] = 1; s[i] = 1; v[i] = 0; return; }
        const double u = var_ub(D, vv);
        double xx = x[i] > xf ? x[i] : xf;
        if (u > 0) { const double cap = u * 0.5 > 1e-2 ? u * 0.5 : 1e-2; if (xx > cap) xx = cap; }
        x[i] = xx;
        const double ss = var_cost(D, vv, p) - at_val(D, vv, p, y);
        s[i] = ss > 1.0 ? ss : 1.0;
        v[i] = u > 0 ? 1.0 : 0.0;
    } else if (i < nv + D.GV) {
        const int g = (int)(i - nv);
        if (!gvar_present(D, g)) { xg[g] = 1; sg[g] = 1; vg[g] = 0; return; }
        const double u = gvar_ub(D, g);
        double xx = xg[g] > xf ? xg[g] : xf;
        if (u > 0) { const double cap = u * 0.5 > 1e-2 ? u * 0.5 : 1e-2; if (xx > cap) xx = cap; }
        xg[g] = xx;
        const double ss = gvar_cost(D, g) - at_val_g(D, g, y.rc);
        sg[g] = ss > 1.0 ? ss : 1.0;
        vg[g] = u > 0 ? 1.0 : 0.0;
    }
}
// the cost vector as a variable-space vector (A c for the starting point)
__global__ void k_lp_cost(LpDev D, double *z, double *zg) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t nv = (size_t)D.NV * D.P;
    if (i < nv) z[i] = var_present(D, (int)(i / D.P), (int)(i % D.P)) ? var_cost(D, (int)(i / D.P), (int)(i % D.P)) : 0.0;
    else if (i < nv + D.GV) zg[i - nv] = gvar_present(D, (int)(i - nv)) ? gvar_cost(D, (int)(i - nv)) : 0.0;
}
// dual residual rd = c - A^T y - s + v and the sums {|rd|^2, x.s + w.v, c.x, u.v}; one record per block
__global__ void __launch_bounds__(kRedBlock) k_lp_resid(LpDev D, VarVec x, VarVec s, VarVec v, RowVec y, VarVec rd, double *rec) {
    if (LP_STOPPED(D)) return;
    const size_t nv = (size_t)D.NV * D.P, stride = (size_t)gridDim.x * blockDim.x;
    double a[4] = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv + D.GV; i += stride) {
        if (i < nv) {
            const int vv = (int)(i / D.P), p = (int)(i % D.P);
            if (var_present(D, vv, p)) {
                const double u = var_ub(D, vv), c = var_cost(D, vv, p);
                const double r = c - at_val(D, vv, p, y) - s.z[i] + v.z[i];
                rd.z[i] = r; a[0] += r * r; a[1] += x.z[i] * s.z[i]; a[2] += c * x.z[i];
                if (u > 0) { a[1] += (u - x.z[i]) * v.z[i]; a[3] += u * v.z[i]; }
            }
        } else {
            const int g = (int)(i - nv);
            if (gvar_present(D, g)) {
                const double u = gvar_ub(D, g), c = gvar_cost(D, g);
                const double r = c - at_val_g(D, g, y.rc) - s.zg[g] + v.zg[g];
                rd.zg[g] = r;
                if (D.shard == 0) {
                    a[0] += r * r; a[1] += x.zg[g] * s.zg[g]; a[2] += c * x.zg[g];
                    if (u > 0) { a[1] += (u - x.zg[g]) * v.zg[g]; a[3] += u * v.zg[g]; }
                }
            }
        }
    }
    block_reduce(a, 4, false, rec + (size_t)blockIdx.x * kRedVals);
}
// sums over the rows: {|rp|^2 (local rows), b.y (local rows)}; one record per block (thread per partition)
__global__ void __launch_bounds__(kRedBlock) k_lp_rowsums(LpDev D, RowVec rp, RowVec y, double *rec) {
    if (LP_STOPPED(D)) return;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    double a[2] = {0, 0};
    if (p < D.P) {
        const int P = D.P;
        a[0] = rp.r1[p] * rp.r1[p] + rp.r2[p] * rp.r2[p];
        a[1] = D.RF * y.r1[p] + y.r2[p];
        for (int r = 0; r < D.R; ++r) { const double q = rp.r7[(size_t)r * P + p]; a[0] += q * q; a[1] += D.phi * y.r7[(size_t)r * P + p]; }
        for (int j = 0; j < D.NJ; ++j) {
            const double q = rp.r5[(size_t)j * P + p]; a[0] += q * q;
            if (D.has_c5 && cur_b(D, p, j) >= 0) a[1] += y.r5[(size_t)j * P + p];
        }
    }
    block_reduce(a, 2, false, rec + (size_t)blockIdx.x * kRedVals);
}
// the same over the coupling rows (one block)
__global__ void __launch_bounds__(kRedBlock) k_lp_rowsums_c(LpDev D, const double *rpc, const double *yc, double *out) {
    if (LP_STOPPED(D)) return;
    double a[2] = {0, 0};
    for (int i = threadIdx.x; i < D.mc; i += kRedBlock) { a[0] += rpc[i] * rpc[i]; if (D.rowc[i]) a[1] += D.bc[i] * yc[i]; }
    block_reduce(a, 2, false, out);
}
// h = rd - rxs / x + rwv / w and g = theta h; rxs, rwv are parked in ds, dv (k_lp_dir turns them into the directions).
// pass 1 (corrector): rxs = sigma mu - x s - dx_aff ds_aff, rwv = sigma mu - w v + dx_aff dv_aff
__global__ void k_lp_h(LpDev D, int pass, VarVec x, VarVec s, VarVec v, VarVec th, VarVec rd, VarVec dxa, VarVec dsa, VarVec dva,
                       VarVec h, VarVec g, VarVec ds, VarVec dv) {
    if (LP_STOPPED(D)) return;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t nv = (size_t)D.NV * D.P;
    const double sigma_mu = D.sc[SC_SIGMU];
    if (i < nv) {
        const int vv = (int)(i / D.P), p = (int)(i % D.P);
        if (!var_present(D, vv, p)) return;      // (absent variables' entries of every direction / right-hand-side vector are zero from lp_begin's memset on: nobody ever stores anything else there)
        const double u = var_ub(D, vv), w = u > 0 ? u - x.z[i] : 1.0;
        double rxs = -x.z[i] * s.z[i], rwv = u > 0 ? -w * v.z[i] : 0.0;
        if (pass) { rxs += sigma_mu - dxa.z[i] * dsa.z[i]; if (u > 0) rwv += sigma_mu + dxa.z[i] * dva.z[i]; }
        const double hh = rd.z[i] - rxs / x.z[i] + (u > 0 ? rwv / w : 0.0);
        h.z[i] = hh; g.z[i] = th.z[i] * hh; ds.z[i] = rxs; dv.z[i] = rwv;
    } else if (i < nv + D.GV) {
        const int k = (int)(i - nv);
        if (!gvar_present(D, k)) return;
        const double u = gvar_ub(D, k), w = u > 0 ? u - x.zg[k] : 1.0;
        double rxs = -x.zg[k] * s.zg[k], rwv = u > 0 ? -w * v.zg[k] : 0.0;
        if (pass) { rxs += sigma_mu - dxa.zg[k] * dsa.zg[k]; if (u > 0) rwv += sigma_mu + dxa.zg[k] * dva.zg[k]; }
        const double hh = rd.zg[k] - rxs / x.zg[k] + (u > 0 ? rwv / w : 0.0);
        h.zg[k] = hh; g.zg[k] = th.zg[k] * hh; ds.zg[k] = rxs; dv.zg[k] = rwv;
    }
}
// dx = theta (A^T dy - h), ds = (rxs - s dx) / x, dv = (rwv + v dx) / w; step lengths to the boundary {alpha_p, alpha_d} (min)
__global__ void __launch_bounds__(kRedBlock) k_lp_dir(LpDev D, VarVec x, VarVec s, VarVec v, VarVec th, VarVec h, RowVec dy, VarVec dx, VarVec ds, VarVec dv, double *rec) {
    if (LP_STOPPED(D)) return;
    const size_t nv = (size_t)D.NV * D.P, stride = (size_t)gridDim.x * blockDim.x;
    double a[2] = {1.0, 1.0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv + D.GV; i += stride) {
        double xi = 0, si = 0, vi = 0, thi = 0, hi = 0, atv = 0, u = 0, rxs = 0, rwv = 0;
        bool on = false;
        if (i < nv) {
            const int vv = (int)(i / D.P), p = (int)(i % D.P);
            on = var_present(D, vv, p);
            if (on) { xi = x.z[i]; si = s.z[i]; vi = v.z[i]; thi = th.z[i]; hi = h.z[i]; atv = at_val(D, vv, p, dy); u = var_ub(D, vv); rxs = ds.z[i]; rwv = dv.z[i]; }
        } else {
            const int k = (int)(i - nv);
            on = gvar_present(D, k);
            if (on) { xi = x.zg[k]; si = s.zg[k]; vi = v.zg[k]; thi = th.zg[k]; hi = h.zg[k]; atv = at_val_g(D, k, dy.rc); u = gvar_ub(D, k); rxs = ds.zg[k]; rwv = dv.zg[k]; }
        }
        double ddx = 0, dds = 0, ddv = 0;
        if (on) {
            ddx = thi * (atv - hi);
            dds = (rxs - si * ddx) / xi;
            if (ddx < 0) a[0] = fmin(a[0], -xi / ddx);
            if (dds < 0) a[1] = fmin(a[1], -si / dds);
            if (u > 0) {
                const double w = u - xi;
                ddv = (rwv + vi * ddx) / w;
                if (ddx > 0) a[0] = fmin(a[0], w / ddx);
                if (ddv < 0) a[1] = fmin(a[1], -vi / ddv);
            }
        }
        if (on) { dx.z[i] = ddx; ds.z[i] = dds; dv.z[i] = ddv; }       // (the global variables sit right behind the partition ones; absent entries stay zero)
    }
    block_reduce(a, 2, true, rec + (size_t)blockIdx.x * kRedVals);
}
// corrector right-hand side: h, g = theta h and the parked targets (rxs -> dsc, rwv -> dvc) from the trial point of the direction (dx, ds, dv)
__global__ void k_lp_mcc_h(LpDev D, VarVec x, VarVec s, VarVec v, VarVec th, VarVec dx, VarVec ds, VarVec dv, VarVec h, VarVec g, VarVec dsc, VarVec dvc) {
    if (LP_STOPPED(D) || D.sc[SC_MCC_GO] == 0.0) return;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t nv = (size_t)D.NV * D.P;
    if (i >= nv + D.GV) return;
    const double ap = D.sc[SC_AP], ad = D.sc[SC_AD], mut = D.sc[SC_SIGMU];
    const double apt = ap + kMccDelta < 1.0 ? ap + kMccDelta : 1.0, adt = ad + kMccDelta < 1.0 ? ad + kMccDelta : 1.0;
    bool on; double u;
    if (i < nv) { const int vv = (int)(i / D.P), p = (int)(i % D.P); on = var_present(D, vv, p); u = var_ub(D, vv); }
    else { const int k = (int)(i - nv); on = gvar_present(D, k); u = gvar_ub(D, k); }
    if (!on) return;   // (the global variables sit right behind the partition ones; absent entries stay zero)
    const double xi = x.z[i], si = s.z[i], vi = v.z[i];
    double pr = (xi + apt * dx.z[i]) * (si + adt * ds.z[i]);
    double tg = pr < kMccBmin * mut ? kMccBmin * mut : (pr > kMccBmax * mut ? kMccBmax * mut : pr);
    double rxs = tg - pr; if (rxs < -kMccBmax * mut) rxs = -kMccBmax * mut;
    double rwv = 0.0, w = 1.0;
    if (u > 0) {
        w = u - xi;
        pr = (w - apt * dx.z[i]) * (vi + adt * dv.z[i]);
        tg = pr < kMccBmin * mut ? kMccBmin * mut : (pr > kMccBmax * mut ? kMccBmax * mut : pr);
        rwv = tg - pr; if (rwv < -kMccBmax * mut) rwv = -kMccBmax * mut;
    }
    const double hh = -rxs / xi + (u > 0 ? rwv / w : 0.0);
    h.z[i] = hh; g.z[i] = th.z[i] * hh; dsc.z[i] = rxs; dvc.z[i] = rwv;
}
// correction direction (dxc, dsc, dvc) from dy and the parked targets; step lengths of direction + correction {alpha_p, alpha_d} (min)
__global__ void __launch_bounds__(kRedBlock) k_lp_mcc_dir(LpDev D, VarVec x, VarVec s, VarVec v, VarVec th, VarVec h, RowVec dy, VarVec dxc, VarVec dsc, VarVec dvc,
                                                            VarVec dx, VarVec ds, VarVec dv, double *rec) {
    if (LP_STOPPED(D) || D.sc[SC_MCC_GO] == 0.0) return;
    const size_t nv = (size_t)D.NV * D.P, stride = (size_t)gridDim.x * blockDim.x;
    double a[2] = {1.0, 1.0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv + D.GV; i += stride) {
        bool on = false; double u = 0, atv = 0;
        if (i < nv) { const int vv = (int)(i / D.P), p = (int)(i % D.P); on = var_present(D, vv, p); if (on) { u = var_ub(D, vv); atv = at_val(D, vv, p, dy); } }
        else { const int k = (int)(i - nv); on = gvar_present(D, k); if (on) { u = gvar_ub(D, k); atv = at_val_g(D, k, dy.rc); } }
        double c1 = 0, c2 = 0, c3 = 0;
        if (on) {
            const double xi = x.z[i], si = s.z[i], vi = v.z[i];
            c1 = th.z[i] * (atv - h.z[i]);
            c2 = (dsc.z[i] - si * c1) / xi;
            const double tx = dx.z[i] + c1, ts = ds.z[i] + c2;
            if (tx < 0) a[0] = fmin(a[0], -xi / tx);
            if (ts < 0) a[1] = fmin(a[1], -si / ts);
            if (u > 0) {
                const double w = u - xi;
                c3 = (dvc.z[i] + vi * c1) / w;
                const double tv = dv.z[i] + c3;
                if (tx > 0) a[0] = fmin(a[0], w / tx);
                if (tv < 0) a[1] = fmin(a[1], -vi / tv);
            }
        }
        if (on) { dxc.z[i] = c1; dsc.z[i] = c2; dvc.z[i] = c3; }
    }
    block_reduce(a, 2, true, rec + (size_t)blockIdx.x * kRedVals);
}
// an accepted corrector joins the direction: variables, then rows
__global__ void k_lp_mcc_acc(LpDev D, VarVec dx, VarVec ds, VarVec dv, VarVec dxc, VarVec dsc, VarVec dvc) {
    if (LP_STOPPED(D) || D.sc[SC_MCC_ACC] == 0.0) return;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)D.NV * D.P + D.GV) return;
    dx.z[i] += dxc.z[i]; ds.z[i] += dsc.z[i]; dv.z[i] += dvc.z[i];
}
__global__ void k_lp_mcc_acc_rows(const double *sc, const double *d, double *y, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (sc[SC_STOP] != 0.0 || sc[SC_MCC_ACC] == 0.0) return;
    if (i < n) y[i] += d[i];
}
// sum (x + ap dx)(s + ad ds) + (w - ap dx)(v + ad dv)
__global__ void __launch_bounds__(kRedBlock) k_lp_muaff(LpDev D, VarVec x, VarVec s, VarVec v, VarVec dx, VarVec ds, VarVec dv, double *rec) {
    if (LP_STOPPED(D)) return;
    const size_t nv = (size_t)D.NV * D.P, stride = (size_t)gridDim.x * blockDim.x;
    double a[1] = {0};
    const double ap = D.sc[SC_AP], ad = D.sc[SC_AD];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv + D.GV; i += stride) {
        bool on; double u;
        if (i < nv) { const int vv = (int)(i / D.P), p = (int)(i % D.P); on = var_present(D, vv, p); u = var_ub(D, vv); }
        else { const int k = (int)(i - nv); on = gvar_present(D, k) && D.shard == 0; u = gvar_ub(D, k); }
        if (!on) continue;
        a[0] += (x.z[i] + ap * dx.z[i]) * (s.z[i] + ad * ds.z[i]);
        if (u > 0) a[0] += (u - x.z[i] - ap * dx.z[i]) * (v.z[i] + ad * dv.z[i]);
    }
    block_reduce(a, 1, false, rec + (size_t)blockIdx.x * kRedVals);
}
__global__ void k_lp_update(LpDev D, VarVec x, VarVec s, VarVec v, VarVec dx, VarVec ds, VarVec dv) {
    if (LP_STOPPED(D)) return;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t nv = (size_t)D.NV * D.P;
    const double ap = D.sc[SC_AP], ad = D.sc[SC_AD];
    if (i < nv) {
        if (!var_present(D, (int)(i / D.P), (int)(i % D.P))) return;
        x.z[i] += ap * dx.z[i]; s.z[i] += ad * ds.z[i];
        if (var_ub(D, (int)(i / D.P)) > 0) v.z[i] += ad * dv.z[i];
    } else if (i < nv + D.GV) {
        const int k = (int)(i - nv);
        if (!gvar_present(D, k)) return;
        x.zg[k] += ap * dx.zg[k]; s.zg[k] += ad * ds.zg[k];
        if (gvar_ub(D, k) > 0) v.zg[k] += ad * dv.zg[k];
    }
}
// y += ad dy over all rows (local rows stored contiguously: r1 r2 r7 r5, then rc)
__global__ void k_lp_axpy(const double *sc, const double *d, double *y, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (sc[SC_STOP] != 0.0) return;
    if (i < n) y[i] += sc[SC_AD] * d[i];
}
// row duals -> K-bound multipliers in its fixed point: g[r] = -y_C6[r], a[b] = -y_C3[b] - g[rack b], l[b] = -y_C4[b]
// The primal iterate in centi-units for the rounding on the host (lp_round_assignment; specification oracle/kao_lp.py round_primal):
// q[k * P + p] = min(250, rint(100 x)), k = j (f_j), NJ + j (l_j), 2 NJ + r (yf_r), 2 NJ + R + r (yl_r); absent variables 0;
// zq[b] = rint(zf_b), zq[B + b] = rint(zl_b)
__global__ void k_lp_round(LpDev D, const double *x, const double *xg, uint8_t *q, int32_t *zq) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int K = 2 * D.NJ + 2 * D.R;
    if (i < (size_t)K * D.P) {
        const int k = (int)(i / D.P), p = (int)(i % D.P);
        const int v = k < D.NJ ? VF(k) : (k < 2 * D.NJ ? VL(k - D.NJ) : (k < 2 * D.NJ + D.R ? VYF(D, k - 2 * D.NJ) : VYL(D, k - 2 * D.NJ - D.R)));
        double c = 0.0;
        if (var_present(D, v, p)) c = fmin(250.0, fmax(0.0, rint(100.0 * x[(size_t)v * D.P + p])));
        q[i] = (uint8_t)(c == c ? c : 255.0);   // a non-finite iterate: 255 (never an integer: the partition is fractional)
    } else if (i < (size_t)K * D.P + 2 * (size_t)D.B) {
        const int g = (int)(i - (size_t)K * D.P);
        const double z = rint(xg[g]);
        zq[g] = (z == z && fabs(z) < 1e9) ? (int32_t)z : 0;
    }
}
__global__ void k_lp_multipliers(LpDev D, const double *yc, int32_t *a, int32_t *l, int32_t *g) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    auto fx = [](double v) { v = rint(v * kDualScale); v = fmin(fmax(v, -(double)kDualClamp), (double)kDualClamp); return (int32_t)v; };
    if (i < D.B) {
        const double gr = D.rowc[RC6(D, D.rack[i])] == 1 ? -yc[RC6(D, D.rack[i])] : 0.0;
        a[i] = fx(-yc[RC3(D, i)] - gr);
        l[i] = fx(-yc[RC4(D, i)]);
    }
    if (i < D.R) g[i] = fx(D.rowc[RC6(D, i)] == 1 ? -yc[RC6(D, i)] : 0.0);
}

}  // namespace


// ------------------------------------------------------------------------------------------------------------------------
// Host driver.  An iteration is a fixed sequence of kernel launches on the context's stream with no host round trip: the
// scalars live in LpDev::sc.  The host enqueues iterations in batches and looks at the stop flag in between.
struct LpCtx {
    LpDev D{};
    int device = 0;
    hipStream_t st = nullptr;
    std::vector<void *> bufs;
    long nvar = 0, nub = 0;
    double nb = 1, ncn = 1;
    VarVec x{}, s{}, v{}, th{}, rd{}, h{}, g{}, d1{}, d2{}, dsa{}, dva{}, ds{}, dv{};
    RowVec y{}, rp{}, w1{}, w2{};
    size_t rows_local = 0;     // doubles of the local rows of one RowVec (r1 r2 r7 r5 contiguous)
    double *fj = nullptr, *fr = nullptr, *ti = nullptr, *S = nullptr, *Linv = nullptr, *diag0 = nullptr, *cb = nullptr, *cr = nullptr;
    double *rec = nullptr, *redA = nullptr, *redB = nullptr, *redC = nullptr, *part = nullptr, *ylast = nullptr, *trace = nullptr;
    int32_t *d_mult = nullptr, *d_zq = nullptr;
    VarVec dc{}; RowVec wc{}; int mcc = 2;   // centrality correctors per iteration (KAO_LP_MCC; 0: none) and their direction / row vector
    bool rack_mfma = true;         // the rack x rack block of the Schur complement on the matrix cores (2R <= 128; KAO_LP_RACK=old: the LDS-tiled kernel)
    double *qd = nullptr, *wr = nullptr; int *qc = nullptr; int ncp = 0;   // the partitions' coupling columns / replica rows for k_lp_schur_broker (k_lp_factor_local); ncp = columns per partition, padded (0: more than 128, not kept)
    double *rack_part = nullptr;   // [2 R][kRackChunks] slice sums of the rack rows
    int broker_u = 4;          // incidences in flight per wavefront in k_lp_schur_broker (KAO_LP_BROKER_U: 4 / 8 / 16)
    double *xz = nullptr;      // exchange vectors of the triangular solves (kao_chol.hip)
    LpFan *fan = nullptr;      // shard 0 of a fan (kao_internal.h): the lp_* entry points forward to it
    double *tri = nullptr;     // a shard's packed lower triangle of S for the all-reduce
    LpColl *coll = nullptr; int rank = 0;   // a shard of one LP over several devices (kao_internal.h LpShard); null: the whole topic
    int coll_rc = KAO_OK;      // first failure of a collective (checked by lp_enqueue / lp_begin)
    void all_sum(double *buf, size_t n) { if (coll && !coll_rc) coll_rc = coll->allreduce(rank, buf, n, false, st); }
    void all_min(double *buf, size_t n) { if (coll && !coll_rc) coll_rc = coll->allreduce(rank, buf, n, true, st); }
    uint8_t *d_q = nullptr;   // quantised primal iterate (lp_primal)
    int nblk_var = 0, nblk_p = 0, rack_chunk = 0, rack_tile = 0, rack_blocks = 0, broker_waves = 0;
    int maxit = 80, trace_cap = 0;
    double *h_sc = nullptr;    // pinned mirror of the scalars: slot 0 for lp_poll / lp_begin, slots 1..kLpRing for the marks of lp_enqueue_mark
    hipEvent_t ev[32] = {};
    hipGraphExec_t graph = nullptr;   // one iteration, captured once: a launch instead of ~250 (every kernel argument is fixed for the context's lifetime)
    bool graph_tried = false;
    double t_begin = 0;
    int enqueued = 0;          // iterations enqueued since lp_begin
    bool used = false;         // lp_begin has run on this context before

    // Device memory comes in a few large chunks, carved in order (256-byte aligned): a context of a 100,000-partition topic is ~70 buffers and
    // 3 GB -- one hipMalloc / hipFree each cost 6 + 7.5 ms of a 0.42-s solve, the frees between the last iteration and the rounding.
    char *chunk = nullptr; size_t chunk_size = 0, chunk_used = 0;
    template <class T> int alloc(T **p, size_t n) {
        const size_t bytes = (std::max<size_t>(n, 1) * sizeof(T) + 255) & ~(size_t)255;
        if (!chunk || chunk_used + bytes > chunk_size) {
            const size_t want = std::min<size_t>(1ull << 30, std::max<size_t>(16ull << 20, 128 * ((size_t)D.NV * D.P + D.GV)));
            const size_t sz = std::max(bytes, want);
            void *q = nullptr;
            if (hipMalloc(&q, sz) != hipSuccess) { (void)hipGetLastError(); if (sz == bytes || hipMalloc(&q, bytes) != hipSuccess) return fail(KAO_ERR_NOMEM, "KAO-LP: hipMalloc failed"); chunk_size = bytes; }
            else chunk_size = sz;
            bufs.push_back(q);
            chunk = static_cast<char *>(q); chunk_used = 0;
        }
        *p = reinterpret_cast<T *>(chunk + chunk_used);
        chunk_used += bytes;
        return KAO_OK;
    }
    template <class T> int upload(T **p, const std::vector<T> &hv) {
        int rc = alloc(p, hv.size());
        if (rc) return rc;
        if (!hv.empty()) HIP_TRY(hipMemcpy(*p, hv.data(), hv.size() * sizeof(T), hipMemcpyHostToDevice));
        return KAO_OK;
    }
    int var_vec(VarVec &vv) {
        int rc = alloc(&vv.z, (size_t)D.NV * D.P + D.GV);
        vv.zg = vv.z + (size_t)D.NV * D.P;      // contiguous: kernels index the global variables right behind the partition ones
        return rc;
    }
    int row_vec(RowVec &r) {
        int rc = alloc(&r.r1, rows_local + D.mcp);
        r.r2 = r.r1 + D.P; r.r7 = r.r2 + D.P; r.r5 = r.r7 + (size_t)D.P * D.R; r.rc = r.r1 + rows_local;
        return rc;
    }
    ~LpCtx() {
        for (void *p : bufs) (void)hipFree(p);
        if (h_sc) (void)hipHostFree(h_sc);
        for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e);
        if (graph) (void)hipGraphExecDestroy(graph);
        if (st) (void)hipStreamDestroy(st);
    }
};

namespace {

void lp_reduce(LpCtx &c, int nrec, int n, bool is_min, double *out) {
    hipLaunchKernelGGL(k_lp_red_final, dim3(1), dim3(kRedBlock), 0, c.st, c.D.sc, c.rec, nrec, n, is_min ? 1 : 0, out);
    if (is_min) c.all_min(out, (size_t)n); else c.all_sum(out, (size_t)n);
}
// local rows of A z into `out` (mode 0 plain, 1 = b - A z, 2 = A z + add)
void lp_rows_local(LpCtx &c, const VarVec &z, const RowVec &out, int mode, const RowVec &add, int gated = 0) {
    hipLaunchKernelGGL(k_lp_A_local, dim3(c.nblk_p), dim3(256), 0, c.st, c.D, z.z, out, mode, add, gated);
}
// coupling rows, gathered: rows of A z (+ the elimination terms cb / cr) (+ add)
void lp_rows_coupling(LpCtx &c, const VarVec &z, double *out_rc, int mode, const double *add_rc, const double *cb, const double *cr, int gated = 0) {
    const LpDev &D = c.D;
    hipLaunchKernelGGL(k_lp_A_broker, dim3((D.B + 3) / 4), dim3(256), 0, c.st, D, z.z, z.zg, cb, out_rc, mode, add_rc, gated);
    hipLaunchKernelGGL(k_lp_A_rack_part, dim3(2 * D.R * kRackChunks), dim3(kRedBlock), 0, c.st, D, z.z, cr, c.rack_part, gated);
    hipLaunchKernelGGL(k_lp_A_rack_fin, dim3(3 * D.R), dim3(64), 0, c.st, D, c.rack_part, z.zg, out_rc, mode, add_rc, gated);
    c.all_sum(out_rc, (size_t)D.mc);
}

void lp_factor(LpCtx &c) {
    const LpDev &D = c.D;
    hipLaunchKernelGGL(k_lp_factor_local, dim3(c.nblk_p), dim3(256), 0, c.st, D, c.th.z, c.fj, c.fr, c.ti, c.qd, c.qc, c.wr, c.ncp);
    const size_t lds_b = (size_t)c.broker_waves * 2 * D.mc * sizeof(double);
    const dim3 bg((D.B + c.broker_waves - 1) / c.broker_waves), bb(64 * c.broker_waves);
    if (2 * D.NJ + 2 * D.R > 64) hipLaunchKernelGGL((k_lp_schur_broker<4, 2>), bg, bb, lds_b, c.st, D, c.th.z, c.th.zg, c.fj, c.fr, c.ti, c.qd, c.qc, c.wr, c.S);
    else if (c.broker_u >= 16) hipLaunchKernelGGL(k_lp_schur_broker<16>, bg, bb, lds_b, c.st, D, c.th.z, c.th.zg, c.fj, c.fr, c.ti, c.qd, c.qc, c.wr, c.S);
    else if (c.broker_u >= 8) hipLaunchKernelGGL(k_lp_schur_broker<8>, bg, bb, lds_b, c.st, D, c.th.z, c.th.zg, c.fj, c.fr, c.ti, c.qd, c.qc, c.wr, c.S);
    else hipLaunchKernelGGL(k_lp_schur_broker<4>, bg, bb, lds_b, c.st, D, c.th.z, c.th.zg, c.fj, c.fr, c.ti, c.qd, c.qc, c.wr, c.S);
    const int n2 = 2 * D.R, per = 6 * n2 + D.R, t16 = (n2 + 15) / 16;
    const dim3 sg((n2 * n2 + 255) / 256), sb(256), sg2((n2 * n2 + 15) / 16);
#define KAO_RACK_MFMA(T) do { hipLaunchKernelGGL(k_lp_schur_rack_mfma<T>, dim3(kRackMfmaBlocks, rack_mfma_shares(T)), dim3(256), 0, c.st, D, c.th.z, c.fr, c.ti, c.part); \
                              hipLaunchKernelGGL(k_lp_schur_rack_sum2<T>, sg2, sb, 0, c.st, D, c.part, kRackMfmaBlocks, c.th.zg, c.S); } while (0)
    if (c.rack_mfma && t16 == 1) KAO_RACK_MFMA(1);
    else if (c.rack_mfma && t16 == 2) KAO_RACK_MFMA(2);
    else if (c.rack_mfma && t16 == 3) KAO_RACK_MFMA(3);
    else if (c.rack_mfma && t16 == 4) KAO_RACK_MFMA(4);
    else if (c.rack_mfma && t16 == 5) KAO_RACK_MFMA(5);
    else if (c.rack_mfma && t16 == 6) KAO_RACK_MFMA(6);
    else if (c.rack_mfma && t16 == 7) KAO_RACK_MFMA(7);
    else if (c.rack_mfma && t16 == 8) KAO_RACK_MFMA(8);
    else {
        hipLaunchKernelGGL(k_lp_schur_rack, dim3(c.rack_blocks), dim3(256), (size_t)c.rack_tile * per * sizeof(double), c.st, D, c.th.z, c.fj, c.fr, c.ti, c.rack_chunk, c.rack_tile, c.part);
        hipLaunchKernelGGL(k_lp_schur_rack_sum, sg, sb, 0, c.st, D, c.part, c.rack_blocks, c.th.zg, c.S);
    }
#undef KAO_RACK_MFMA
    if (c.coll) {   // shards: every shard gathered its own partitions' part of the lower triangle; the parts meet packed
        const size_t ntri = (size_t)D.mc * (D.mc + 1) / 2;
        hipLaunchKernelGGL(k_lp_tri_pack, dim3(1024), dim3(256), 0, c.st, D, c.S, c.tri, 0, c.S);
        c.all_sum(c.tri, ntri);
        hipLaunchKernelGGL(k_lp_tri_pack, dim3(1024), dim3(256), 0, c.st, D, c.S, c.tri, 1, c.S);
    }
    hipLaunchKernelGGL(k_lp_schur_fix, dim3(D.mcp), dim3(64), 0, c.st, D, c.th.zg, c.S, c.diag0);
    hipLaunchKernelGGL(k_lp_schur_fix_cols, dim3((D.mc + 255) / 256), dim3(256), 0, c.st, D, c.S);
    chol_enqueue(c.st, D.sc + SC_STOP, c.S, D.mcp, c.diag0, c.Linv);      // kao_chol.hip
}

// N dy = rho, in place in `v`: v's local rows hold rho; the coupling right-hand side is GATHERED: rows of A z + the elimination terms (+ add)
void lp_solve_normal(LpCtx &c, const RowVec &v, const VarVec &z, const double *add_rc, int gated = 0) {
    const LpDev &D = c.D;
    hipLaunchKernelGGL(k_lp_elim_local, dim3(c.nblk_p), dim3(256), 0, c.st, D, c.th.z, c.fj, c.fr, c.ti, v, c.cb, c.cr, gated);
    lp_rows_coupling(c, z, v.rc, add_rc ? 2 : 0, add_rc, c.cb, c.cr, gated);
    trsv_enqueue(c.st, D.sc + SC_STOP, gated ? D.sc + SC_MCC_GO : nullptr, c.S, D.mcp, v.rc, c.Linv, c.xz);      // kao_chol.hip
    hipLaunchKernelGGL(k_lp_back_local, dim3(c.nblk_p), dim3(256), 0, c.st, D, c.th.z, c.fj, c.fr, c.ti, v, gated);
}

// residuals, sums, trace, stopping test of the current iterate
void lp_enqueue_resid(LpCtx &c) {
    const LpDev &D = c.D;
    lp_rows_local(c, c.x, c.rp, 1, c.rp);
    lp_rows_coupling(c, c.x, c.rp.rc, 1, nullptr, nullptr, nullptr);
    hipLaunchKernelGGL(k_lp_resid, dim3(c.nblk_var), dim3(kRedBlock), 0, c.st, D, c.x, c.s, c.v, c.y, c.rd, c.rec);
    lp_reduce(c, c.nblk_var, 4, false, c.redA);
    hipLaunchKernelGGL(k_lp_rowsums, dim3(c.nblk_p), dim3(kRedBlock), 0, c.st, D, c.rp, c.y, c.rec);
    lp_reduce(c, c.nblk_p, 2, false, c.redB);
    hipLaunchKernelGGL(k_lp_rowsums_c, dim3(1), dim3(kRedBlock), 0, c.st, D, c.rp.rc, c.y.rc, c.redC);
    hipLaunchKernelGGL(k_lp_sc_resid, dim3(1), dim3(1), 0, c.st, D.sc, c.redA, c.redB, c.redC, c.trace);
    hipLaunchKernelGGL(k_lp_keep_last, dim3((D.mcp + 255) / 256), dim3(256), 0, c.st, D.sc, c.y.rc, c.ylast, D.mcp);
}

}  // namespace

// Builds the device image of one topic's compact LP.
int lp_open(const kao_topic *t, LpCtx **out, const LpShard *shard) {
    int rc = require_init();
    if (rc) return rc;
    rc = validate(t);
    if (rc) return rc;
    int32_t bd[8];
    derive_bounds(t, bd);
    const int Pg = t->n_partitions, p0 = shard ? shard->p0 : 0, P = shard ? shard->p1 - shard->p0 : Pg;   // this context's partitions: p0 .. p0 + P - 1
    {   // the bands' implied ends (round 6; oracle/kao_lp.py lp_bands): when the brokers' lower ends add up to all the replicas nobody can be above
        // its lower end -- the band is a point and its slack column is left out (likewise from above; likewise leaders and racks).  Same feasible
        // set.  With config 5's "cap + 1" on a cluster whose average is whole (300 .. 301 at 1000 x 100,000) every feasible point pinned the slack
        // at zero: an LP without interior, 200 iterations without converging, a certificate 2,481 above the optimum the rounding had found.
        const long long tot = (long long)Pg * t->rf;
        const long long nb = t->n_brokers, nr = t->n_racks;
        if (nb * bd[0] == tot) bd[1] = bd[0]; else if (nb * bd[1] == tot) bd[0] = bd[1];
        if (nb * bd[2] == Pg) bd[3] = bd[2]; else if (nb * bd[3] == Pg) bd[2] = bd[3];
        if (nr * bd[4] == tot) bd[5] = bd[4]; else if (nr * bd[5] == tot) bd[4] = bd[5];
    }
    if (shard && (p0 < 0 || P < 1 || shard->p1 > Pg || !shard->coll)) return fail(KAO_ERR_INVALID, "KAO-LP: bad shard");
    const int B = t->n_brokers, R = t->n_racks, NJ = t->rf_cur;
    const int mc = 3 * R + 2 * B;
    if ((size_t)2 * mc * sizeof(double) > 150 * 1024) return fail(KAO_ERR_UNSUPPORTED, "KAO-LP: more than ~4,700 brokers (a Schur row pair must fit LDS)");
    if (NJ > 8 || NJ < 1 || P >= (1 << 28)) return fail(KAO_ERR_UNSUPPORTED, "KAO-LP: current RF outside 1..8");
    LpCtx *c = new LpCtx();
    c->device = cur_device();
    LpDev &D = c->D;
    D.Pg = Pg; D.p0 = p0; D.shard = shard ? shard->rank : 0;
    c->coll = shard ? shard->coll : nullptr; c->rank = shard ? shard->rank : 0;
    D.P = P; D.B = B; D.R = R; D.NJ = NJ; D.RF = t->rf; D.NV = 3 * NJ + 3 * R; D.GV = 4 * B + R; D.mc = mc; D.mcp = (mc + kNB - 1) / kNB * kNB;
    D.phi = bd[7];
    D.has_c5 = bd[7] >= 2; D.has_t = bd[7] > bd[6]; D.t_ub = (D.has_t && bd[6] > 0) ? bd[7] - bd[6] : 0;
    D.has_n = bd[1] > bd[0]; D.n_ub = bd[1] - bd[0];
    D.has_m = bd[3] > bd[2]; D.m_ub = bd[3] - bd[2];
    D.has_k = D.has_n && bd[5] > bd[4]; D.k_ub = bd[5] - bd[4];
    auto bail = [&](int code) { delete c; return code; };
    {   // highest priority: an iteration is a chain of ~250 small dependent kernels; behind a full K-search grid at normal priority
        // every one of them queues (measured on the drifted 1000 x 100,000 topic: 4 iterations in 280 ms beside K-search, 38 ms alone)
        int lo = 0, hi = 0;
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess || hipStreamCreateWithPriority(&c->st, hipStreamNonBlocking, hi) != hipSuccess)
            return bail(fail(KAO_ERR_HIP, "KAO-LP: stream"));
    }
    if (hipHostMalloc(reinterpret_cast<void **>(&c->h_sc), sizeof(double) * kScN * 33) != hipSuccess) return bail(fail(KAO_ERR_NOMEM, "KAO-LP: pinned buffer"));
    for (hipEvent_t &e : c->ev) if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return bail(fail(KAO_ERR_HIP, "KAO-LP: event"));
    // structure
    std::vector<uint16_t> cur(t->current + (size_t)p0 * NJ, t->current + (size_t)(p0 + P) * NJ);
    std::vector<uint8_t> rack(t->rack_of, t->rack_of + B);
    std::vector<int> inc_off((size_t)B + 1, 0), rk_off((size_t)R + 1, 0), rsz((size_t)R, 0);
    for (int p = 0; p < P; ++p)
        for (int j = 0; j < NJ; ++j) { const unsigned b = cur[(size_t)p * NJ + j]; if (b != KAO_NONE && (int)b < B) inc_off[b + 1]++; }
    for (int b = 0; b < B; ++b) { inc_off[(size_t)b + 1] += inc_off[(size_t)b]; rsz[rack[(size_t)b]]++; }
    std::vector<int> inc((size_t)inc_off[(size_t)B]), fill(inc_off.begin(), inc_off.end() - 1);
    for (int p = 0; p < P; ++p)
        for (int j = 0; j < NJ; ++j) { const unsigned b = cur[(size_t)p * NJ + j]; if (b != KAO_NONE && (int)b < B) inc[(size_t)fill[b]++] = (p << 3) | j; }
    for (int r = 0; r < R; ++r) rk_off[(size_t)r + 1] = rk_off[(size_t)r] + rsz[(size_t)r];
    std::vector<int> rk_mem((size_t)B), rfill(rk_off.begin(), rk_off.end() - 1);
    for (int b = 0; b < B; ++b) rk_mem[(size_t)rfill[rack[(size_t)b]]++] = b;
    std::vector<double> cost((size_t)2 * NJ * P, 0.0), cg((size_t)D.GV, 0.0), bc((size_t)mc, 0.0);
    std::vector<unsigned char> rowc((size_t)mc, 0);
    double ncn = 0, nbn = 0;
    long nvar = 0, nub = 0;
    for (int p = 0; p < P; ++p)
        for (int j = 0; j < NJ; ++j) {
            const unsigned b = cur[(size_t)p * NJ + j];
            if (b == KAO_NONE || (int)b >= B) continue;
            const int cr = j == 0 ? 0 : 1, bw = t->broker_w ? t->broker_w[b] : 0, bwl = t->broker_wl ? t->broker_wl[b] : 0;
            const double cf = -(double)(t->w[cr][1] + bw), cl = -(double)(t->w[cr][0] + bw + bwl);
            cost[(size_t)(2 * j) * P + p] = cf; cost[(size_t)(2 * j + 1) * P + p] = cl;
        }
    for (int p = 0; p < Pg; ++p)      // the normalisers of the stopping test count the WHOLE topic (a shard stops when the topic's solve does)
        for (int j = 0; j < NJ; ++j) {
            const unsigned b = t->current[(size_t)p * NJ + j];
            if (b == KAO_NONE || (int)b >= B) continue;
            const int cr = j == 0 ? 0 : 1, bw = t->broker_w ? t->broker_w[b] : 0, bwl = t->broker_wl ? t->broker_wl[b] : 0;
            const double cf = -(double)(t->w[cr][1] + bw), cl = -(double)(t->w[cr][0] + bw + bwl);
            ncn += cf * cf + cl * cl;
            nvar += 2 + (D.has_c5 ? 1 : 0);
            if (D.has_c5) nbn += 1.0;
        }
    nvar += (long)Pg * R * (2 + (D.has_t ? 1 : 0));
    if (D.t_ub) nub += (long)Pg * R;
    for (int b = 0; b < B; ++b) {
        const int bw = t->broker_w ? t->broker_w[b] : 0, bwl = t->broker_wl ? t->broker_wl[b] : 0;
        cg[(size_t)b] = -(double)bw; cg[(size_t)B + b] = -(double)(bw + bwl);
        ncn += cg[(size_t)b] * cg[(size_t)b] + cg[(size_t)B + b] * cg[(size_t)B + b];
        nvar += 2 + (D.has_n ? 1 : 0) + (D.has_m ? 1 : 0);
        nub += (D.has_n ? 1 : 0) + (D.has_m ? 1 : 0);
        rowc[(size_t)(3 * R + 2 * b)] = 1; bc[(size_t)(3 * R + 2 * b)] = bd[0];
        rowc[(size_t)(3 * R + 2 * b + 1)] = 1; bc[(size_t)(3 * R + 2 * b + 1)] = bd[2];
    }
    for (int r = 0; r < R; ++r) {
        if (D.has_k) { nvar++; nub++; }
        rowc[(size_t)r] = 1; rowc[(size_t)(R + r)] = 1;
        if (D.has_n) { rowc[(size_t)(2 * R + r)] = 1; bc[(size_t)(2 * R + r)] = (double)bd[4] - (double)rsz[(size_t)r] * bd[0]; }
    }
    if (!D.has_n) rowc[0] = 2;                 // exact row dependencies (oracle/kao_lp_port.c): NF[0] / NL[0] pinned
    if (!D.has_m) rowc[(size_t)R] = 2;
    nbn += (double)Pg * t->rf * t->rf + (double)Pg + (double)Pg * R * bd[7] * bd[7];
    for (double q : bc) nbn += q * q;
    c->nvar = nvar; c->nub = nub; c->nb = 1.0 + std::sqrt(nbn); c->ncn = 1.0 + std::sqrt(ncn);
    uint16_t *d_cur; uint8_t *d_rack; int *d_io, *d_inc, *d_ro, *d_rm; double *d_c, *d_cg, *d_bc; unsigned char *d_rowc;
    if ((rc = c->upload(&d_cur, cur)) || (rc = c->upload(&d_rack, rack)) || (rc = c->upload(&d_io, inc_off)) || (rc = c->upload(&d_inc, inc)) ||
        (rc = c->upload(&d_ro, rk_off)) || (rc = c->upload(&d_rm, rk_mem)) || (rc = c->upload(&d_c, cost)) || (rc = c->upload(&d_cg, cg)) ||
        (rc = c->upload(&d_bc, bc)) || (rc = c->upload(&d_rowc, rowc)))
        return bail(rc);
    D.cur = d_cur; D.rack = d_rack; D.inc_off = d_io; D.inc = d_inc; D.rk_off = d_ro; D.rk_mem = d_rm; D.c = d_c; D.cg = d_cg; D.bc = d_bc; D.rowc = d_rowc;
    c->rows_local = (size_t)P * (2 + R + NJ);
    for (VarVec *vv : {&c->x, &c->s, &c->v, &c->th, &c->rd, &c->h, &c->g, &c->d1, &c->d2, &c->dsa, &c->dva, &c->ds, &c->dv})
        if ((rc = c->var_vec(*vv))) return bail(rc);
    for (RowVec *rv : {&c->y, &c->rp, &c->w1, &c->w2, &c->wc})
        if ((rc = c->row_vec(*rv))) return bail(rc);
    if ((rc = c->var_vec(c->dc))) return bail(rc);
    { const char *e = std::getenv("KAO_LP_MCC"); c->mcc = e ? std::max(0, std::min(4, std::atoi(e))) : 2; }
    const size_t nvtot = (size_t)D.NV * P + D.GV;
    c->nblk_var = (int)std::min<size_t>((nvtot + kRedBlock - 1) / kRedBlock, (size_t)kVarBlocks);
    c->nblk_p = (P + 255) / 256;
    c->broker_waves = std::max(1, std::min(4, (int)((150 * 1024) / ((size_t)2 * mc * sizeof(double)))));
    const int n2 = 2 * R, per = 6 * n2 + R;
    c->rack_tile = std::max(1, std::min(16, (int)((64 * 1024) / ((size_t)per * sizeof(double)))));
    c->rack_chunk = std::max(c->rack_tile, ((P + 255) / 256 + c->rack_tile - 1) / c->rack_tile * c->rack_tile);   // about 256 blocks
    c->rack_blocks = (P + c->rack_chunk - 1) / c->rack_chunk;
    c->trace_cap = 512;
    c->ncp = 2 * NJ + 2 * R <= 128 ? (2 * NJ + 2 * R + 7) & ~7 : 0;   // (k_lp_schur_broker: one or two columns per lane)
    if ((rc = c->alloc(&c->fj, (size_t)6 * NJ * P)) || (rc = c->alloc(&c->fr, (size_t)6 * R * P)) || (rc = c->alloc(&c->ti, (size_t)3 * P)) ||
        (rc = c->alloc(&c->S, (size_t)D.mcp * D.mcp)) || (rc = c->alloc(&c->Linv, (size_t)D.mcp * kNB)) || (rc = c->alloc(&c->diag0, (size_t)D.mcp)) || (rc = c->alloc(&c->cb, (size_t)2 * NJ * P)) ||
        (rc = c->alloc(&c->cr, (size_t)2 * R * P)) || (rc = c->alloc(&c->rec, (size_t)std::max(c->nblk_var, c->nblk_p) * kRedVals)) ||
        (rc = c->alloc(&c->redA, (size_t)kRedVals)) || (rc = c->alloc(&c->redB, (size_t)kRedVals)) || (rc = c->alloc(&c->redC, (size_t)kRedVals)) ||
        (rc = c->alloc(&c->part, std::max((size_t)c->rack_blocks * n2 * n2, (size_t)kRackMfmaBlocks * rack_mfma_record(std::min(kRackMfmaT16Max, (n2 + 15) / 16))))) || (rc = c->alloc(&c->ylast, (size_t)D.mcp)) ||
        (rc = c->alloc(&c->xz, (size_t)2 * D.mcp)) || (rc = c->alloc(&c->tri, shard ? (size_t)mc * (mc + 1) / 2 : 1)) || (rc = c->alloc(&c->qd, (size_t)3 * c->ncp * P)) || (rc = c->alloc(&c->qc, (size_t)c->ncp * P)) || (rc = c->alloc(&c->wr, (size_t)4 * NJ * P)) || (rc = c->alloc(&c->rack_part, (size_t)2 * R * kRackChunks)) ||
        (rc = c->alloc(&c->d_mult, (size_t)2 * B + R)) || (rc = c->alloc(&D.sc, (size_t)kScN)) || (rc = c->alloc(&c->trace, (size_t)5 * c->trace_cap)))
        return bail(rc);
    // dynamic LDS beyond 64 KiB has to be enabled per kernel
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_lp_schur_broker<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_lp_schur_broker<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_lp_schur_broker<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_lp_schur_broker<4, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    { const char *e = std::getenv("KAO_LP_RACK"); c->rack_mfma = !(e && e[0] == 'o'); }
    { const char *e = std::getenv("KAO_LP_BROKER_U"); c->broker_u = e ? std::atoi(e) : 4; }   // (round 6, measured at 100,000 partitions: 4.10 / 4.29 / 4.41 ms an iteration with 4 / 8 / 16)
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_lp_schur_rack), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    for (const VarVec *vv : {&c->x, &c->s, &c->v, &c->th, &c->rd, &c->h, &c->g, &c->d1, &c->d2, &c->dsa, &c->dva, &c->ds, &c->dv, &c->dc})
        HIP_TRY(hipMemsetAsync(vv->z, 0, nvtot * sizeof(double), c->st));
    for (const RowVec *rv : {&c->y, &c->rp, &c->w1, &c->w2, &c->wc}) HIP_TRY(hipMemsetAsync(rv->r1, 0, (c->rows_local + D.mcp) * sizeof(double), c->st));
    HIP_TRY(hipMemsetAsync(c->S, 0, (size_t)D.mcp * D.mcp * sizeof(double), c->st));
    HIP_TRY(hipMemsetAsync(c->ylast, 0, (size_t)D.mcp * sizeof(double), c->st));
    HIP_TRY(hipStreamSynchronize(c->st));
    *out = c;
    return KAO_OK;
}

// Raises the stop flag from the host: whatever is still enqueued turns into no-ops (a context that is closed with work in flight
// would otherwise wait for all of it).
thread_local bool t_lp_inner = false;
void lp_set_fan(LpCtx *c, LpFan *fan) { c->fan = fan; }
void lp_abort(LpCtx *c) {
    if (!c) return;
    if (c->fan && !t_lp_inner) { c->fan->abort(); return; }
    (void)hipSetDevice(c->device);
    const double four = 4.0;   // its own value: an aborted solve's iterate is mid-way, nobody may take it for a finished one
    (void)hipMemcpy(c->D.sc + SC_STOP, &four, sizeof four, hipMemcpyHostToDevice);
}
void lp_close(LpCtx *c) {
    if (c && c->fan && !t_lp_inner) { c->fan->close(); return; }
    delete c;
}

// Enqueues the starting point (theta = 1: x~ = A^T (A A^T)^-1 b, y = (A A^T)^-1 A c, s = c - A^T y, pushed into the interior) and the
// residuals of iterate 0.  Asynchronous on the context's stream.
int lp_begin(LpCtx *cp, double tol, int maxit, double pert, uint32_t salt) {
    if (cp->fan && !t_lp_inner) return cp->fan->begin(tol, maxit, pert, salt);
    LpCtx &c = *cp;
    const LpDev &D = c.D;
    HIP_TRY(hipSetDevice(c.device));
    c.t_begin = now_s();
    c.maxit = std::min(maxit, c.trace_cap - 2);
    c.enqueued = 0;
    if (c.used) {   // a second solve on the context (kao_solve: the perturbed LP after the certificate's): everything lp_open zeroed
        const size_t nv0 = (size_t)D.NV * D.P + D.GV;
        for (const VarVec *vv : {&c.x, &c.s, &c.v, &c.th, &c.rd, &c.h, &c.g, &c.d1, &c.d2, &c.dsa, &c.dva, &c.ds, &c.dv, &c.dc})
            HIP_TRY(hipMemsetAsync(vv->z, 0, nv0 * sizeof(double), c.st));
        for (const RowVec *rv : {&c.y, &c.rp, &c.w1, &c.w2, &c.wc}) HIP_TRY(hipMemsetAsync(rv->r1, 0, (c.rows_local + D.mcp) * sizeof(double), c.st));
        HIP_TRY(hipMemsetAsync(c.S, 0, (size_t)D.mcp * D.mcp * sizeof(double), c.st));
        HIP_TRY(hipMemsetAsync(c.ylast, 0, (size_t)D.mcp * sizeof(double), c.st));
    }
    c.used = true;
    double init[kScN];
    std::memset(init, 0, sizeof init);
    init[SC_GAMMA] = lp_gamma(); init[SC_SIGEXP] = lp_sigexp((long long)c.D.Pg * c.D.RF); init[SC_TOL] = tol; init[SC_MAXIT] = c.maxit; init[SC_NVU] = (double)(c.nvar + c.nub); init[SC_NB] = c.nb; init[SC_NCN] = c.ncn;
    init[SC_PERT] = pert > 0.0 ? pert : 0.0; init[SC_SALT] = (double)salt;
    init[SC_MU_REF] = 1e300; init[SC_IT_REF] = 0.0; init[SC_PINF_BEST] = 1e300;
    std::memcpy(c.h_sc, init, sizeof init);
    HIP_TRY(hipMemcpyAsync(D.sc, c.h_sc, sizeof init, hipMemcpyHostToDevice, c.st));
    const size_t nvtot = (size_t)D.NV * D.P + D.GV;
    const dim3 gv((unsigned)((nvtot + 255) / 256)), b256(256);
    hipLaunchKernelGGL(k_lp_theta, dim3((unsigned)(((size_t)D.NV * D.P + 255) / 256)), b256, 0, c.st, D, c.x.z, c.s.z, c.v.z, c.th.z, 1);
    hipLaunchKernelGGL(k_lp_theta_g, dim3((D.GV + 255) / 256), b256, 0, c.st, D, c.x.zg, c.s.zg, c.v.zg, c.th.zg, 1);
    lp_factor(c);
    HIP_TRY(hipMemsetAsync(c.g.z, 0, nvtot * sizeof(double), c.st));
    lp_rows_local(c, c.g, c.w1, 1, c.w1);                               // w1 = b - A 0 = b
    lp_rows_coupling(c, c.g, c.w1.rc, 1, nullptr, nullptr, nullptr);
    lp_solve_normal(c, c.w1, c.g, c.w1.rc);                             // coupling rhs = b_c + the elimination terms (gathered in place)
    hipLaunchKernelGGL(k_lp_AT, gv, b256, 0, c.st, D, c.w1, c.x.z, c.x.zg);
    hipLaunchKernelGGL(k_lp_cost, gv, b256, 0, c.st, D, c.g.z, c.g.zg);
    lp_rows_local(c, c.g, c.y, 0, c.y);                                 // y = A c
    lp_solve_normal(c, c.y, c.g, nullptr);
    hipLaunchKernelGGL(k_lp_start, gv, b256, 0, c.st, D, c.y, c.x.z, c.x.zg, c.s.z, c.s.zg, c.v.z, c.v.zg, lp_xfloor());
    lp_enqueue_resid(c);
    HIP_TRY(hipGetLastError());
    return c.coll_rc;
}

// Enqueues `k` iterations (each: factor, predictor, corrector, update, residuals + stopping test of the new iterate).  Asynchronous;
// iterations behind the one that sets the stop flag are no-ops.
static void lp_enqueue_one(LpCtx &c) {
    const LpDev &D = c.D;
    const size_t nvtot = (size_t)D.NV * D.P + D.GV;
    const dim3 gv((unsigned)((nvtot + 255) / 256)), b256(256);
    {
        hipLaunchKernelGGL(k_lp_theta, dim3((unsigned)(((size_t)D.NV * D.P + 255) / 256)), b256, 0, c.st, D, c.x.z, c.s.z, c.v.z, c.th.z, 0);
        hipLaunchKernelGGL(k_lp_theta_g, dim3((D.GV + 255) / 256), b256, 0, c.st, D, c.x.zg, c.s.zg, c.v.zg, c.th.zg, 0);
        lp_factor(c);
        for (int pass = 0; pass < 2; ++pass) {
            const VarVec &dx = pass ? c.d2 : c.d1, &pds = pass ? c.ds : c.dsa, &pdv = pass ? c.dv : c.dva;
            const RowVec &dy = pass ? c.w2 : c.w1;
            hipLaunchKernelGGL(k_lp_h, gv, b256, 0, c.st, D, pass, c.x, c.s, c.v, c.th, c.rd, c.d1, c.dsa, c.dva, c.h, c.g, pds, pdv);
            lp_rows_local(c, c.g, dy, 2, c.rp);                          // local rows: A (theta h) + rp
            lp_solve_normal(c, dy, c.g, c.rp.rc);                        // coupling rows gathered with the elimination terms + rp
            hipLaunchKernelGGL(k_lp_dir, dim3(c.nblk_var), dim3(kRedBlock), 0, c.st, D, c.x, c.s, c.v, c.th, c.h, dy, dx, pds, pdv, c.rec);
            lp_reduce(c, c.nblk_var, 2, true, c.redA);
            hipLaunchKernelGGL(k_lp_sc_step, dim3(1), dim3(1), 0, c.st, D.sc, c.redA, pass);
            if (pass == 0) {
                hipLaunchKernelGGL(k_lp_muaff, dim3(c.nblk_var), dim3(kRedBlock), 0, c.st, D, c.x, c.s, c.v, c.d1, c.dsa, c.dva, c.rec);
                lp_reduce(c, c.nblk_var, 1, false, c.redA);
                hipLaunchKernelGGL(k_lp_sc_sigma, dim3(1), dim3(1), 0, c.st, D.sc, c.redA);
            }
        }
        const size_t nrow = c.rows_local + D.mcp;
        for (int k = 0; k < c.mcc; ++k) {   // centrality correctors (no-ops once one was rejected or the step is full)
            hipLaunchKernelGGL(k_lp_mcc_h, gv, b256, 0, c.st, D, c.x, c.s, c.v, c.th, c.d2, c.ds, c.dv, c.h, c.g, c.dsa, c.dva);
            lp_rows_local(c, c.g, c.wc, 0, c.wc, 1);                     // local rows: A (theta h)
            lp_solve_normal(c, c.wc, c.g, nullptr, 1);
            hipLaunchKernelGGL(k_lp_mcc_dir, dim3(c.nblk_var), dim3(kRedBlock), 0, c.st, D, c.x, c.s, c.v, c.th, c.h, c.wc, c.dc, c.dsa, c.dva, c.d2, c.ds, c.dv, c.rec);
            lp_reduce(c, c.nblk_var, 2, true, c.redA);
            hipLaunchKernelGGL(k_lp_sc_mcc, dim3(1), dim3(1), 0, c.st, D.sc, c.redA);
            hipLaunchKernelGGL(k_lp_mcc_acc, gv, b256, 0, c.st, D, c.d2, c.ds, c.dv, c.dc, c.dsa, c.dva);
            hipLaunchKernelGGL(k_lp_mcc_acc_rows, dim3((unsigned)((nrow + 255) / 256)), b256, 0, c.st, D.sc, c.wc.r1, c.w2.r1, nrow);
        }
        hipLaunchKernelGGL(k_lp_sc_final, dim3(1), dim3(1), 0, c.st, D.sc);
        hipLaunchKernelGGL(k_lp_update, gv, b256, 0, c.st, D, c.x, c.s, c.v, c.d2, c.ds, c.dv);
        hipLaunchKernelGGL(k_lp_axpy, dim3((unsigned)((nrow + 255) / 256)), b256, 0, c.st, D.sc, c.w2.r1, c.y.r1, nrow);
        lp_enqueue_resid(c);
    }
}
int lp_enqueue(LpCtx *cp, int k) {
    LpCtx &c = *cp;
    HIP_TRY(hipSetDevice(c.device));
    if (!c.graph_tried) {   // capture one iteration (KAO_LP_GRAPH=0: plain launches)
        c.graph_tried = true;
        const char *e = std::getenv("KAO_LP_GRAPH");
        if (!(e && e[0] == '0') && !c.coll && hipStreamBeginCapture(c.st, hipStreamCaptureModeThreadLocal) == hipSuccess) {   // (a shard's iteration holds collectives: plain launches)
            lp_enqueue_one(c);
            hipGraph_t g = nullptr;
            if (hipStreamEndCapture(c.st, &g) == hipSuccess && g) {
                if (hipGraphInstantiate(&c.graph, g, nullptr, nullptr, 0) != hipSuccess) c.graph = nullptr;
                (void)hipGraphDestroy(g);
            }
            (void)hipGetLastError();
        }
    }
    for (int q = 0; q < k; ++q) {
        if (c.graph) HIP_TRY(hipGraphLaunch(c.graph, c.st));
        else lp_enqueue_one(c);
    }
    c.enqueued += k;
    HIP_TRY(hipGetLastError());
    return c.coll_rc;
}

// lp_enqueue + a MARK: behind the k iterations the scalars are copied to ring slot `slot` (0..31) and an event is recorded.  lp_poll_mark
// waits for exactly that mark -- not for whatever was enqueued after it -- so a caller that keeps a few marks in flight (kao_solve: one per
// K-search launch, read three launches later) hardly ever blocks and still sees the state of a fixed iteration count: deterministic.
int lp_enqueue_mark(LpCtx *cp, int k, int slot) {
    if (cp->fan && !t_lp_inner) return cp->fan->enqueue_mark(k, slot);
    int rc = lp_enqueue(cp, k);
    if (rc) return rc;
    LpCtx &c = *cp;
    HIP_TRY(hipMemcpyAsync(c.h_sc + (size_t)(1 + (slot & 31)) * kScN, c.D.sc, sizeof(double) * kScN, hipMemcpyDeviceToHost, c.st));
    HIP_TRY(hipEventRecord(c.ev[slot & 31], c.st));
    return KAO_OK;
}
int lp_poll_mark(LpCtx *cp, int slot, int *status, int *iterations, double deadline) {
    if (cp->fan && !t_lp_inner) return cp->fan->poll_mark(slot, status, iterations, deadline);
    LpCtx &c = *cp;
    HIP_TRY(hipSetDevice(c.device));
    if (deadline > 0) {   // a bounded wait: past the deadline the stop flag goes up (what is still enqueued turns into no-ops) and the mark arrives at once
        bool aborted = false;
        while (hipEventQuery(c.ev[slot & 31]) == hipErrorNotReady) {
            if (!aborted && now_s() >= deadline) { lp_abort(cp); aborted = true; }
            std::this_thread::sleep_for(std::chrono::microseconds(100));
        }
        (void)hipGetLastError();
    }
    HIP_TRY(hipEventSynchronize(c.ev[slot & 31]));
    const double *h = c.h_sc + (size_t)(1 + (slot & 31)) * kScN;
    if (status) *status = (int)h[SC_STOP];
    if (iterations) *iterations = (int)h[SC_IT];
    return KAO_OK;
}

// Waits for what has been enqueued and reads the scalars: *status = the stop flag (0 = still running), *iterations so far.
int lp_poll(LpCtx *cp, int *status, int *iterations) {
    LpCtx &c = *cp;
    HIP_TRY(hipSetDevice(c.device));
    HIP_TRY(hipMemcpyAsync(c.h_sc, c.D.sc, sizeof(double) * kScN, hipMemcpyDeviceToHost, c.st));
    HIP_TRY(hipStreamSynchronize(c.st));
    if (status) *status = (int)c.h_sc[SC_STOP];
    if (iterations) *iterations = (int)c.h_sc[SC_IT];
    return KAO_OK;
}

// The multipliers of the last finite iterate (host, may be null: a[B] l[B] g[R] in K-bound's fixed point); stats[8] = {iterations,
// README objective of the primal iterate, of the dual iterate, status (0 converged, 1 iteration limit, 3 stalled: the last finite
// iterate is returned), mu, relative primal infeasibility, relative dual infeasibility, milliseconds since lp_begin}; trace (may be
// null): 5 doubles per iterate, iterations + 1 of them.
int lp_finish(LpCtx *cp, int32_t *multipliers, double stats[8], double *trace) {
    if (cp->fan && !t_lp_inner) return cp->fan->finish(multipliers, stats, trace);
    LpCtx &c = *cp;
    const LpDev &D = c.D;
    int st = 0, it = 0;
    int rc = lp_poll(cp, &st, &it);
    if (rc) return rc;
    if (c.h_sc[SC_HAVE_LAST] == 0.0) return fail(KAO_ERR_HIP, "KAO-LP: the starting point is not finite");
    hipLaunchKernelGGL(k_lp_multipliers, dim3((std::max(D.B, D.R) + 255) / 256), dim3(256), 0, c.st, D, c.ylast, c.d_mult, c.d_mult + D.B, c.d_mult + 2 * D.B);
    HIP_TRY(hipGetLastError());
    if (multipliers) HIP_TRY(hipMemcpyAsync(multipliers, c.d_mult, ((size_t)2 * D.B + D.R) * 4, hipMemcpyDeviceToHost, c.st));
    if (trace) HIP_TRY(hipMemcpyAsync(trace, c.trace, sizeof(double) * 5 * (size_t)(it + 1), hipMemcpyDeviceToHost, c.st));
    HIP_TRY(hipStreamSynchronize(c.st));
    if (stats) {
        stats[0] = it; stats[1] = -c.h_sc[SC_PLAST]; stats[2] = -c.h_sc[SC_DLAST]; stats[3] = st == 1 ? 0 : (st == 2 || st == 0 || st == 4 ? 1 : 3);
        stats[4] = c.h_sc[SC_MU]; stats[5] = c.h_sc[SC_PINF]; stats[6] = c.h_sc[SC_DINF]; stats[7] = (now_s() - c.t_begin) * 1e3;
    }
    return KAO_OK;
}

// The primal iterate, quantised (k_lp_round): q[(2 NJ + 2 R) * P] bytes, zq[2 B] ints, both host memory.  After lp_finish / once the
// stop flag is up (the iterate does not move behind it).
int lp_primal(LpCtx *cp, uint8_t *q, int32_t *zq) {
    if (cp->fan && !t_lp_inner) return cp->fan->primal(q, zq);
    LpCtx &c = *cp;
    const LpDev &D = c.D;
    HIP_TRY(hipSetDevice(c.device));
    const size_t nq = (size_t)(2 * D.NJ + 2 * D.R) * D.P;
    if (!c.d_q) { int rc = c.alloc(&c.d_q, nq); if (rc) return rc; rc = c.alloc(&c.d_zq, (size_t)2 * D.B); if (rc) return rc; }
    hipLaunchKernelGGL(k_lp_round, dim3((unsigned)((nq + 2 * (size_t)D.B + 255) / 256)), dim3(256), 0, c.st, D, c.x.z, c.x.zg, c.d_q, c.d_zq);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(q, c.d_q, nq, hipMemcpyDeviceToHost, c.st));
    HIP_TRY(hipMemcpyAsync(zq, c.d_zq, (size_t)2 * D.B * 4, hipMemcpyDeviceToHost, c.st));
    HIP_TRY(hipStreamSynchronize(c.st));
    return KAO_OK;
}

// One shot: iterations in batches of four until the stop flag is up.
int lp_solve(LpCtx *c, double tol, int maxit, int32_t *multipliers, double stats[8], double *trace, double pert, uint32_t salt) {
    int rc = lp_begin(c, tol, maxit, pert, salt);
    if (rc) return rc;
    for (int st = 0, it = 0; !st;) {
        if ((rc = lp_enqueue(c, 4)) || (rc = lp_poll(c, &st, &it))) return rc;
    }
    return lp_finish(c, multipliers, stats, trace);
}

}  // namespace kao
