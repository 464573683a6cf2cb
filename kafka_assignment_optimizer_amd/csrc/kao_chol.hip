// kao_chol.hip -- KAO-LP's dense piece: Cholesky of the coupling rows' Schur complement and the two triangular solves (round 6).
//
// The Schur complement of the 3R + 2B coupling rows (kao_lp.hip) is a dense symmetric positive definite matrix of ~2,000 rows at the
// north-star size; an interior-point iteration factors it once and solves with it two to four times.  Round 5 did that with 3 launches
// per 64-row tile column (a one-wavefront diagonal factor of 50 us, scalar 4 x 4 micro-tile panel solve and trailing update): 2.8 ms of
// a 7.3-ms iteration, most of it the 64-step dependent chain of the diagonal tile.  Here:
//
//   * ONE launch per tile column k (k_chol_step): workgroup (i, j) of the trailing matrix recomputes the two panel tiles it needs,
//     L_ik = A_ik Linv_kk^T and L_jk (a triangular product: only the k-steps below the diagonal of Linv are issued), subtracts
//     L_ik L_jk^T from its tile, and the workgroup of tile (k+1, k+1) goes on to factor and invert it -- so the next launch finds
//     Linv_{k+1,k+1} ready.  No workgroup ever waits for another one: the only synchronisation is the kernel boundary.
//   * every 64 x 64 x 64 product runs on v_mfma_f64_16x16x4_f64 (operands from LDS, leading dimension 65; accumulators in the C layout
//     col = lane & 15, row = (lane >> 4) + 4 reg), eight wavefronts per workgroup, two 16 x 16 accumulators each.
//   * the diagonal tile is factored in four panels of 16 columns: one wavefront holds the panel with lane = row and the 16 columns in
//     registers, pivots travel by v_readlane, 1 / sqrt(pivot) is v_rsq_f64 + two Newton steps (no division, no sqrt in the chain);
//     the trailing 16 x 16 blocks and the blocks of the tile's inverse are MFMA products; the 16 x 16 diagonal inverses are computed
//     by an otherwise idle wavefront while the next panel is factored.
//   * the factor is written transposed into the upper triangle as the tiles are finished (nothing else lives there) and mirrored into
//     the lower triangle by one launch at the end: both triangular solves then read rows of S with consecutive lanes on consecutive
//     columns, as in round 5.
//   * the triangular solves (k_trsv) keep round 5's shape -- one workgroup per row tile, x_j handed from workgroup to workgroup -- but
//     the hand-off is the datum itself: the exchange vector is preset to a NaN pattern no computation produces, a producer stores its
//     64 values with agent-scope relaxed stores, a consumer polls the value it needs.  No flag, no fence: a hop costs one store -> load.
//
// Pivots that lost all but 1e-12 of their entry pin a dependent row (L_jj = 1e64), as before.  Every sum has a fixed order: the same
// bits on every run.  f64 throughout.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "kao_host.h"

namespace kao {
namespace {

constexpr int NB = 64;                 // tile
constexpr int LD = 65;                 // leading dimension of a tile in LDS (odd: lanes on consecutive rows hit distinct banks)
constexpr int kStepThreads = 512;      // eight wavefronts
constexpr double kPivotRel = 1e-12, kPivotBig = 1e64, kPivotBigInv = 1e-64;
constexpr int kWbLd = 17;
constexpr bool kPanelBlocked = true;   // the diagonal tile's panels in register-blocked halves (false: round 6's column-by-column panel, kept for A/B)
// LDS carve of k_chol_step / k_chol_first: two tiles, three 16 x 17 scratch blocks, the tile's original diagonal, the pivots' 1 / sqrt
constexpr size_t kStepLds = sizeof(double) * (2 * (size_t)NB * LD + 3 * 16 * kWbLd + 2 * NB + 2 * NB);   // (+ two column buffers of the panel factor)

typedef double v4d __attribute__((ext_vector_type(4)));

__device__ __forceinline__ v4d mfma(double a, double b, v4d c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
__device__ __forceinline__ double readlane_d(double v, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
// 1 / sqrt(p), p > 0: the hardware estimate and two Newton steps y <- y + y (1/2 - (p y) (y / 2))
__device__ __forceinline__ double rsqrt_nr(double p) {
    double y = __builtin_amdgcn_rsq(p);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const double e = fma(-(p * y), 0.5 * y, 0.5);
        y = fma(y, e, y);
    }
    return y;
}

// 1 / sqrt(p), p > 0: the hardware estimate (2^-23 relative) and ONE third-order step: with e = 1 - p y^2, 1 / sqrt(p) = y (1 + e / 2 + 3 e^2 / 8 + O(e^3))
// -- five operations, four deep, against eight and six for the two Newton steps (the blocked panel's chain is this function)
__device__ __forceinline__ double rsqrt_h3(double p) {
    const double y = __builtin_amdgcn_rsq(p);
    const double e = fma(-(p * y), y, 1.0);
    return fma(y * e, fma(0.375, e, 0.5), y);
}

// 1 / p, p > 0: the hardware estimate and two Newton steps y <- y + y (1 - p y)
__device__ __forceinline__ double rcp_nr(double p) {
    double y = __builtin_amdgcn_rcp(p);
#pragma unroll
    for (int k = 0; k < 2; ++k) y = fma(y, fma(-p, y, 1.0), y);
    return y;
}

// ---- the diagonal tile: T (LDS, lower triangle valid) -> L in place, V = L^-1 (LDS, zeros above the diagonal) ----------------------
// 16 x 16 diagonal block at c0 of the factored tile -> its inverse into V (lanes 0..15 of one wavefront: lane = column)
__device__ __forceinline__ void dinv_block(const double *T, double *V, const double *rinv_s, int c0, int lane) {
    if (lane >= 16) return;
    // column `lane` of the inverse, right-looking: once x_k is final every later row's sum takes L_ik x_k -- the dependent chain is one
    // multiply and one fused multiply-add per row instead of a dot product per row
    double s[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) s[i] = i == lane ? -1.0 : 0.0;     // x_i = -s_i / L_ii: the unit right-hand side enters as s_lane = -1
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const double xk = k < lane ? 0.0 : -s[k] * rinv_s[c0 + k];
        s[k] = xk;
#pragma unroll
        for (int i = k + 1; i < 16; ++i) s[i] = fma(T[(c0 + i) * LD + c0 + k], xk, s[i]);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) V[(c0 + i) * LD + c0 + lane] = s[i];
}

// All kStepThreads threads of the workgroup call this.  d0s[64]: the tile's diagonal before the factorisation (pivot test).
__device__ __forceinline__ void potrf_inv(double *T, double *V, double *Wb, const double *d0s, double *rinv_s, double *colb, int tid, long long *dbg = nullptr) {
    const int w = tid >> 6, lane = tid & 63, lm = lane & 15, lq = lane >> 4;
    int dn = 0;
#define KAO_TICK() do { if (dbg && tid == 0) dbg[dn] = (long long)__builtin_amdgcn_s_memtime(); ++dn; } while (0)
    KAO_TICK();
    for (int e = tid; e < 6 * 256; e += kStepThreads) {      // the six 16 x 16 blocks above the diagonal (the rest of V is written below)
        const int blk = e >> 8, bi = blk < 3 ? 0 : (blk < 5 ? 1 : 2), bj = blk < 3 ? blk + 1 : (blk < 5 ? blk - 1 : 3);
        V[(16 * bi + ((e >> 4) & 15)) * LD + 16 * bj + (e & 15)] = 0.0;
    }
    for (int p = 0; p < 4; ++p) {
        const int c0 = 16 * p;
        if (w == 0 && kPanelBlocked) {
            // The panel in two halves of eight columns (round 6, last).  The chain of the column-by-column panel below -- pivot, 1 / pivot, the
            // column through LDS to the other lanes, their multiply-add, the next pivot by v_readlane -- cost ~430 cycles a column whatever its
            // instruction count.  Here EVERY lane factors the half's 8 x 8 diagonal block for itself, in registers (36 entries, read from LDS as
            // broadcasts): the chain is 1 / sqrt(pivot) -> column entry -> next pivot with no cross-lane step in it, and the redundant work (120
            // multiply-adds) fills the issue slots beside it.  With the block's factor in registers a lane's own row is a forward substitution
            // of 36 operations; the second half's columns then take the first half's rank-8 update lane by lane (the block of the factor they
            // need, rows cc + 8 .. cc + 15, comes back from LDS as broadcasts).  Pinned pivots as before: L_jj = 1e64, the column scaled by 1e-64.
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int cc = c0 + 8 * h;
                const double *Tc = T + cc * LD + cc;          // the half's diagonal block
                double d[36], ri[8], thr[8];
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j <= i; ++j) d[i * (i + 1) / 2 + j] = Tc[i * LD + j];
#pragma unroll
                for (int j = 0; j < 8; ++j) thr[j] = d0s[cc + j];          // (kPanelBlocked: d0s holds the thresholds, kPivotRel x the original diagonal)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int jj = j * (j + 1) / 2 + j;
                    // a pivot under its threshold is replaced by 1e128: 1 / sqrt = 1e-64 and L_jj = 1e64 come out of the same operations (one
                    // select, no branch around the chain)
                    const double pj = d[jj] > thr[j] ? d[jj] : kPivotBig * kPivotBig;
                    const double rinv = rsqrt_h3(pj);
                    d[jj] = pj * rinv;
                    ri[j] = rinv;
#pragma unroll
                    for (int i = j + 1; i < 8; ++i) d[i * (i + 1) / 2 + j] *= rinv;
#pragma unroll
                    for (int i = j + 1; i < 8; ++i)
#pragma unroll
                        for (int k = j + 1; k <= i; ++k) d[i * (i + 1) / 2 + k] = fma(-d[i * (i + 1) / 2 + j], d[k * (k + 1) / 2 + j], d[i * (i + 1) / 2 + k]);
                }
                // this lane's row of the half: x L^T = a (right-looking); the rows of the diagonal block itself take the pivot entry from the factor
                double a[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) a[c] = T[lane * LD + cc + c];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const double x = a[j] * ri[j];
                    a[j] = lane == cc + j ? d[j * (j + 1) / 2 + j] : x;
#pragma unroll
                    for (int i = j + 1; i < 8; ++i) a[i] = fma(-x, d[i * (i + 1) / 2 + j], a[i]);
                }
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    if (lane >= cc + c) T[lane * LD + cc + c] = a[c];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (lane == cc + j) rinv_s[cc + j] = ri[j];
                if (h == 0) {
                    // the second half's columns, rows cc + 8 and below: a2[c] -= sum_k a[k] L[cc + 8 + c][cc + k] (this wavefront's own LDS writes
                    // above are in order with the reads below: one wavefront, one LDS queue)
                    // On the matrix core: per block of 16 rows two k-steps of v_mfma_f64_16x16x4_f64 (columns 8..15 of the product are not
                    // stored); ~12 instructions a row block against ~110 for the lane-by-lane form.  What lands above the diagonal of the second
                    // half's block, or in rows above the panel, is never read.
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    const double b0 = lm < 8 ? T[(cc + 8 + lm) * LD + cc + lq] : 0.0, b1 = lm < 8 ? T[(cc + 8 + lm) * LD + cc + 4 + lq] : 0.0;
                    for (int mb = p; mb < 4; ++mb) {
                        v4d acc;
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[r] = T[(16 * mb + lq + 4 * r) * LD + cc + 8 + (lm & 7)];
                        acc = mfma(-T[(16 * mb + lm) * LD + cc + lq], b0, acc);
                        acc = mfma(-T[(16 * mb + lm) * LD + cc + 4 + lq], b1, acc);
                        if (lm < 8) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) T[(16 * mb + lq + 4 * r) * LD + cc + 8 + lm] = acc[r];
                        }
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
            }
        } else if (w == 0) {
            // the panel: lane = row, the panel's 16 columns in registers; rows above the panel's diagonal block carry garbage nobody reads
            // The pivots run ahead on wave-uniform values: pivot j+1 = d - (u r2) u with u = the entry below pivot j, d = the diagonal
            // entry behind it (both read BEFORE column j is scaled, while 1 / sqrt(pivot j) is still being refined) and r2 = 1 / pivot j
            // -- twelve dependent operations a column; the scaling of the column and the updates of the panel's other columns,
            // a[c] -= (a[j] r2) u_c with the unscaled u_c, depend on r2 only and fill the issue slots beside the chain.
            // The other columns' multipliers u_c (the unscaled entries of column j in the rows of the panel's diagonal block) reach the lanes
            // through LDS: the column is written once (64 lanes, one ds_write_b64) and every u_c is a broadcast read -- a v_readlane pair per
            // multiplier made the column 70 vector instructions, 500 cycles; only the chain's own u_{j+1} still travels by v_readlane.
            double a[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) a[c] = T[lane * LD + c0 + c];
            const double d0l = d0s[lane];
            double piv = readlane_d(a[0], c0);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int pr = c0 + j;
                double *cb = colb + (j & 1) * NB;
                cb[lane] = a[j];
                const double unext = j + 1 < 16 ? readlane_d(a[j], pr + 1) : 0.0;
                const double dnext = j + 1 < 16 ? readlane_d(a[j + 1], pr + 1) : 0.0;
                const double thr = kPivotRel * readlane_d(d0l, pr);
                const double pj = piv;
                const bool ok = pj > thr;
                // 1 / pivot for the updates (on the chain: five dependent operations), 1 / sqrt(pivot) for the column's own entries (beside it)
                const double r2 = ok ? rcp_nr(pj) : kPivotBigInv * kPivotBigInv;
                const double rinv = ok ? rsqrt_nr(pj) : kPivotBigInv;
                if (j + 1 < 16) piv = fma(-(unext * r2), unext, dnext);                     // (the same operations lane pr + 1 applies to its own entry below)
                const double t = a[j] * r2;
#pragma unroll
                for (int c = j + 1; c < 16; ++c) a[c] = fma(-t, cb[c0 + c], a[c]);
                const double ljj = ok ? pj * rinv : kPivotBig;
                a[j] = lane == pr ? ljj : a[j] * rinv;
                if (lane == pr) rinv_s[pr] = rinv;
            }
#pragma unroll
            for (int c = 0; c < 16; ++c)
                if (lane >= c0 + c) T[lane * LD + c0 + c] = a[c];
        } else if (w == 7 && p > 0) {
            dinv_block(T, V, rinv_s, c0 - 16, lane);      // the previous panel's diagonal block, beside the chain
        }
        KAO_TICK();
        __syncthreads();
        KAO_TICK();
        // trailing 16 x 16 blocks (mb, nb), p < nb <= mb <= 3: -= panel(mb) panel(nb)^T
        const int rel = 3 - p;
        if (w < rel * (rel + 1) / 2) {
            const int ta = w >= 3 ? 2 : (w >= 1 ? 1 : 0), tb = w - ta * (ta + 1) / 2;
            const int mb = p + 1 + ta, nb = p + 1 + tb;
            v4d acc;
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = T[(16 * mb + lq + 4 * r) * LD + 16 * nb + lm];
#pragma unroll
            for (int s = 0; s < 4; ++s)
                acc = mfma(-T[(16 * mb + lm) * LD + c0 + 4 * s + lq], T[(16 * nb + lm) * LD + c0 + 4 * s + lq], acc);
#pragma unroll
            for (int r = 0; r < 4; ++r) T[(16 * mb + lq + 4 * r) * LD + 16 * nb + lm] = acc[r];
        }
        __syncthreads();
        KAO_TICK();
    }
    // The blocks of the inverse below the diagonal: X_ij = -Dinv_i sum_{j <= k < i} L_ik X_kj.  Behind the last panel only the last diagonal
    // block's inverse is missing (wavefront 7, ~4,000 cycles, 16 lanes).  Everything that does not need it runs beside it, a block column per
    // wavefront so that no workgroup barrier sits between dependent blocks (a wavefront's own LDS writes and reads are in order): the rows 1 and 2
    // of the inverse and the sums W_3j = sum_k L_3k X_kj of row 3; one barrier; then X_3j = -Dinv_3 W_3j.  (Level by level -- three rounds of two
    // barriers, the first waiting for wavefront 7 -- this tail was 7,900 cycles of the tile's 37,600.)
    auto block_sum = [&](int i, int j, int k1) {        // sum_{j <= k < k1} L_ik X_kj in the accumulator layout
        v4d acc = {0.0, 0.0, 0.0, 0.0};
        for (int k = j; k < k1; ++k)
#pragma unroll
            for (int s = 0; s < 4; ++s)
                acc = mfma(T[(16 * i + lm) * LD + 16 * k + 4 * s + lq], V[(16 * k + 4 * s + lq) * LD + 16 * j + lm], acc);
        return acc;
    };
    auto to_wb = [&](double *wb, v4d acc) {
#pragma unroll
        for (int r = 0; r < 4; ++r) wb[(lq + 4 * r) * kWbLd + lm] = acc[r];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    auto block_finish = [&](int i, int j, const double *wb) {   // X_ij = -Dinv_i W (W in wb) -> V
        v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < 4; ++s)
            acc = mfma(-V[(16 * i + lm) * LD + 16 * i + 4 * s + lq], wb[(4 * s + lq) * kWbLd + lm], acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) V[(16 * i + lq + 4 * r) * LD + 16 * j + lm] = acc[r];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    double *wb = Wb + (w < 3 ? w : 0) * 16 * kWbLd;
    if (w == 7) dinv_block(T, V, rinv_s, 48, lane);
    if (w == 0) {
        to_wb(wb, block_sum(1, 0, 1)); block_finish(1, 0, wb);
        to_wb(wb, block_sum(2, 0, 2)); block_finish(2, 0, wb);
        to_wb(wb, block_sum(3, 0, 3));
    } else if (w == 1) {
        to_wb(wb, block_sum(2, 1, 2)); block_finish(2, 1, wb);
        to_wb(wb, block_sum(3, 1, 3));
    } else if (w == 2) {
        to_wb(wb, block_sum(3, 2, 3));
    }
    __syncthreads();
    KAO_TICK();
    if (w < 3) block_finish(3, w, wb);
    __syncthreads();
    KAO_TICK();
#undef KAO_TICK
}
// the factored tile and its inverse leave LDS: L into the lower triangle of S's diagonal tile, L^-1 (zeros above the diagonal) into Linv
__device__ __forceinline__ void diag_out(const double *T, const double *V, double *S, int n, int base, double *inv, int tid) {
#pragma unroll
    for (int u = 0; u < NB * NB / kStepThreads; ++u) {
        const int e = tid + kStepThreads * u, r = e >> 6, c = e & 63;
        inv[e] = c <= r ? V[r * LD + c] : 0.0;
        if (c <= r) S[(size_t)(base + r) * n + base + c] = T[r * LD + c];
    }
}

// workgroup index -> (bi, bj), 0 <= bj <= bi: index = bi (bi + 1) / 2 + bj
__device__ __forceinline__ void pair_of(int idx, int &bi, int &bj) {
    bi = (int)((sqrtf(8.0f * (float)idx + 1.0f) - 1.0f) * 0.5f);
    while ((bi + 1) * (bi + 2) / 2 <= idx) ++bi;
    while (bi * (bi + 1) / 2 > idx) --bi;
    bj = idx - bi * (bi + 1) / 2;
}
// a 64 x 64 tile of S (row-major, leading dimension n) -> LDS (leading dimension LD): all eight loads of a thread are issued before the
// first LDS write (written as a loop the compiler waits for every load in turn: 8 x 2 dependent round trips, half of a step's time)
__device__ __forceinline__ void tile_to_lds(const double *__restrict__ src, int n, double *dst, int tid) {
    double v[NB * NB / kStepThreads];
#pragma unroll
    for (int u = 0; u < NB * NB / kStepThreads; ++u) { const int e = tid + kStepThreads * u; v[u] = src[(size_t)(e >> 6) * n + (e & 63)]; }
#pragma unroll
    for (int u = 0; u < NB * NB / kStepThreads; ++u) { const int e = tid + kStepThreads * u; dst[(e >> 6) * LD + (e & 63)] = v[u]; }
}
__device__ __forceinline__ void tile_pair_to_lds(const double *__restrict__ srcA, const double *__restrict__ srcB, int n, double *dstA, double *dstB, int tid) {
    double va[NB * NB / kStepThreads], vb[NB * NB / kStepThreads];
#pragma unroll
    for (int u = 0; u < NB * NB / kStepThreads; ++u) { const int e = tid + kStepThreads * u; va[u] = srcA[(size_t)(e >> 6) * n + (e & 63)]; vb[u] = srcB[(size_t)(e >> 6) * n + (e & 63)]; }
#pragma unroll
    for (int u = 0; u < NB * NB / kStepThreads; ++u) { const int e = tid + kStepThreads * u; dstA[(e >> 6) * LD + (e & 63)] = va[u]; dstB[(e >> 6) * LD + (e & 63)] = vb[u]; }
}

// tile (0, 0)
__global__ void __launch_bounds__(kStepThreads) k_chol_first(const double *stop, double *S, int n, const double *diag0, double *Linv, long long *dbg = nullptr) {
    if (stop && *stop != 0.0) return;
    extern __shared__ double lds[];
    double *T = lds, *V = lds + NB * LD, *Wb = V + NB * LD, *d0s = Wb + 3 * 16 * kWbLd, *rinv_s = d0s + NB, *colb = rinv_s + NB;
    const int tid = threadIdx.x;
    tile_to_lds(S, n, T, tid);
    if (tid < NB) d0s[tid] = (kPanelBlocked ? kPivotRel : 1.0) * diag0[tid];
    __syncthreads();
    if (dbg && tid == 0) dbg[30] = (long long)__builtin_amdgcn_s_memtime();
    potrf_inv(T, V, Wb, d0s, rinv_s, colb, tid, dbg);
    diag_out(T, V, S, n, 0, Linv, tid);
    if (dbg && tid == 0) dbg[31] = (long long)__builtin_amdgcn_s_memtime();
}

// The two wavefronts of a 16-row band of the tile share its four 16-column blocks as {0, 3} and {1, 2}: the triangular product
// with Linv^T needs 4 (nb + 1) k-steps for column block nb, so both get 20.
template <int NB0, int NB1, bool diag>
__device__ __forceinline__ void step_products(const double *__restrict__ Lk, double *bufA, double *bufB, bool first, bool write_upper, double *S, int n, int k, int ti, int tj,
                                              int tid, v4d c[2]) {
    const int w = tid >> 6, lane = tid & 63, lm = lane & 15, lq = lane >> 4, mb = w >> 1;
    constexpr int nbs[2] = {NB0, NB1};
    constexpr int cnt[2] = {4 * (NB0 + 1), 4 * (NB1 + 1)};
    constexpr int cmax = cnt[0] > cnt[1] ? cnt[0] : cnt[1];
    // rows of Linv_kk for the two column blocks (B operand: B[kk][nn] = Linv[nn][kk]), straight from global memory
    double b0[cnt[0]], b1[cnt[1]];
#pragma unroll
    for (int kk = 0; kk < cnt[0]; ++kk) b0[kk] = Lk[(16 * NB0 + lm) * NB + 4 * kk + lq];
#pragma unroll
    for (int kk = 0; kk < cnt[1]; ++kk) b1[kk] = Lk[(16 * NB1 + lm) * NB + 4 * kk + lq];
    v4d li[2] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}}, lj[2] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
#pragma unroll
    for (int kk = 0; kk < cmax; ++kk) {
        const double ai = bufA[(16 * mb + lm) * LD + 4 * kk + lq];
        const double aj = diag ? 0.0 : bufB[(16 * mb + lm) * LD + 4 * kk + lq];
        if (kk < cnt[0]) { li[0] = mfma(ai, b0[kk], li[0]); if (!diag) lj[0] = mfma(aj, b0[kk], lj[0]); }
        if (kk < cnt[1]) { li[1] = mfma(ai, b1[kk], li[1]); if (!diag) lj[1] = mfma(aj, b1[kk], lj[1]); }
    }
    __syncthreads();           // every wavefront has read its A operands: the tiles are replaced by the panel tiles
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            bufA[(16 * mb + lq + 4 * r) * LD + 16 * nbs[t] + lm] = li[t][r];
            if (!diag) bufB[(16 * mb + lq + 4 * r) * LD + 16 * nbs[t] + lm] = lj[t][r];
        }
    __syncthreads();
    if (write_upper) {         // L_ik^T into the upper triangle: element (c, m) of tile (k, ti)
#pragma unroll
        for (int u = 0; u < NB * NB / kStepThreads; ++u) {
            const int e = tid + kStepThreads * u;
            S[(size_t)(k * NB + (e >> 6)) * n + ti * NB + (e & 63)] = bufA[(e & 63) * LD + (e >> 6)];
        }
    }
    const double *bufY = diag ? bufA : bufB;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        if (diag && nbs[t] > mb) continue;        // above the diagonal of a diagonal tile
#pragma unroll
        for (int kk = 0; kk < 16; ++kk)
            c[t] = mfma(-bufA[(16 * mb + lm) * LD + 4 * kk + lq], bufY[(16 * nbs[t] + lm) * LD + 4 * kk + lq], c[t]);
    }
    if (!first) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * mb + lq + 4 * r, col = 16 * nbs[t] + lm;
                if (!diag || col <= row) S[(size_t)(ti * NB + row) * n + tj * NB + col] = c[t][r];
            }
    } else {
        __syncthreads();       // the panel tile in bufA has been read by everyone: tile (k+1, k+1) takes its place
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) bufA[(16 * mb + lq + 4 * r) * LD + 16 * nbs[t] + lm] = c[t][r];
    }
}

// tile column k is final up to its panel solve: workgroup (bi, bj), 0 <= bj <= bi < nt - k - 1, updates tile (k+1+bi, k+1+bj);
// workgroup (0, 0) then factors tile (k+1, k+1)
__global__ void __launch_bounds__(kStepThreads) k_chol_step(const double *stop, double *S, int n, int k, const double *diag0, double *Linv) {
    if (stop && *stop != 0.0) return;
    extern __shared__ double lds[];
    double *bufA = lds, *bufB = lds + NB * LD, *Wb = bufB + NB * LD, *d0s = Wb + 3 * 16 * kWbLd, *rinv_s = d0s + NB, *colb = rinv_s + NB;
    int bi, bj;
    pair_of((int)blockIdx.x, bi, bj);
    const int ti = k + 1 + bi, tj = k + 1 + bj, tid = threadIdx.x;
    const bool diag = bi == bj, first = blockIdx.x == 0;
    const int w = tid >> 6, lane = tid & 63, lm = lane & 15, lq = lane >> 4, mb = w >> 1;
    const int nb0 = (w & 1) ? 1 : 0, nb1 = 3 - nb0;
    // the tile to update, in the accumulator layout (issued first: its latency hides behind the panel products)
    v4d c[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int nb = t ? nb1 : nb0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * mb + lq + 4 * r, col = 16 * nb + lm;
            c[t][r] = (!diag || col <= row) ? S[(size_t)(ti * NB + row) * n + tj * NB + col] : 0.0;
        }
    }
    if (diag) tile_to_lds(S + (size_t)ti * NB * n + (size_t)k * NB, n, bufA, tid);
    else tile_pair_to_lds(S + (size_t)ti * NB * n + (size_t)k * NB, S + (size_t)tj * NB * n + (size_t)k * NB, n, bufA, bufB, tid);
    if (first && tid < NB) d0s[tid] = (kPanelBlocked ? kPivotRel : 1.0) * diag0[(k + 1) * NB + tid];
    __syncthreads();
    const double *Lk = Linv + (size_t)k * NB * NB;
    if (diag) {
        if (w & 1) step_products<1, 2, true>(Lk, bufA, bufB, first, bj == 0, S, n, k, ti, tj, tid, c);
        else step_products<0, 3, true>(Lk, bufA, bufB, first, bj == 0, S, n, k, ti, tj, tid, c);
    } else {
        if (w & 1) step_products<1, 2, false>(Lk, bufA, bufB, first, bj == 0, S, n, k, ti, tj, tid, c);
        else step_products<0, 3, false>(Lk, bufA, bufB, first, bj == 0, S, n, k, ti, tj, tid, c);
    }
    if (!first) return;
    __syncthreads();
    potrf_inv(bufA, bufB, Wb, d0s, rinv_s, colb, tid);
    diag_out(bufA, bufB, S, n, (k + 1) * NB, Linv + (size_t)(k + 1) * NB * NB, tid);
}

// lower tile (i, j) <- transpose of upper tile (j, i), i > j
__global__ void __launch_bounds__(256) k_chol_mirror(const double *stop, double *S, int n) {
    if (stop && *stop != 0.0) return;
    __shared__ double Ts[NB * LD];
    int bi, bj;
    pair_of((int)blockIdx.x, bi, bj);
    const int i = bi + 1, j = bj;       // 0 <= j < i
    double v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) { const int e = threadIdx.x + 256 * u; v[u] = S[(size_t)(j * NB + (e >> 6)) * n + i * NB + (e & 63)]; }
#pragma unroll
    for (int u = 0; u < 16; ++u) { const int e = threadIdx.x + 256 * u; Ts[(e >> 6) * LD + (e & 63)] = v[u]; }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 16; ++u) { const int e = threadIdx.x + 256 * u; S[(size_t)(i * NB + (e >> 6)) * n + j * NB + (e & 63)] = Ts[(e & 63) * LD + (e >> 6)]; }
}

// ---- L z = r then L^T x = z, in place in r: one workgroup per row tile -------------------------------------------------------------
// xz[2 n]: the exchange vectors (forward results, then backward results), preset to kUnset (all ones: a NaN no arithmetic produces).
// Workgroup i subtracts L_ij z_j for j < i as the z_j appear, publishes z_i = Linv_ii acc; then the same downwards on L^T.  A workgroup
// only ever waits for values of workgroups that hold no resource it needs; the spin is bounded (a stalled solve raises the stop flag).
constexpr long long kUnsetBits = -1LL;
constexpr int kSpinMax = 1 << 22;
__global__ void __launch_bounds__(256) k_trsv(double *stop, const double *gate, const double *S, int n, double *r, const double *Linv, double *xz) {
    if (stop && *stop != 0.0) return;
    if (gate && *gate == 0.0) return;
    __shared__ double acc[NB], xj[NB], part[4][NB];
    const int t = threadIdx.x, a = t & 63, kq = t >> 6, i = blockIdx.x, nt = n / NB;
    const double *inv = Linv + (size_t)i * NB * NB;
    auto fetch = [&](const double *src) {          // the 64 values of one published tile -> xj
        if (t < NB) {
            double v = __hip_atomic_load(src + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int spins = 0;
            while (__double_as_longlong(v) == kUnsetBits && spins < kSpinMax) {
                __builtin_amdgcn_s_sleep(1);
                v = __hip_atomic_load(src + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ++spins;
            }
            if (spins >= kSpinMax && stop) *stop = 3.0;
            xj[t] = v;
        }
        __syncthreads();
    };
    auto tile_load = [&](const double *base, size_t ks, size_t as, double v[16]) {
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = __builtin_nontemporal_load(base + (size_t)(kq * 16 + u) * ks + (size_t)a * as);
    };
    auto tile_apply = [&](const double v[16], bool subtract) {
        double p0 = 0.0, p1 = 0.0;
#pragma unroll
        for (int u = 0; u < 16; u += 2) { p0 += v[u] * xj[kq * 16 + u]; p1 += v[u + 1] * xj[kq * 16 + u + 1]; }
        part[kq][a] = p0 + p1;
        __syncthreads();
        if (t < NB) { const double sum = (part[0][t] + part[1][t]) + (part[2][t] + part[3][t]); acc[t] = subtract ? acc[t] - sum : sum; }
        __syncthreads();
    };
    // the diagonal tile's inverse in both orientations, fetched now: nothing but the polled value is waited for on the critical path
    double vf[16], vb[16];
    tile_load(inv, 1, (size_t)NB, vf);                                              // Linv[a][k]
    tile_load(inv, (size_t)NB, 1, vb);                                              // Linv^T[a][k] = Linv[k][a]
    if (t < NB) acc[t] = r[(size_t)i * NB + t];
    __syncthreads();
    for (int j = 0; j < i; ++j) {
        double v[16];
        tile_load(S + (size_t)j * NB * n + (size_t)i * NB, (size_t)n, 1, v);        // L[i*64 + a][j*64 + k] from the upper copy
        fetch(xz + (size_t)j * NB);
        tile_apply(v, true);
    }
    {
        if (t < NB) xj[t] = acc[t];
        __syncthreads();
        tile_apply(vf, false);
        if (t < NB) __hip_atomic_store(xz + (size_t)i * NB + t, acc[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    for (int j = nt - 1; j > i; --j) {
        double v[16];
        tile_load(S + (size_t)j * NB * n + (size_t)i * NB, (size_t)n, 1, v);        // L^T[i*64 + a][j*64 + k] = L[j*64 + k][i*64 + a]: the lower triangle
        fetch(xz + (size_t)n + (size_t)j * NB);
        tile_apply(v, true);
    }
    {
        if (t < NB) xj[t] = acc[t];
        __syncthreads();
        tile_apply(vb, false);
        if (t < NB) {
            __hip_atomic_store(xz + (size_t)n + (size_t)i * NB + t, acc[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            r[(size_t)i * NB + t] = acc[t];
        }
    }
}

bool g_attr_set[kMaxDevices] = {};
void set_attrs() {
    const int d = cur_device();
    if (d >= 0 && d < kMaxDevices && g_attr_set[d]) return;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_chol_first), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kStepLds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_chol_step), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kStepLds);
    if (d >= 0 && d < kMaxDevices) g_attr_set[d] = true;
}

}  // namespace

// Cholesky of the n x n matrix in the lower triangle of S (row-major, n a multiple of 64; diag0 = its diagonal), enqueued on `st`:
// afterwards the lower triangle holds L, the upper triangle L^T (tile-wise), Linv[(n / 64)][64][64] the inverses of the diagonal tiles.
// `stop` (device, may be null): every kernel returns at its first line when *stop != 0.
void chol_enqueue(void *stream, const double *stop, double *S, int n, const double *diag0, double *Linv) {
    set_attrs();
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nt = n / NB;
    hipLaunchKernelGGL(k_chol_first, dim3(1), dim3(kStepThreads), kStepLds, st, stop, S, n, diag0, Linv, (long long *)nullptr);
    for (int k = 0; k + 1 < nt; ++k) {
        const int nrem = nt - k - 1;
        hipLaunchKernelGGL(k_chol_step, dim3(nrem * (nrem + 1) / 2), dim3(kStepThreads), kStepLds, st, stop, S, n, k, diag0, Linv);
    }
    if (nt > 1) hipLaunchKernelGGL(k_chol_mirror, dim3(nt * (nt - 1) / 2), dim3(256), 0, st, stop, S, n);
}
// S x = r in place (after chol_enqueue); xz[2 n] scratch.  `gate` (device, may be null): skipped when *gate == 0.
void trsv_enqueue(void *stream, double *stop, const double *gate, const double *S, int n, double *r, const double *Linv, double *xz) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    (void)hipMemsetAsync(xz, 0xFF, sizeof(double) * 2 * (size_t)n, st);
    hipLaunchKernelGGL(k_trsv, dim3(n / NB), dim3(256), 0, st, stop, gate, S, n, r, Linv, xz);
}

}  // namespace kao

// Test hook (include/kao.h): the dense kernels alone on a caller's matrix.
extern "C" int kao_dense_spd_test(const double *A, int32_t n, const double *rhs, double *factor, double *linv, double *x, double ms[2]) {
    int rc = require_init();
    if (rc) return rc;
    if (!A || n < NB || n % NB != 0 || n > 64 * 160) return fail(KAO_ERR_INVALID, "kao_dense_spd_test: n must be a multiple of 64 in 64..10240");
    const size_t nn = (size_t)n * n;
    double *dS = nullptr, *dA = nullptr, *dd = nullptr, *dL = nullptr, *dr = nullptr, *dxz = nullptr;
    hipStream_t st = nullptr;
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
    auto cleanup = [&]() {
        for (void *p : {(void *)dS, (void *)dA, (void *)dd, (void *)dL, (void *)dr, (void *)dxz}) if (p) (void)hipFree(p);
        for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e);
        if (st) (void)hipStreamDestroy(st);
    };
#define SPD_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { cleanup(); return fail(KAO_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); } } while (0)
    SPD_TRY(hipStreamCreate(&st));
    for (hipEvent_t &e : ev) SPD_TRY(hipEventCreate(&e));
    SPD_TRY(hipMalloc(&dS, nn * 8)); SPD_TRY(hipMalloc(&dA, nn * 8)); SPD_TRY(hipMalloc(&dd, (size_t)n * 8)); SPD_TRY(hipMalloc(&dL, (size_t)n * NB * 8));
    SPD_TRY(hipMalloc(&dr, (size_t)n * 8)); SPD_TRY(hipMalloc(&dxz, (size_t)n * 16));
    std::vector<double> lower(nn, 0.0), dg((size_t)n), rr((size_t)n, 1.0);
    for (int i = 0; i < n; ++i) { for (int j = 0; j <= i; ++j) lower[(size_t)i * n + j] = A[(size_t)i * n + j]; dg[(size_t)i] = A[(size_t)i * n + i]; }
    if (rhs) std::memcpy(rr.data(), rhs, (size_t)n * 8);
    SPD_TRY(hipMemcpy(dA, lower.data(), nn * 8, hipMemcpyHostToDevice));
    SPD_TRY(hipMemcpy(dd, dg.data(), (size_t)n * 8, hipMemcpyHostToDevice));
    for (int rep = 0; rep < 2; ++rep) {          // the second pass is the timed one
        SPD_TRY(hipMemcpyAsync(dS, dA, nn * 8, hipMemcpyDeviceToDevice, st));
        SPD_TRY(hipMemcpyAsync(dr, rr.data(), (size_t)n * 8, hipMemcpyHostToDevice, st));
        SPD_TRY(hipEventRecord(ev[0], st));
        chol_enqueue(st, nullptr, dS, n, dd, dL);
        SPD_TRY(hipEventRecord(ev[1], st));
        trsv_enqueue(st, nullptr, nullptr, dS, n, dr, dL, dxz);
        SPD_TRY(hipEventRecord(ev[2], st));
        SPD_TRY(hipStreamSynchronize(st));
        SPD_TRY(hipGetLastError());
    }
    if (std::getenv("KAO_CHOL_DEBUG")) {     // clock ticks (s_memtime) of the diagonal tile's phases, to stderr
        long long *dbgd = nullptr, hd[32] = {0};
        SPD_TRY(hipMalloc(&dbgd, sizeof hd));
        SPD_TRY(hipMemset(dbgd, 0, sizeof hd));
        SPD_TRY(hipMemcpyAsync(dS, dA, nn * 8, hipMemcpyDeviceToDevice, st));
        set_attrs();
        hipLaunchKernelGGL(k_chol_first, dim3(1), dim3(kStepThreads), kStepLds, st, (const double *)nullptr, dS, n, dd, dL, dbgd);
        SPD_TRY(hipStreamSynchronize(st));
        SPD_TRY(hipMemcpy(hd, dbgd, sizeof hd, hipMemcpyDeviceToHost));
        (void)hipFree(dbgd);
        std::fprintf(stderr, "[kao-chol] diagonal tile, ticks since entry (s_memtime, 100 MHz):");
        for (int q = 0; q < 30 && (q == 0 || hd[q]); ++q) std::fprintf(stderr, " %lld", hd[q] - hd[30]);
        std::fprintf(stderr, " | whole kernel body %lld\n", hd[31] - hd[30]);
    }
    float m0 = 0, m1 = 0;
    SPD_TRY(hipEventElapsedTime(&m0, ev[0], ev[1])); SPD_TRY(hipEventElapsedTime(&m1, ev[1], ev[2]));
    if (ms) { ms[0] = m0; ms[1] = m1; }
    if (factor) SPD_TRY(hipMemcpy(factor, dS, nn * 8, hipMemcpyDeviceToHost));
    if (linv) SPD_TRY(hipMemcpy(linv, dL, (size_t)n * NB * 8, hipMemcpyDeviceToHost));
    if (x) SPD_TRY(hipMemcpy(x, dr, (size_t)n * 8, hipMemcpyDeviceToHost));
#undef SPD_TRY
    cleanup();
    return KAO_OK;
}
