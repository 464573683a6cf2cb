// kao_cycle.hip -- KAO-CX: cyclic-exchange improvement of a feasible assignment, gfx950 only.
//
// K-search moves one or two slots at a time; on rigid instances (replicas per broker and per rack fixed exactly, as in
// every drifted topic whose P*RF is a multiple of B) the last improvements are cyclic exchanges over 4..10 partitions whose
// intermediate states are all infeasible.  KAO-CX finds them by shortest paths instead of by chance:
//   k_cx_edges   : two transfer graphs on the brokers -- F (a follower slot takes another broker: one replica unit moves),
//                  S (leader and follower of one partition swap roles: one leader unit moves) -- cheapest slot per broker
//                  pair by 64-bit atomicMin of (cost, slot);
//   k_cx_edges_l : L, generalised leader transfers (five variants of "the partition led by u gets leader v", each priced with
//                  the F path that compensates its replica effect), built on the closure of F;
//   k_cx_dist0   : edge keys -> cost matrices, slack node Z (brokers with room inside their band absorb / give a unit);
//   k_cx_square  : min-plus squaring with the midpoint of every pair, three times: cheapest paths of <= 8 edges;
//   k_cx_seeds   : for every partition every new row that replaces <= 2 replicas (one by a current replica) with any
//                  leader, priced as gain - closure of its replica imbalance (F) - of its leader imbalance (S);
// the host unrolls the best candidates into slot changes (a partition is used once), K-eval scores them exactly, and the
// best -- or a merge of partition-disjoint ones -- becomes the next assignment.  Every definition follows
// oracle/kao_cycle.py, which the parity tests compare with bit for bit.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <array>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <unordered_map>
#include <string>
#include <vector>

#include "../../include/kao.h"
#include "kao_internal.h"

namespace kao {
namespace {

constexpr int kCxInf = 1 << 17;
constexpr int kCxBias = 1 << 16;
constexpr int kCxLevels = 3;
constexpr int kCxLayers = 3;      // 0 F (follower moves), 1 S (role swaps), 2 L (leader replacements)
constexpr int kCxMaxEval = 512;
constexpr int64_t kCxBulkSlots = 131072;   // beyond: cycle candidates are merged before they are scored (cx_round, bulk mode)
constexpr int kCxMaxRF = 8;
// The per-partition kernels (k_cx_edges, k_cx_edges_l, k_cx_seeds) keep a row and the rows derived from it in small per-thread
// arrays indexed at run time; instantiated for MR = 4 (RF <= 4: the arrays stay in registers) and MR = 8 (RF 5..8) like K-search
// and K-bound -- with one size for all, RF 3 paid for RF 8 (k_cx_edges_l 540 -> 2,109 us at 1000 x 30000).
constexpr unsigned long long kNoEdge = ~0ull;

struct CxParams {
    int32_t B, R, P, RF, rfc, n, np;   // n = B + R + 1 (slack nodes: Z_r = B + r per rack, Z = B + R), np = row stride of the matrices (multiple of 64)
    int32_t rep_lo, rep_hi, lead_lo, lead_hi, prack_lo, prack_hi, rack_lo, rack_hi;
    int32_t w00, w01, w10, w11;
    int32_t ncfg;
    const int32_t *bw, *bwl;   // broker weights by dense index (kao_topic.broker_w / broker_wl), or null
};

// objective weight of broker b in role nr on a partition whose current replicas are cur[0..rfc) (README.md:145-146), plus the
// broker's own weights when the topic carries them (broker_w on both variables of a broker, broker_wl on the `_l` one)
__device__ __forceinline__ int cx_wt(const CxParams &q, const uint16_t *cur, int b, int nr) {
    int w = 0;
    for (int k = 0; k < q.rfc; ++k)
        if ((int)cur[k] == b) w = k == 0 ? (nr == 0 ? q.w00 : q.w01) : (nr == 0 ? q.w10 : q.w11);
    if (q.bw) w += q.bw[b];
    if (q.bwl && nr == 0) w += q.bwl[b];
    return w;
}

// C7 (README.md:178-180) of a row made of `base[0..nb)` plus one more broker of rack ry: nb <= MR - 1
template <int MR> struct CxBase {
    int rk[MR];
    int nb, ndef;
    bool over;
};
template <int MR> __device__ __forceinline__ CxBase<MR> cx_base(const CxParams &q, const uint8_t *rack, const int *base, int nb) {
    CxBase<MR> o;
    o.nb = nb; o.over = false;
    int distinct = 0, deficient_present = 0;
#pragma unroll
    for (int i = 0; i < nb; ++i) o.rk[i] = rack[base[i]];
#pragma unroll
    for (int i = 0; i < nb; ++i) {
        int cnt = 0; bool first = true;
#pragma unroll
        for (int j = 0; j < nb; ++j) { cnt += o.rk[j] == o.rk[i]; first = first && !(j < i && o.rk[j] == o.rk[i]); }
        if (cnt > q.prack_hi) o.over = true;
        if (first) { ++distinct; deficient_present += cnt < q.prack_lo; }
    }
    o.ndef = q.prack_lo > 0 ? (q.R - distinct) + deficient_present : 0;
    return o;
}
template <int MR> __device__ __forceinline__ bool cx_completes(const CxParams &q, const CxBase<MR> &b, int ry) {
    int cnt = 0;
#pragma unroll
    for (int i = 0; i < b.nb; ++i) cnt += b.rk[i] == ry;
    if (cnt + 1 > q.prack_hi) return false;
    if (b.ndef == 0) return true;
    return b.ndef == 1 && cnt < q.prack_lo && cnt + 1 >= q.prack_lo;
}

// atomicMin behind a plain look (round 4): keys only ever decrease, so a key that does not beat the value read -- however stale --
// cannot beat the current one either.  At 1000 x 100,000 every partition offered 2,000 keys to a table of 10^6 entries (2e8
// 64-bit atomics a build); nearly all of them lose to what is already there.  Same table, bit for bit.
__device__ __forceinline__ void cx_offer(unsigned long long *e, unsigned long long key) {
    if (key < __hip_atomic_load(e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(e, key);   // agent scope: read at L2, where the atomics land
}

// ---- edges: one wavefront per partition -------------------------------------------------------------------------------
template <int MR>
__global__ __launch_bounds__(256) void k_cx_edges(CxParams q, const uint16_t *A, const uint16_t *cur, const uint8_t *rack,
                                                  unsigned long long *EF, unsigned long long *ES) {
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (p >= q.P) return;
    int row[MR];
    for (int k = 0; k < q.RF; ++k) row[k] = A[(size_t)p * q.RF + k];
    const uint16_t *c = cur + (size_t)p * q.rfc;
    for (int k = 1; k < q.RF; ++k) {
        const int u = row[k];
        int base[MR]; int nb = 0;
        for (int j = 0; j < q.RF; ++j) if (j != k) base[nb++] = row[j];
        const CxBase<MR> cb = cx_base<MR>(q, rack, base, nb);
        if (!cb.over && cb.ndef <= 1) {
            const int wu = cx_wt(q, c, u, 1);
            for (int v = lane; v < q.B; v += 64) {
                bool in = false;
                for (int j = 0; j < q.RF; ++j) in = in || row[j] == v;
                if (in || !cx_completes(q, cb, rack[v])) continue;
                const int cost = wu - cx_wt(q, c, v, 1);
                const unsigned long long key = ((unsigned long long)(unsigned)(cost + kCxBias) << 32) | (unsigned)(p * q.RF + k);
                cx_offer(&EF[(size_t)u * q.np + v], key);
            }
        }
        if (lane == 0) {
            const int a = row[0];
            const int cs = cx_wt(q, c, a, 0) + cx_wt(q, c, u, 1) - cx_wt(q, c, u, 0) - cx_wt(q, c, a, 1);
            const unsigned long long key = ((unsigned long long)(unsigned)(cs + kCxBias) << 32) | (unsigned)(p * q.RF + k);
            cx_offer(&ES[(size_t)a * q.np + u], key);
        }
    }
}

// ---- generalised leader-transfer edges (layer L), built on the level-3 closure of F: one wavefront per partition ------------
// Edge u -> v = "the partition led by u gets leader v"; nominally a replica unit and a leader unit move u -> v.  Variants whose
// replica effect differs carry the cost of the compensating F path:
//   0 plain   : v replaces u                                                   1 demote : v enters, u stays as follower, slot k
//   2 promote : slot k becomes leader, u leaves, y enters      (+ DF[y][v])               leaves            (+ DF[u][row[k]])
//   3 swap    : slot k becomes leader, u follower              (+ DF[u][v])    4 double : v replaces u and slot k takes y, one of
//                                                                                         them a current replica (+ DF[y][row[k]])
// key = (cost + 2^16) << 44 | p << 20 | variant << 16 | k << 12 | y.  `plain_only`: F has improving cycles, nothing is priced on it.
__device__ __forceinline__ unsigned long long cx_lkey(int cost, int p, int var, int k, int y) {
    return ((unsigned long long)(unsigned)(cost + kCxBias) << 44) | ((unsigned long long)(unsigned)p << 20) | ((unsigned long long)var << 16) |
           ((unsigned long long)k << 12) | (unsigned long long)y;
}
__device__ __forceinline__ long long cx_wave_min(long long v) {
    for (int off = 32; off > 0; off >>= 1) {
        const long long o = __shfl_xor(v, off, 64);
        v = o < v ? o : v;
    }
    return v;
}
// DFt[b][y] = DF[y][b]: k_cx_edges_l reads the compensation DF[y][b] for EVERY entering broker y (lane = y) of a fixed b -- one
// cache line per lane in DF's row-major layout (3,000 such reads per partition: most of the kernel's time at 1000 x 100,000),
// consecutive words in the transpose.  64 x 64 tiles through LDS; np is a multiple of 64.
__global__ __launch_bounds__(256) void k_cx_transpose(int np, const int32_t *__restrict__ D, int32_t *__restrict__ Dt) {
    __shared__ int32_t tile[64][65];
    const int bx = blockIdx.x * 64, by = blockIdx.y * 64, tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4) tile[r][tx] = D[(size_t)(by + r) * np + bx + tx];
    __syncthreads();
    for (int r = ty; r < 64; r += 4) Dt[(size_t)(bx + r) * np + by + tx] = tile[tx][r];
}

// KRF: the replication factor as a compile-time constant (2..4 with MR = 4; 0 = read it from q).  Round 4: with a run-time RF
// the loops over the row stay rolled, `row[k]`, `others[no++]`, `base[nb++]` are indexed at run time and the arrays live in
// scratch (32 B per lane even at MR = 4) -- and `inrow`, called for every target broker, reads them from there: 1.74 ms per call
// at 1000 x 30000, 3.6 ms at 1000 x 100,000.  With RF known every index is a constant and the rows stay in registers.
template <int MR, int KRF>
__global__ __launch_bounds__(256) void k_cx_edges_l(CxParams q, const uint16_t *A, const uint16_t *cur, const uint8_t *rack, const int32_t *DF,
                                                    const int32_t *DFt, int plain_only, unsigned long long *EL) {
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (p >= q.P) return;
    const int RF = KRF ? KRF : q.RF;
    int row[MR];
    #pragma unroll
    for (int k = 0; k < RF; ++k) row[k] = A[(size_t)p * RF + k];
    const uint16_t *c = cur + (size_t)p * q.rfc;
    const int u = row[0];
    int w0 = cx_wt(q, c, u, 0);
    #pragma unroll
    for (int k = 1; k < RF; ++k) w0 += cx_wt(q, c, row[k], 1);
    const int wlu = cx_wt(q, c, u, 0), wfu = cx_wt(q, c, u, 1);
    auto inrow = [&](int x) { bool in = false;
#pragma unroll
        for (int j = 0; j < RF; ++j) in = in || row[j] == x;
        return in; };
    {   // 0 plain
        int base[MR]; int nb = 0;
        #pragma unroll
        for (int j = 1; j < RF; ++j) base[nb++] = row[j];
        const CxBase<MR> cb = cx_base<MR>(q, rack, base, nb);
        if (!cb.over && cb.ndef <= 1)
            for (int v = lane; v < q.B; v += 64) {
                if (inrow(v) || !cx_completes(q, cb, rack[v])) continue;
                const int cost = wlu - cx_wt(q, c, v, 0);
                if (cost < kCxInf / 2) cx_offer(&EL[(size_t)u * q.np + v], cx_lkey(cost, p, 0, 0, 0));
            }
    }
    if (plain_only) return;
    #pragma unroll
    for (int k = 1; k < RF; ++k) {
        const int b = row[k];
        int others[MR], no = 0, osum = 0;
        #pragma unroll
        for (int j = 1; j < RF; ++j) if (j != k) { others[no++] = row[j]; osum += cx_wt(q, c, row[j], 1); }
        {   // 1 demote: row' = (v; u, others)
            int base[MR]; int nb = 0;
            base[nb++] = u;
            #pragma unroll
            for (int j = 0; j < no; ++j) base[nb++] = others[j];
            const CxBase<MR> cb = cx_base<MR>(q, rack, base, nb);
            const int comp = DF[(size_t)u * q.np + b];
            if (!cb.over && cb.ndef <= 1 && comp < kCxInf)
                for (int v = lane; v < q.B; v += 64) {
                    if (inrow(v) || !cx_completes(q, cb, rack[v])) continue;
                    const int cost = w0 - (cx_wt(q, c, v, 0) + wfu + osum) + comp;
                    if (cost < kCxInf / 2) cx_offer(&EL[(size_t)u * q.np + v], cx_lkey(cost, p, 1, k, 0));
                }
        }
        {   // 2 promote: v = row[k]; row' = (v; y, others), best y
            int base[MR]; int nb = 0;
            base[nb++] = b;
            #pragma unroll
            for (int j = 0; j < no; ++j) base[nb++] = others[j];
            const CxBase<MR> cb = cx_base<MR>(q, rack, base, nb);
            long long best = LLONG_MAX;
            if (!cb.over && cb.ndef <= 1) {
                const int wlv = cx_wt(q, c, b, 0);
                for (int y = lane; y < q.B; y += 64) {
                    if (inrow(y) || !cx_completes(q, cb, rack[y])) continue;
                    const long long cost = (long long)w0 - (wlv + cx_wt(q, c, y, 1) + osum) + DFt[(size_t)b * q.np + y];
                    const long long key = (cost + (1ll << 30)) * 4096 + y;
                    best = key < best ? key : best;
                }
            }
            best = cx_wave_min(best);
            if (lane == 0 && best != LLONG_MAX) {
                const long long cost = best / 4096 - (1ll << 30);
                if (cost < kCxInf / 2) cx_offer(&EL[(size_t)u * q.np + b], cx_lkey((int)cost, p, 2, k, (int)(best % 4096)));
            }
        }
        if (lane == 0) {   // 3 swap
            const int cost = wlu + cx_wt(q, c, b, 1) - cx_wt(q, c, b, 0) - wfu + DF[(size_t)u * q.np + b];
            if (cost < kCxInf / 2) cx_offer(&EL[(size_t)u * q.np + b], cx_lkey(cost, p, 3, k, 0));
        }
        // 4 double: new leader v (not in the row) and slot k takes y; v or y is a current replica i of the partition
        for (int ii = 0; ii < q.rfc; ++ii) {
            const int i = c[ii];
            if (i >= q.B || inrow(i)) continue;
            int base[MR]; int nb = 0;
            #pragma unroll
            for (int j = 0; j < no; ++j) base[nb++] = others[j];
            base[nb++] = i;
            const CxBase<MR> cb = cx_base<MR>(q, rack, base, nb);
            if (cb.over || cb.ndef > 1) continue;
            const int compi = DF[(size_t)i * q.np + b];
            if (compi < kCxInf) {   // (a) y = i, any v
                const int wfi = cx_wt(q, c, i, 1);
                for (int v = lane; v < q.B; v += 64) {
                    if (inrow(v) || v == i || !cx_completes(q, cb, rack[v])) continue;
                    const int cost = w0 - (cx_wt(q, c, v, 0) + wfi + osum) + compi;
                    if (cost < kCxInf / 2) cx_offer(&EL[(size_t)u * q.np + v], cx_lkey(cost, p, 4, k, i));
                }
            }
            {   // (b) v = i, best y
                long long best = LLONG_MAX;
                const int wli = cx_wt(q, c, i, 0);
                for (int y = lane; y < q.B; y += 64) {
                    if (inrow(y) || y == i || !cx_completes(q, cb, rack[y])) continue;
                    const long long cost = (long long)w0 - (wli + cx_wt(q, c, y, 1) + osum) + DFt[(size_t)b * q.np + y];
                    const long long key = (cost + (1ll << 30)) * 4096 + y;
                    best = key < best ? key : best;
                }
                best = cx_wave_min(best);
                if (lane == 0 && best != LLONG_MAX) {
                    const long long cost = best / 4096 - (1ll << 30);
                    if (cost < kCxInf / 2) cx_offer(&EL[(size_t)u * q.np + i], cx_lkey((int)cost, p, 4, k, (int)(best % 4096)));
                }
            }
        }
    }
}

// ---- edge keys -> level-0 cost matrix (with the slack nodes) ----------------------------------------------------------
// mode 0 (F, replica units): u -> Z_rack(u) when u may take one more replica inside its band, Z_rack(v) -> v when v may give one
//        up -- a path through Z_r alone leaves every rack total as it is --, Z_r -> Z when rack r may take one more (C6,
//        README.md:173-175), Z -> Z_r when it may give one up.  Round 4; with ONE slack node a path through it moved a unit
//        between racks, K-eval rejected the realisation and the rack-conserving path of equal cost was never found: on topics
//        whose bands have slack (P*RF not a multiple of B) the fixpoints ended 8-21 units below the optimum.
// mode 1 (S, leader units: they know no racks): u -> Z, Z -> v by the leader band.   mode 2 (L): no slack edges.
// cnt: c[B] | l[B] | K[R] (replicas, leaders per broker; replicas per rack)
__global__ __launch_bounds__(256) void k_cx_dist0(CxParams q, const unsigned long long *E, const int32_t *cnt, const uint8_t *rack, int mode, int shift, int32_t *D) {
    const int j = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
    if (j >= q.np) return;
    int d = kCxInf;
    const int B = q.B, Z = q.B + q.R;
    if (i < q.n && j < q.n) {
        if (i == j) d = 0;
        else if (i < B && j < B) {
            const unsigned long long key = E[(size_t)i * q.np + j];
            d = key == kNoEdge ? kCxInf : (int)(key >> shift) - kCxBias;
        } else if (mode == 0) {
            if (i < B && j < Z) d = ((int)rack[i] == j - B && cnt[i] < q.rep_hi) ? 0 : kCxInf;
            else if (j < B && i < Z) d = ((int)rack[j] == i - B && cnt[j] > q.rep_lo) ? 0 : kCxInf;
            else if (i >= B && i < Z && j == Z) d = cnt[2 * B + (i - B)] < q.rack_hi ? 0 : kCxInf;
            else if (i == Z && j >= B && j < Z) d = cnt[2 * B + (j - B)] > q.rack_lo ? 0 : kCxInf;
        } else if (mode == 1) {
            if (i < B && j == Z) d = cnt[B + i] < q.lead_hi ? 0 : kCxInf;
            else if (i == Z && j < B) d = cnt[B + j] > q.lead_lo ? 0 : kCxInf;
        }
    }
    D[(size_t)i * q.np + j] = d;
}

// ---- min-plus squaring with midpoints: 64 x 64 outputs per workgroup, 4 x 4 per lane, LDS tiles of 16 midpoints --------
// composite key ((sum + 2^18) << 12) | prio, prio = 0 for k == i (the pair's own entry), k + 1 otherwise: the smallest key is
// the cheapest midpoint, "no midpoint" preferred, then the lowest index -- the oracle's rule.
__global__ __launch_bounds__(256) void k_cx_square(int np, const int32_t *__restrict__ D, int32_t *__restrict__ Dn, uint16_t *__restrict__ mid) {
    __shared__ int32_t sa[64][17];   // D[i0 + r][k0 + kk]
    __shared__ int32_t sb[16][64];   // D[k0 + kk][j0 + c]
    const int i0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
    const int tr = (threadIdx.x >> 4) * 4, tc = (threadIdx.x & 15) * 4;
    uint32_t best[4][4];
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) best[a][b] = 0xFFFFFFFFu;
    for (int k0 = 0; k0 < np; k0 += 16) {
        for (int e = threadIdx.x; e < 64 * 16; e += 256) {
            const int r = e >> 4, kk = e & 15;
            sa[r][kk] = D[(size_t)(i0 + r) * np + k0 + kk];
        }
        for (int e = threadIdx.x; e < 16 * 64; e += 256) {
            const int kk = e >> 6, c = e & 63;
            sb[kk][c] = D[(size_t)(k0 + kk) * np + j0 + c];
        }
        __syncthreads();
#pragma unroll 4
        for (int kk = 0; kk < 16; ++kk) {
            const int k = k0 + kk;
            int bv[4];
            for (int b = 0; b < 4; ++b) bv[b] = sb[kk][tc + b];
            for (int a = 0; a < 4; ++a) {
                const int av = sa[tr + a][kk] + (1 << 18);
                const uint32_t prio = (k == i0 + tr + a) ? 0u : (uint32_t)(k + 1);
                for (int b = 0; b < 4; ++b) {
                    const uint32_t key = ((uint32_t)(av + bv[b]) << 12) | prio;
                    best[a][b] = key < best[a][b] ? key : best[a][b];
                }
            }
        }
        __syncthreads();
    }
    for (int a = 0; a < 4; ++a)
        for (int b = 0; b < 4; ++b) {
            const int i = i0 + tr + a, j = j0 + tc + b;
            const int sum = (int)(best[a][b] >> 12) - (1 << 18);
            const uint32_t prio = best[a][b] & 0xFFFu;
            Dn[(size_t)i * np + j] = sum < kCxInf ? sum : kCxInf;
            mid[(size_t)i * np + j] = (uint16_t)(prio == 0 ? i : (int)prio - 1);
        }
}

// The same squaring for SMALL matrices, split over the midpoints (round 5): (np / 64)^2 workgroups are 81 at 500 brokers -- a third of
// the compute units, each walking all np midpoints behind two barriers per 16 (0.11 ms a squaring, 37 % of the GPU time of a 3-s solve of
// 500 x 5000).  Here blockIdx.z takes a slice of the midpoints and the composite keys meet by atomicMin (the minimum does not depend on the
// order: the same winner, the same matrices, bit for bit); k_cx_square_fin decodes them.
__global__ __launch_bounds__(256) void k_cx_square_part(int np, int kchunk, const int32_t *__restrict__ D, uint32_t *__restrict__ keys) {
    __shared__ int32_t sa[64][17];
    __shared__ int32_t sb[16][64];
    const int i0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
    const int kbeg = blockIdx.z * kchunk, kend = min(np, kbeg + kchunk);
    const int tr = (threadIdx.x >> 4) * 4, tc = (threadIdx.x & 15) * 4;
    uint32_t best[4][4];
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) best[a][b] = 0xFFFFFFFFu;
    for (int k0 = kbeg; k0 < kend; k0 += 16) {
        for (int e = threadIdx.x; e < 64 * 16; e += 256) {
            const int r = e >> 4, kk = e & 15;
            sa[r][kk] = D[(size_t)(i0 + r) * np + k0 + kk];
        }
        for (int e = threadIdx.x; e < 16 * 64; e += 256) {
            const int kk = e >> 6, c = e & 63;
            sb[kk][c] = D[(size_t)(k0 + kk) * np + j0 + c];
        }
        __syncthreads();
#pragma unroll 4
        for (int kk = 0; kk < 16; ++kk) {
            const int k = k0 + kk;
            int bv[4];
            for (int b = 0; b < 4; ++b) bv[b] = sb[kk][tc + b];
            for (int a = 0; a < 4; ++a) {
                const int av = sa[tr + a][kk] + (1 << 18);
                const uint32_t prio = (k == i0 + tr + a) ? 0u : (uint32_t)(k + 1);
                for (int b = 0; b < 4; ++b) {
                    const uint32_t key = ((uint32_t)(av + bv[b]) << 12) | prio;
                    best[a][b] = key < best[a][b] ? key : best[a][b];
                }
            }
        }
        __syncthreads();
    }
    for (int a = 0; a < 4; ++a)
        for (int b = 0; b < 4; ++b) atomicMin(&keys[(size_t)(i0 + tr + a) * np + j0 + tc + b], best[a][b]);
}
__global__ void k_cx_square_fin(int np, const uint32_t *__restrict__ keys, int32_t *__restrict__ Dn, uint16_t *__restrict__ mid) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (size_t)np * np) return;
    const uint32_t key = keys[e];
    const int sum = (int)(key >> 12) - (1 << 18);
    const uint32_t prio = key & 0xFFFu;
    Dn[e] = sum < kCxInf ? sum : kCxInf;
    mid[e] = (uint16_t)(prio == 0 ? (int)(e / (size_t)np) : (int)prio - 1);
}

// ---- seeds: one wavefront per partition -------------------------------------------------------------------------------
// table[p][cfg] = (total, y) of the best completion of configuration cfg (total <= 0: none).  Configuration numbering:
//   [0, RF-1)                                      role swap with follower slot cfg + 1
//   RF-1 + rmi*RF + li                             slot rmi replaced by y; leader = base[li] (li < RF-1) or y (li = RF-1)
//   RF-1 + RF*RF + ((pair*rfc + ii)*RF + li)       slots of `pair` replaced by cur[p][ii] and y; base = kept + cur[p][ii]
__device__ __forceinline__ long long cx_wave_max(long long v) {
    for (int off = 32; off > 0; off >>= 1) {
        const long long o = __shfl_xor(v, off, 64);
        v = o > v ? o : v;
    }
    return v;
}

template <int MR>
__global__ __launch_bounds__(256) void k_cx_seeds(CxParams q, const uint16_t *A, const uint16_t *cur, const uint8_t *rack,
                                                  const int32_t *DF, const int32_t *DS, const int32_t *DL, int2 *table) {
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (p >= q.P) return;
    const int RF = q.RF;
    int row[MR];
    for (int k = 0; k < RF; ++k) row[k] = A[(size_t)p * RF + k];
    const uint16_t *c = cur + (size_t)p * q.rfc;
    int w0 = cx_wt(q, c, row[0], 0);
    for (int k = 1; k < RF; ++k) w0 += cx_wt(q, c, row[k], 1);
    int2 *out = table + (size_t)p * q.ncfg;
    if (lane > 0 && lane < RF) {   // role swaps
        const int k = lane;
        const int g = cx_wt(q, c, row[k], 0) + cx_wt(q, c, row[0], 1) - cx_wt(q, c, row[0], 0) - cx_wt(q, c, row[k], 1);
        const int tot = g - DS[(size_t)row[k] * q.np + row[0]];
        out[k - 1] = tot > 0 ? make_int2(tot, 0) : make_int2(0, 0);
    }
    const int n_single = RF, n_pair = RF * (RF - 1) / 2;
    for (int bc = 0; bc < n_single + n_pair * q.rfc; ++bc) {
        int base[MR], removed[2], nb = 0, nrm = 0, cfg0;
        bool valid = true;
        if (bc < n_single) {
            for (int j = 0; j < RF; ++j) { if (j == bc) removed[nrm++] = row[j]; else base[nb++] = row[j]; }
            cfg0 = (RF - 1) + bc * RF;
        } else {
            const int pi = (bc - n_single) / q.rfc, ii = (bc - n_single) % q.rfc;
            int a = 0, b = 1;   // pi-th pair in lexicographic order
            for (int t = 0; t < pi; ++t) { if (++b == RF) { ++a; b = a + 1; } }
            for (int j = 0; j < RF; ++j) { if (j == a || j == b) removed[nrm++] = row[j]; else base[nb++] = row[j]; }
            const int i = c[ii];
            valid = i < q.B;
            for (int j = 0; j < RF; ++j) valid = valid && row[j] != i;
            base[nb++] = valid ? i : 0;
            cfg0 = (RF - 1) + RF * RF + (pi * q.rfc + ii) * RF;
        }
        const CxBase<MR> cb = cx_base<MR>(q, rack, base, nb);
        valid = valid && !cb.over && cb.ndef <= 1;
        if (!valid) {
            if (lane < RF) out[cfg0 + lane] = make_int2(0, 0);
            continue;
        }
        int wl_base[MR], wf_sum = 0, cl_base[MR];
        for (int i = 0; i < nb; ++i) {
            wl_base[i] = cx_wt(q, c, base[i], 0);
            wf_sum += cx_wt(q, c, base[i], 1);
            cl_base[i] = DS[(size_t)base[i] * q.np + row[0]];
        }
        long long best[MR];
        for (int li = 0; li < RF; ++li) best[li] = -1;
        for (int y = lane; y < q.B; y += 64) {
            bool ok = cx_completes(q, cb, rack[y]);
            for (int j = 0; j < RF; ++j) ok = ok && row[j] != y;
            for (int i = 0; i < nb; ++i) ok = ok && base[i] != y;
            if (!ok) continue;
            int cR;
            if (nrm == 1) cR = DF[(size_t)y * q.np + removed[0]];
            else {
                const int i = base[nb - 1];
                const int m0 = DF[(size_t)i * q.np + removed[0]] + DF[(size_t)y * q.np + removed[1]];
                const int m1 = DF[(size_t)i * q.np + removed[1]] + DF[(size_t)y * q.np + removed[0]];
                cR = m0 < m1 ? m0 : m1;
            }
            const int wfy = cx_wt(q, c, y, 1), wly = cx_wt(q, c, y, 0);
            // one-replica seeds may also close their replica imbalance through L (option 1): free when y replaces the leader as
            // leader, plus a swap path r -> y when the leader stays
            const int cRL = nrm == 1 ? DL[(size_t)y * q.np + removed[0]] : 0;
            for (int li = 0; li < RF; ++li) {
                int tot, opt = 0;
                if (li < RF - 1) {
                    const int gain = wl_base[li] + wf_sum - cx_wt(q, c, base[li], 1) + wfy - w0;
                    tot = gain - cR - cl_base[li];
                    if (nrm == 1 && base[li] == row[0]) {
                        const int alt = gain - cRL - DS[(size_t)removed[0] * q.np + y];
                        if (alt > tot) { tot = alt; opt = 1; }
                    }
                } else {
                    const int gain = wly + wf_sum - w0;
                    tot = gain - cR - DS[(size_t)y * q.np + row[0]];
                    if (nrm == 1 && removed[0] == row[0]) {
                        const int alt = gain - cRL;
                        if (alt > tot) { tot = alt; opt = 1; }
                    }
                }
                if (tot > 0) {
                    const long long key = ((long long)tot << 13) | ((long long)(4095 - y) << 1) | (long long)opt;   // larger total, then lower y
                    best[li] = key > best[li] ? key : best[li];
                }
            }
        }
        for (int li = 0; li < RF; ++li) {
            const long long m = cx_wave_max(best[li]);
            if (lane == 0) out[cfg0 + li] = m > 0 ? make_int2((int)(m >> 13), 4095 - (int)((m >> 1) & 4095) + 4096 * (int)(m & 1)) : make_int2(0, 0);
        }
    }
}

// ---- candidates = the incumbent with a few rows rewritten: built on the device from (candidate, partition, row) patches ----
__global__ __launch_bounds__(256) void k_cx_copy(const uint16_t *__restrict__ A, uint16_t *__restrict__ out, int slots) {
    const size_t base = (size_t)blockIdx.y * (size_t)slots;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < slots; i += gridDim.x * 256) out[base + i] = A[i];
}
__global__ __launch_bounds__(256) void k_cx_patch(const int32_t *__restrict__ pq, const uint16_t *__restrict__ prow, int n_patches, int RF, int slots,
                                                  uint16_t *__restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_patches * RF) return;
    const int pi = i / RF, k = i - pi * RF;
    out[(size_t)pq[2 * pi] * (size_t)slots + (size_t)pq[2 * pi + 1] * RF + k] = prow[i];
}

#define CX_TRY(expr)                                                                                        \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess) return api_fail(KAO_ERR_HIP, (std::string(#expr) + ": " + hipGetErrorString(e_)).c_str()); \
    } while (0)

int cx_ncfg(int rf, int rfc) { return (rf - 1) + rf * rf + (rf * (rf - 1) / 2) * rfc * rf; }

// a realisation = the partitions it rewrites (each at most once) and their new rows
struct CxReal { std::vector<int> used; std::vector<uint16_t> rows; };

// Device buffers and host images of one topic's KAO-CX state
struct Cx {
    const kao_topic *t = nullptr;
    CxParams q{};
    hipStream_t stream = nullptr;
    kao_eval_plan *plan = nullptr;
    uint16_t *d_A = nullptr, *d_cur = nullptr; uint8_t *d_rack = nullptr;
    int32_t *d_cnt = nullptr;                       // c[B] | l[B]
    int32_t *d_bw = nullptr;                        // broker weights bw[B] | bwl[B] (topics that carry them)
    unsigned long long *d_E[kCxLayers] = {};
    int32_t *d_D[kCxLayers][kCxLevels + 1] = {};
    int32_t *d_DFt = nullptr;                      // transpose of the level-3 F closure (k_cx_edges_l)
    uint32_t *d_sqkey = nullptr;                   // composite keys of a squaring split over the midpoints (k_cx_square_part)
    uint16_t *d_M[kCxLayers][kCxLevels + 1] = {};
    int2 *d_table = nullptr;
    uint16_t *d_cand = nullptr; int32_t *d_obj = nullptr, *d_viol = nullptr;
    int32_t *d_pq = nullptr; uint16_t *d_prow = nullptr; size_t patch_cap = 0;   // patches of the candidates being built
    // host images of the current round
    std::vector<uint16_t> A;
    std::vector<unsigned long long> hE[kCxLayers];
    std::vector<int32_t> hD3[kCxLayers];
    std::vector<uint16_t> hM[kCxLayers][kCxLevels + 1];
    std::vector<int32_t> diag;
    std::vector<int2> table;
    bool have_paths = false;

    ~Cx() {
        (void)hipFree(d_A); (void)hipFree(d_cur); (void)hipFree(d_rack); (void)hipFree(d_cnt); (void)hipFree(d_bw); (void)hipFree(d_table);
        (void)hipFree(d_DFt); (void)hipFree(d_cand); (void)hipFree(d_obj); (void)hipFree(d_viol); (void)hipFree(d_pq); (void)hipFree(d_prow);
        for (int l = 0; l < kCxLayers; ++l) {
            (void)hipFree(d_E[l]);
            for (int v = 0; v <= kCxLevels; ++v) { (void)hipFree(d_D[l][v]); (void)hipFree(d_M[l][v]); }
        }
        if (d_sqkey) { (void)hipFree(d_sqkey); d_sqkey = nullptr; }
        if (stream) (void)hipStreamDestroy(stream);
        kao_eval_plan_destroy(plan);
    }

    int open(const kao_topic *topic) {
        t = topic;
        int32_t bd[8];
        int rc = kao_derive_bounds(t, bd);
        if (rc) return rc;
        if (t->rf > kCxMaxRF || t->rf_cur > 8 || t->n_brokers + t->n_racks + 1 > 2048 || t->rf < 2)
            return api_fail(KAO_ERR_UNSUPPORTED, "KAO-CX: needs 2 <= RF <= 8 and brokers + racks <= 2047");
        q.B = t->n_brokers; q.R = t->n_racks; q.P = t->n_partitions; q.RF = t->rf; q.rfc = t->rf_cur;
        q.n = q.B + q.R + 1; q.np = (q.n + 63) & ~63;
        q.rep_lo = bd[0]; q.rep_hi = bd[1]; q.lead_lo = bd[2]; q.lead_hi = bd[3]; q.rack_lo = bd[4]; q.rack_hi = bd[5]; q.prack_lo = bd[6]; q.prack_hi = bd[7];
        q.w00 = t->w[0][0]; q.w01 = t->w[0][1]; q.w10 = t->w[1][0]; q.w11 = t->w[1][1];
        q.ncfg = cx_ncfg(q.RF, q.rfc);
        if ((rc = api_require_init())) return rc;
        if ((rc = kao_eval_plan_create(t, &plan))) return rc;
        CX_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        const size_t nn = (size_t)q.np * q.np, slots = (size_t)q.P * q.RF;
        CX_TRY(hipMalloc(reinterpret_cast<void **>(&d_A), slots * 2));
        CX_TRY(hipMalloc(reinterpret_cast<void **>(&d_cur), (size_t)q.P * q.rfc * 2));
        CX_TRY(hipMalloc(reinterpret_cast<void **>(&d_rack), (size_t)q.B));
        CX_TRY(hipMalloc(reinterpret_cast<void **>(&d_cnt), ((size_t)q.B * 2 + (size_t)q.R) * 4));
        CX_TRY(hipMalloc(reinterpret_cast<void **>(&d_table), (size_t)q.P * q.ncfg * sizeof(int2)));
        CX_TRY(hipMalloc(reinterpret_cast<void **>(&d_DFt), nn * 4));
        for (int l = 0; l < kCxLayers; ++l) {
            CX_TRY(hipMalloc(reinterpret_cast<void **>(&d_E[l]), nn * 8));
            for (int v = 0; v <= kCxLevels; ++v) {
                CX_TRY(hipMalloc(reinterpret_cast<void **>(&d_D[l][v]), nn * 4));
                if (v) CX_TRY(hipMalloc(reinterpret_cast<void **>(&d_M[l][v]), nn * 2));
            }
        }
        CX_TRY(hipMemcpy(d_cur, t->current, (size_t)q.P * q.rfc * 2, hipMemcpyHostToDevice));
        CX_TRY(hipMemcpy(d_rack, t->rack_of, (size_t)q.B, hipMemcpyHostToDevice));
        if (t->broker_w || t->broker_wl) {   // both arrays in one allocation: bw[B] | bwl[B]
            CX_TRY(hipMalloc(reinterpret_cast<void **>(&d_bw), (size_t)q.B * 8));
            CX_TRY(hipMemset(d_bw, 0, (size_t)q.B * 8));
            if (t->broker_w) { CX_TRY(hipMemcpy(d_bw, t->broker_w, (size_t)q.B * 4, hipMemcpyHostToDevice)); q.bw = d_bw; }
            if (t->broker_wl) { CX_TRY(hipMemcpy(d_bw + q.B, t->broker_wl, (size_t)q.B * 4, hipMemcpyHostToDevice)); q.bwl = d_bw + q.B; }
        }
        return KAO_OK;
    }

    // edges, closures (device); diagonals to the host
    // one min-plus squaring; small matrices are split over the midpoints so that the launch fills the chip (KAO_CX_SPLITK=0: never)
    int square(const int32_t *Din, int32_t *Dout, uint16_t *Mout) {
        const int nt = q.np / 64, tiles = nt * nt;
        static const bool split_on = [] { const char *e = std::getenv("KAO_CX_SPLITK"); return !(e && e[0] == '0'); }();
        int ks = split_on && tiles < 200 ? std::min(8, std::max(2, (512 + tiles - 1) / tiles)) : 1;
        const int kchunk = ((q.np + ks - 1) / ks + 15) / 16 * 16;
        ks = (q.np + kchunk - 1) / kchunk;
        if (ks <= 1) { hipLaunchKernelGGL(k_cx_square, dim3(nt, nt), dim3(256), 0, stream, q.np, Din, Dout, Mout); return KAO_OK; }
        const size_t nn = (size_t)q.np * q.np;
        if (!d_sqkey) CX_TRY(hipMalloc(reinterpret_cast<void **>(&d_sqkey), nn * 4));
        CX_TRY(hipMemsetAsync(d_sqkey, 0xFF, nn * 4, stream));
        hipLaunchKernelGGL(k_cx_square_part, dim3(nt, nt, ks), dim3(256), 0, stream, q.np, kchunk, Din, d_sqkey);
        hipLaunchKernelGGL(k_cx_square_fin, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, stream, q.np, d_sqkey, Dout, Mout);
        return KAO_OK;
    }
    // lazy (round 5; plain rounds only -- a bulk round takes the cycles of EVERY level): a layer is squared only up to its first
    // level with a negative diagonal entry, the one cx_round takes its cycles from (drifted 500 x 5000, round 4: 5,814 squarings for 646
    // builds, 17 % of a solve's GPU time, nine per build whether level 1 had found something or not); the matrices of the levels above
    // stay stale and are not read (the L layer prices its compensations on the F closure only when F has NO cycle, i.e. when every F
    // level was computed; the seeds are looked at only when no layer has a cycle).  Same candidates as the eager build.
    int build(const uint16_t *assign, bool lazy = false) {
        const size_t nn = (size_t)q.np * q.np, slots = (size_t)q.P * q.RF;
        A.assign(assign, assign + slots);
        std::vector<int32_t> cnt((size_t)q.B * 2 + (size_t)q.R, 0);   // c[B] | l[B] | K[R]
        for (int p = 0; p < q.P; ++p)
            for (int k = 0; k < q.RF; ++k) {
                const unsigned b = A[(size_t)p * q.RF + k];
                if (b >= (unsigned)q.B) return api_fail(KAO_ERR_INVALID, "KAO-CX: the assignment has an empty or out-of-range slot");
                ++cnt[b];
                ++cnt[(size_t)q.B * 2 + t->rack_of[b]];
                if (k == 0) ++cnt[(size_t)q.B + b];
            }
        CX_TRY(hipMemcpyAsync(d_A, A.data(), slots * 2, hipMemcpyHostToDevice, stream));
        CX_TRY(hipMemcpyAsync(d_cnt, cnt.data(), cnt.size() * 4, hipMemcpyHostToDevice, stream));
        for (int l = 0; l < kCxLayers; ++l) CX_TRY(hipMemsetAsync(d_E[l], 0xFF, nn * 8, stream));
        if (q.RF <= 4) hipLaunchKernelGGL(k_cx_edges<4>, dim3((q.P + 3) / 4), dim3(256), 0, stream, q, d_A, d_cur, d_rack, d_E[0], d_E[1]);
        else hipLaunchKernelGGL(k_cx_edges<8>, dim3((q.P + 3) / 4), dim3(256), 0, stream, q, d_A, d_cur, d_rack, d_E[0], d_E[1]);
        const dim3 g0((q.np + 255) / 256, q.np);
        hipLaunchKernelGGL(k_cx_dist0, g0, dim3(256), 0, stream, q, d_E[0], d_cnt, d_rack, 0, 32, d_D[0][0]);
        hipLaunchKernelGGL(k_cx_dist0, g0, dim3(256), 0, stream, q, d_E[1], d_cnt, d_rack, 1, 32, d_D[1][0]);
        const dim3 gs(q.np / 64, q.np / 64);
        diag.assign((size_t)kCxLayers * kCxLevels * q.B, 0);
        auto fetch_level = [&](int l, int v) -> int {
            CX_TRY(hipMemcpy2DAsync(&diag[((size_t)l * kCxLevels + (v - 1)) * q.B], 4, d_D[l][v], ((size_t)q.np + 1) * 4, 4, (size_t)q.B,
                                    hipMemcpyDeviceToHost, stream));
            return KAO_OK;
        };
        auto level_neg = [&](int l, int v) { const int32_t *dg = &diag[((size_t)l * kCxLevels + (v - 1)) * q.B]; for (int b = 0; b < q.B; ++b) if (dg[b] < 0) return true; return false; };
        int rc;
        if (!lazy) {
            for (int v = 1; v <= kCxLevels; ++v)
                for (int l = 0; l < 2; ++l)
                    if ((rc = square(d_D[l][v - 1], d_D[l][v], d_M[l][v]))) return rc;
            CX_TRY(hipGetLastError());
            for (int l = 0; l < 2; ++l) for (int v = 1; v <= kCxLevels; ++v) if ((rc = fetch_level(l, v))) return rc;
            CX_TRY(hipStreamSynchronize(stream));
        } else {
            bool found[2] = {false, false};
            for (int v = 1; v <= kCxLevels && !(found[0] && found[1]); ++v) {
                for (int l = 0; l < 2; ++l) {
                    if (found[l]) continue;
                    if ((rc = square(d_D[l][v - 1], d_D[l][v], d_M[l][v]))) return rc;
                    if ((rc = fetch_level(l, v))) return rc;
                }
                CX_TRY(hipGetLastError());
                CX_TRY(hipStreamSynchronize(stream));
                for (int l = 0; l < 2; ++l) if (!found[l]) found[l] = level_neg(l, v);
            }
        }
        // the L graph prices its compensations on the F closure: only when F has no improving cycle of its own
        int f_neg = 0;
        for (size_t i = 0; i < (size_t)kCxLevels * q.B; ++i) f_neg |= diag[i] < 0;
        if (!f_neg) hipLaunchKernelGGL(k_cx_transpose, dim3(q.np / 64, q.np / 64), dim3(256), 0, stream, q.np, d_D[0][kCxLevels], d_DFt);   // (plain_only reads no compensation)
        {
            const dim3 gl((q.P + 3) / 4), bl(256);
            if (q.RF == 2) hipLaunchKernelGGL((k_cx_edges_l<4, 2>), gl, bl, 0, stream, q, d_A, d_cur, d_rack, d_D[0][kCxLevels], d_DFt, f_neg, d_E[2]);
            else if (q.RF == 3) hipLaunchKernelGGL((k_cx_edges_l<4, 3>), gl, bl, 0, stream, q, d_A, d_cur, d_rack, d_D[0][kCxLevels], d_DFt, f_neg, d_E[2]);
            else if (q.RF == 4) hipLaunchKernelGGL((k_cx_edges_l<4, 4>), gl, bl, 0, stream, q, d_A, d_cur, d_rack, d_D[0][kCxLevels], d_DFt, f_neg, d_E[2]);
            else hipLaunchKernelGGL((k_cx_edges_l<8, 0>), gl, bl, 0, stream, q, d_A, d_cur, d_rack, d_D[0][kCxLevels], d_DFt, f_neg, d_E[2]);
        }
        hipLaunchKernelGGL(k_cx_dist0, g0, dim3(256), 0, stream, q, d_E[2], d_cnt, d_rack, 2, 44, d_D[2][0]);   // L: no slack edges
        for (int v = 1; v <= kCxLevels; ++v) {
            if ((rc = square(d_D[2][v - 1], d_D[2][v], d_M[2][v]))) return rc;
            if ((rc = fetch_level(2, v))) return rc;
            if (lazy) { CX_TRY(hipStreamSynchronize(stream)); if (level_neg(2, v)) break; }
        }
        CX_TRY(hipGetLastError());
        CX_TRY(hipStreamSynchronize(stream));
        have_paths = false;
        return KAO_OK;
    }

    int eval_buffers() {
        if (d_cand) return KAO_OK;
        const size_t slots = (size_t)q.P * q.RF;
        CX_TRY(hipMalloc(reinterpret_cast<void **>(&d_cand), (size_t)(kCxMaxEval + 1) * slots * 2));
        CX_TRY(hipMalloc(reinterpret_cast<void **>(&d_obj), (size_t)(kCxMaxEval + 1) * 4));
        CX_TRY(hipMalloc(reinterpret_cast<void **>(&d_viol), (size_t)(kCxMaxEval + 1) * 32));
        return KAO_OK;
    }

    // exact scores by K-eval (at most kCxMaxEval + 1 assignments)
    int eval(const std::vector<const uint16_t *> &xs, std::vector<int32_t> &obj, std::vector<int32_t> &viol) {
        int rc = eval_buffers();
        if (rc) return rc;
        const size_t slots = (size_t)q.P * q.RF;
        for (size_t i = 0; i < xs.size(); ++i)
            CX_TRY(hipMemcpyAsync(d_cand + i * slots, xs[i], slots * 2, hipMemcpyHostToDevice, stream));
        CX_TRY(hipStreamSynchronize(stream));
        rc = kao_eval_plan_run(plan, d_cand, (int64_t)xs.size(), d_obj, d_viol, nullptr);
        if (!rc) rc = kao_eval_plan_sync(plan, nullptr);
        if (rc) return rc;
        obj.resize(xs.size()); viol.resize(xs.size() * 8);
        CX_TRY(hipMemcpy(obj.data(), d_obj, xs.size() * 4, hipMemcpyDeviceToHost));
        CX_TRY(hipMemcpy(viol.data(), d_viol, xs.size() * 32, hipMemcpyDeviceToHost));
        return KAO_OK;
    }

    int seeds() {
        if (q.RF <= 4) hipLaunchKernelGGL(k_cx_seeds<4>, dim3((q.P + 3) / 4), dim3(256), 0, stream, q, d_A, d_cur, d_rack, d_D[0][kCxLevels], d_D[1][kCxLevels], d_D[2][kCxLevels], d_table);
        else hipLaunchKernelGGL(k_cx_seeds<8>, dim3((q.P + 3) / 4), dim3(256), 0, stream, q, d_A, d_cur, d_rack, d_D[0][kCxLevels], d_D[1][kCxLevels], d_D[2][kCxLevels], d_table);
        CX_TRY(hipGetLastError());
        table.resize((size_t)q.P * q.ncfg);
        CX_TRY(hipMemcpyAsync(table.data(), d_table, table.size() * sizeof(int2), hipMemcpyDeviceToHost, stream));
        CX_TRY(hipStreamSynchronize(stream));
        return KAO_OK;
    }

    int fetch_paths() {
        if (have_paths) return KAO_OK;
        const size_t nn = (size_t)q.np * q.np;
        for (int l = 0; l < kCxLayers; ++l) {
            hE[l].resize(nn); hD3[l].resize(nn);
            CX_TRY(hipMemcpyAsync(hE[l].data(), d_E[l], nn * 8, hipMemcpyDeviceToHost, stream));
            CX_TRY(hipMemcpyAsync(hD3[l].data(), d_D[l][kCxLevels], nn * 4, hipMemcpyDeviceToHost, stream));
            for (int v = 1; v <= kCxLevels; ++v) {
                hM[l][v].resize(nn);
                CX_TRY(hipMemcpyAsync(hM[l][v].data(), d_M[l][v], nn * 2, hipMemcpyDeviceToHost, stream));
            }
        }
        CX_TRY(hipStreamSynchronize(stream));
        have_paths = true;
        return KAO_OK;
    }

    // ---- realisation (host; mirrors oracle/kao_cycle.py Round._path / _walk / realise_*) ----
    void path(int layer, int u, int v, int lev, std::vector<int> &out) const {   // appends the nodes after u
        if (u == v) return;
        if (lev == 0) { out.push_back(v); return; }
        const int m = hM[layer][lev][(size_t)u * q.np + v];
        path(layer, u, m, lev - 1, out);
        path(layer, m, v, lev - 1, out);
    }
    bool walk(CxReal &r, int layer, int from, const std::vector<int> &nodes) const {
        int s = from;
        for (int d : nodes) {
            const int s0 = s;
            s = d;
            if (s0 == d || s0 >= q.B || d >= q.B) continue;   // slack nodes carry no slot
            const unsigned long long key = hE[layer][(size_t)s0 * q.np + d];
            if (key == kNoEdge) return false;
            if (layer == 2) {   // generalised leader transfer: the partition's row changes, then the compensating F path
                const unsigned long long pay = key & ((1ull << 44) - 1);
                const int qq = (int)(pay >> 20), var = (int)((pay >> 16) & 15), k = (int)((pay >> 12) & 15), y = (int)(pay & 4095);
                if (std::find(r.used.begin(), r.used.end(), qq) != r.used.end()) return false;
                r.used.push_back(qq);
                const size_t o = r.rows.size();
                const uint16_t *row = &A[(size_t)qq * q.RF];
                r.rows.insert(r.rows.end(), row, row + q.RF);
                const int u = row[0];
                int c0 = -1, c1 = -1;
                if (var == 0) r.rows[o] = (uint16_t)d;
                else if (var == 1) { c0 = u; c1 = row[k]; r.rows[o] = (uint16_t)d; r.rows[o + (size_t)k] = (uint16_t)u; }
                else if (var == 2) { c0 = y; c1 = d; r.rows[o] = (uint16_t)d; r.rows[o + (size_t)k] = (uint16_t)y; }
                else if (var == 3) { c0 = u; c1 = d; r.rows[o] = row[k]; r.rows[o + (size_t)k] = (uint16_t)u; }
                else { c0 = y; c1 = row[k]; r.rows[o] = (uint16_t)d; r.rows[o + (size_t)k] = (uint16_t)y; }
                if (c0 >= 0 && c0 != c1) {
                    std::vector<int> comp;
                    path(0, c0, c1, kCxLevels, comp);
                    if (!walk(r, 0, c0, comp)) return false;
                }
                continue;
            }
            const unsigned slot = (unsigned)(key & 0xFFFFFFFFu);
            const int qq = (int)(slot / (unsigned)q.RF), j = (int)(slot % (unsigned)q.RF);
            if (std::find(r.used.begin(), r.used.end(), qq) != r.used.end()) return false;
            r.used.push_back(qq);
            const size_t o = r.rows.size();
            r.rows.insert(r.rows.end(), &A[(size_t)qq * q.RF], &A[(size_t)qq * q.RF] + q.RF);
            if (layer == 1) std::swap(r.rows[o], r.rows[o + (size_t)j]);
            else r.rows[o + (size_t)j] = (uint16_t)d;
        }
        return true;
    }
    // candidates on the device: `count` copies of A with the patches of reals[first + i] applied, then K-eval
    int eval_patched(const std::vector<const CxReal *> &rs, std::vector<int32_t> &obj, std::vector<int32_t> &viol) {
        int rc = eval_buffers();
        if (rc) return rc;
        const size_t n = rs.size();
        std::vector<int32_t> pq;            // (candidate, partition) per patch
        std::vector<uint16_t> prow;
        for (size_t i = 0; i < n; ++i)
            for (size_t u = 0; u < rs[i]->used.size(); ++u) {
                pq.push_back((int32_t)i); pq.push_back(rs[i]->used[u]);
                prow.insert(prow.end(), &rs[i]->rows[u * q.RF], &rs[i]->rows[u * q.RF] + q.RF);
            }
        const size_t np_ = pq.size() / 2;
        if (np_ > patch_cap) {
            (void)hipFree(d_pq); (void)hipFree(d_prow); d_pq = nullptr; d_prow = nullptr;
            patch_cap = std::max<size_t>(2 * np_, 4096);
            CX_TRY(hipMalloc(reinterpret_cast<void **>(&d_pq), patch_cap * 8));
            CX_TRY(hipMalloc(reinterpret_cast<void **>(&d_prow), patch_cap * kCxMaxRF * 2));
        }
        CX_TRY(hipMemcpyAsync(d_pq, pq.data(), pq.size() * 4, hipMemcpyHostToDevice, stream));
        CX_TRY(hipMemcpyAsync(d_prow, prow.data(), prow.size() * 2, hipMemcpyHostToDevice, stream));
        launch_apply(n, np_);
        CX_TRY(hipGetLastError());
        CX_TRY(hipStreamSynchronize(stream));
        rc = kao_eval_plan_run(plan, d_cand, (int64_t)n, d_obj, d_viol, nullptr);
        if (!rc) rc = kao_eval_plan_sync(plan, nullptr);
        if (rc) return rc;
        obj.resize(n); viol.resize(n * 8);
        CX_TRY(hipMemcpy(obj.data(), d_obj, n * 4, hipMemcpyDeviceToHost));
        CX_TRY(hipMemcpy(viol.data(), d_viol, n * 32, hipMemcpyDeviceToHost));
        return KAO_OK;
    }
    void launch_apply(size_t n, size_t n_patches);
    void seed_row(int p, int cfg, int y, int *nr) const {
        const int RF = q.RF;
        const uint16_t *row = &A[(size_t)p * RF];
        if (cfg < RF - 1) { for (int k = 0; k < RF; ++k) nr[k] = row[k]; std::swap(nr[0], nr[cfg + 1]); return; }
        cfg -= RF - 1;
        int full[kCxMaxRF], nf = 0, li;
        if (cfg < RF * RF) {
            const int rmi = cfg / RF; li = cfg % RF;
            for (int j = 0; j < RF; ++j) if (j != rmi) full[nf++] = row[j];
        } else {
            cfg -= RF * RF;
            li = cfg % RF;
            const int qd = cfg / RF, pi = qd / q.rfc, ii = qd % q.rfc;
            int a = 0, b = 1;
            for (int tt = 0; tt < pi; ++tt) { if (++b == RF) { ++a; b = a + 1; } }
            for (int j = 0; j < RF; ++j) if (j != a && j != b) full[nf++] = row[j];
            full[nf++] = t->current[(size_t)p * q.rfc + ii];
        }
        full[nf++] = y;
        nr[0] = full[li];
        int o = 1;
        for (int i = 0; i < nf; ++i) if (i != li) nr[o++] = full[i];
    }
};

void Cx::launch_apply(size_t n, size_t n_patches) {
    const int slots = q.P * q.RF;
    hipLaunchKernelGGL(k_cx_copy, dim3((unsigned)std::min(64, (slots + 255) / 256), (unsigned)n), dim3(256), 0, stream, d_A, d_cand, slots);
    if (n_patches)
        hipLaunchKernelGGL(k_cx_patch, dim3((unsigned)((n_patches * q.RF + 255) / 256)), dim3(256), 0, stream, d_pq, d_prow, (int)n_patches, q.RF, slots, d_cand);
}

struct CxCand { int total, a, b, c; };   // seed: (total, p, cfg, y); cycle: (gain, layer, level, broker)


// one round from `assign` (feasible, objective `base`): returns 1 and overwrites assign when it improved, 0 when nothing was found,
// a negative KAO_ERR_* code on failure
int cx_round(Cx &cx, uint16_t *assign, int32_t base, int32_t *new_obj, int32_t stats[8]) {
    const CxParams &q = cx.q;
    static const bool trace = std::getenv("KAO_CX_TRACE") != nullptr;
    const double tt0 = api_now_s();
    const char *bulk_env = std::getenv("KAO_CX_BULK_SLOTS");   // test hook (read every round: tests switch it)
    const int64_t bulk_slots = bulk_env && *bulk_env ? (int64_t)std::atoll(bulk_env) : kCxBulkSlots;
    const bool bulk_topic = (int64_t)q.P * q.RF > bulk_slots;
    static const bool lazy_on = [] { const char *e = std::getenv("KAO_CX_LAZY"); return !(e && e[0] == '0'); }();   // KAO_CX_LAZY=0: every level in every build
    int rc = cx.build(assign, lazy_on && !bulk_topic);
    if (rc) return rc;
    const double tt1 = api_now_s();
    // ---- candidates ----
    auto collect = [&](bool all_levels) {   // the lowest level of every layer that has a negative diagonal entry; every such level in bulk mode
        std::vector<CxCand> cyc;
        for (int l = 0; l < kCxLayers; ++l)
            for (int v = 1; v <= kCxLevels; ++v) {
                const int32_t *dg = &cx.diag[((size_t)l * kCxLevels + (v - 1)) * q.B];
                bool any = false;
                for (int b = 0; b < q.B; ++b) if (dg[b] < 0) { cyc.push_back({-dg[b], l, v, b}); any = true; }
                if (any && !all_levels) break;
            }
        std::sort(cyc.begin(), cyc.end(), [](const CxCand &x, const CxCand &y) {
            return x.total != y.total ? x.total > y.total : (x.a != y.a ? x.a < y.a : (x.b != y.b ? x.b < y.b : x.c < y.c)); });
        return cyc;
    };
    std::vector<CxCand> cyc = collect(bulk_topic);   // (every level on small topics too: hard family 12 / 12 / 12 of 14 against 12 / 11 / 13 -- no gain, GPU call 18)
    std::vector<CxCand> cands;
    const bool cycles = !cyc.empty();
    if (cycles) {
        cands = cyc;
    } else {
        if ((rc = cx.seeds())) return rc;
        for (int p = 0; p < q.P; ++p)
            for (int c = 0; c < q.ncfg; ++c) {
                const int2 e = cx.table[(size_t)p * q.ncfg + c];
                if (e.x > 0) cands.push_back({e.x, p, c, e.y});
            }
        std::sort(cands.begin(), cands.end(), [](const CxCand &x, const CxCand &y) {
            return x.total != y.total ? x.total > y.total : (x.a != y.a ? x.a < y.a : x.b < y.b); });
    }
    stats[4] += (int32_t)cands.size();
    if (cands.empty()) return 0;
    const double tt2 = api_now_s();
    if ((rc = cx.fetch_paths())) return rc;
    const double tt3 = api_now_s();
    // ---- realisations ----
    const size_t slots = (size_t)q.P * q.RF;
    std::vector<CxReal> reals;
    std::set<std::vector<int>> seen;
    auto push = [&](CxReal &&r) {
        std::vector<size_t> order(r.used.size());
        for (size_t i = 0; i < order.size(); ++i) order[i] = i;
        std::sort(order.begin(), order.end(), [&](size_t x, size_t y) { return r.used[x] < r.used[y]; });
        std::vector<int> sig;
        for (size_t i : order) { sig.push_back(r.used[i]); for (int k = 0; k < q.RF; ++k) sig.push_back(r.rows[i * q.RF + k]); }
        if (seen.insert(sig).second) reals.push_back(std::move(r));
    };
    // Bulk mode (round 4; topics beyond kCxBulkSlots replica slots, cycle candidates): every candidate is unrolled, a
    // partition-disjoint set is taken in candidate order BEFORE any scoring, and only its merges (all, half, a quarter, ... , the
    // best candidate alone) are scored exactly -- a dozen K-eval candidates instead of 512 (drifted 1000 x 100,000: 512 of 512
    // realisations improved in nearly every round, scoring them one by one was a quarter of a round).  A round whose merges
    // all fail falls through to the one-by-one path.
    const bool bulk = cycles && bulk_topic;
    const int max_real = bulk ? 8 * kCxMaxEval : kCxMaxEval;
    for (const CxCand &cd : cands) {
        if ((int)reals.size() >= max_real) break;
        if (cycles) {
            const int layer = cd.a, lev = cd.b, b = cd.c;
            const int m = cx.hM[layer][lev][(size_t)b * q.np + b];
            std::vector<int> nodes;
            cx.path(layer, b, m, lev - 1, nodes);
            cx.path(layer, m, b, lev - 1, nodes);
            CxReal r;
            if (cx.walk(r, layer, b, nodes)) push(std::move(r));
        } else {
            const int p = cd.a, opt = cd.c >> 12, y = cd.c & 4095;
            int nr[kCxMaxRF];
            cx.seed_row(p, cd.b, y, nr);
            const uint16_t *row = &cx.A[(size_t)p * q.RF];
            int Rm[kCxMaxRF] = {}, Ad[kCxMaxRF] = {}, nrm = 0, nad = 0;
            for (int k = 0; k < q.RF; ++k) { bool in = false; for (int j = 0; j < q.RF; ++j) in = in || nr[j] == row[k]; if (!in) Rm[nrm++] = row[k]; }
            for (int k = 0; k < q.RF; ++k) { bool in = false; for (int j = 0; j < q.RF; ++j) in = in || row[j] == nr[k]; if (!in) Ad[nad++] = nr[k]; }
            if (opt) {   // replica imbalance closed through L (and the leader imbalance, when the leader stayed, through S)
                CxReal r;
                for (int k = 0; k < q.RF; ++k) r.rows.push_back((uint16_t)nr[k]);
                r.used.push_back(p);
                std::vector<int> nodes;
                cx.path(2, Ad[0], Rm[0], kCxLevels, nodes);
                bool good = cx.walk(r, 2, Ad[0], nodes);
                if (good && nr[0] == row[0]) {
                    nodes.clear();
                    cx.path(1, Rm[0], Ad[0], kCxLevels, nodes);
                    good = cx.walk(r, 1, Rm[0], nodes);
                }
                if (good) push(std::move(r));
                continue;
            }
            int orders[2][2] = {{Rm[0], nrm > 1 ? Rm[1] : 0}, {nrm > 1 ? Rm[1] : 0, Rm[0]}};
            int n_orders = nrm == 2 ? 2 : 1;
            if (nrm == 2) {
                const std::vector<int32_t> &D3 = cx.hD3[0];
                const int m0 = D3[(size_t)Ad[0] * q.np + Rm[0]] + D3[(size_t)Ad[1] * q.np + Rm[1]];
                const int m1 = D3[(size_t)Ad[0] * q.np + Rm[1]] + D3[(size_t)Ad[1] * q.np + Rm[0]];
                if (!(m0 <= m1)) { std::swap(orders[0][0], orders[1][0]); std::swap(orders[0][1], orders[1][1]); }
            }
            for (int o = 0; o < n_orders && (int)reals.size() < kCxMaxEval; ++o) {
                CxReal r;
                for (int k = 0; k < q.RF; ++k) r.rows.push_back((uint16_t)nr[k]);
                r.used.push_back(p);
                bool good = true;
                for (int i = 0; i < nad && good; ++i) {
                    std::vector<int> nodes;
                    cx.path(0, Ad[i], orders[o][i], kCxLevels, nodes);
                    good = cx.walk(r, 0, Ad[i], nodes);
                }
                if (good && nr[0] != row[0]) {
                    std::vector<int> nodes;
                    cx.path(1, nr[0], row[0], kCxLevels, nodes);
                    good = cx.walk(r, 1, nr[0], nodes);
                }
                if (good) push(std::move(r));
            }
        }
    }
    stats[2] += (int32_t)reals.size();
    if (reals.empty()) return 0;
    const double tt4 = api_now_s();
    if (bulk) {
        std::vector<char> taken((size_t)q.P, 0);
        std::vector<int> chosen;
        for (size_t i = 0; i < reals.size(); ++i) {
            bool clash = false;
            for (int u : reals[i].used) clash = clash || taken[(size_t)u];
            if (clash) continue;
            for (int u : reals[i].used) taken[(size_t)u] = 1;
            chosen.push_back((int)i);
        }
        std::vector<int> sizes;
        for (int k = (int)chosen.size(); k >= 1; k /= 2) sizes.push_back(k);
        std::vector<CxReal> merges(sizes.size());
        std::vector<const CxReal *> ms(sizes.size());
        for (size_t m = 0; m < sizes.size(); ++m) {
            for (int c = 0; c < sizes[m]; ++c) {
                const CxReal &r = reals[(size_t)chosen[(size_t)c]];
                merges[m].used.insert(merges[m].used.end(), r.used.begin(), r.used.end());
                merges[m].rows.insert(merges[m].rows.end(), r.rows.begin(), r.rows.end());
            }
            ms[m] = &merges[m];
        }
        std::vector<int32_t> o1, v1;
        if ((rc = cx.eval_patched(ms, o1, v1))) return rc;
        int win = -1;
        for (size_t m = 0; m < sizes.size(); ++m)
            if (v1[m * 8] == 0 && o1[m] > base && (win < 0 || o1[m] > o1[(size_t)win])) win = (int)m;
        if (win >= 0) {
            const CxReal &winner = merges[(size_t)win];
            std::memcpy(assign, cx.A.data(), slots * 2);
            for (size_t u = 0; u < winner.used.size(); ++u)
                std::memcpy(&assign[(size_t)winner.used[u] * q.RF], &winner.rows[u * q.RF], (size_t)q.RF * 2);
            *new_obj = o1[(size_t)win];
            stats[3] += sizes[(size_t)win];
            if (sizes[(size_t)win] > 1) stats[5] += sizes[(size_t)win];
            if (trace)
                std::fprintf(stderr, "[kao-cx] cycles (bulk) build %.2f ms, candidates %zu in %.2f ms, paths %.2f ms, %zu realisations %.2f ms, eval+merge %.2f ms: %d -> %d (%d merged of %zu partition-disjoint)\n",
                             (tt1 - tt0) * 1e3, cands.size(), (tt2 - tt1) * 1e3, (tt3 - tt2) * 1e3, reals.size(), (tt4 - tt3) * 1e3,
                             (api_now_s() - tt4) * 1e3, base, *new_obj, sizes[(size_t)win], chosen.size());
            return 1;
        }
        if (reals.size() > (size_t)kCxMaxEval) reals.resize((size_t)kCxMaxEval);   // one by one: the best kCxMaxEval candidates
    }
    // ---- exact evaluation by K-eval ----
    const size_t n = reals.size();
    std::vector<const CxReal *> rs(n);
    for (size_t i = 0; i < n; ++i) rs[i] = &reals[i];
    std::vector<int32_t> obj, viol;
    if ((rc = cx.eval_patched(rs, obj, viol))) return rc;
    int best = -1, n_good = 0;
    std::vector<char> taken((size_t)q.P, 0);
    std::vector<int> chosen;   // partition-disjoint improving realisations, candidate order
    for (size_t i = 0; i < n; ++i) {
        if (viol[i * 8] != 0 || obj[i] <= base) continue;
        ++n_good;
        if (best < 0 || obj[i] > obj[(size_t)best]) best = (int)i;
        bool clash = false;
        for (int u : reals[i].used) clash = clash || taken[(size_t)u];
        if (clash) continue;
        for (int u : reals[i].used) taken[(size_t)u] = 1;
        chosen.push_back((int)i);
    }
    stats[3] += n_good;
    if (best < 0) return 0;
    int32_t win_obj = obj[(size_t)best];
    int n_taken = 1;
    // merges of the first m, m/2, m/4, ... chosen ones (two compounds may still clash on a band's slack): the best feasible wins
    std::vector<int> sizes;
    for (int k = (int)chosen.size(); k >= 2; k /= 2) sizes.push_back(k);
    std::vector<CxReal> merges(sizes.size());
    if (!sizes.empty()) {
        std::vector<const CxReal *> ms(sizes.size());
        for (size_t m = 0; m < sizes.size(); ++m) {
            for (int c = 0; c < sizes[m]; ++c) {
                const CxReal &r = reals[(size_t)chosen[(size_t)c]];
                merges[m].used.insert(merges[m].used.end(), r.used.begin(), r.used.end());
                merges[m].rows.insert(merges[m].rows.end(), r.rows.begin(), r.rows.end());
            }
            ms[m] = &merges[m];
        }
        std::vector<int32_t> o1, v1;
        if ((rc = cx.eval_patched(ms, o1, v1))) return rc;
        for (size_t m = 0; m < sizes.size(); ++m)
            if (v1[m * 8] == 0 && o1[m] > win_obj) { win_obj = o1[m]; n_taken = -(int)(m + 1); }
    }
    const CxReal &winner = n_taken < 0 ? merges[(size_t)(-n_taken - 1)] : reals[(size_t)best];
    if (n_taken < 0) { n_taken = sizes[(size_t)(-n_taken - 1)]; stats[5] += n_taken; }
    std::memcpy(assign, cx.A.data(), slots * 2);
    for (size_t u = 0; u < winner.used.size(); ++u)
        std::memcpy(&assign[(size_t)winner.used[u] * q.RF], &winner.rows[u * q.RF], (size_t)q.RF * 2);
    *new_obj = win_obj;
    if (trace)
        std::fprintf(stderr, "[kao-cx] %s build %.2f ms, candidates %zu in %.2f ms, paths %.2f ms, %zu realisations %.2f ms, eval+merge %.2f ms: %d -> %d (%d merged; %d improving, %zu partition-disjoint)\n",
                     cycles ? "cycles" : "seeds", (tt1 - tt0) * 1e3, cands.size(), (tt2 - tt1) * 1e3, (tt3 - tt2) * 1e3, reals.size(), (tt4 - tt3) * 1e3,
                     (api_now_s() - tt4) * 1e3, base, win_obj, n_taken, n_good, chosen.size());
    return 1;
}

// One round of the COMPOUND-EDGE layer (kao_pairs.cpp; specification oracle/kao_cycle_pairs.py::find_improvement) from a fixpoint
// of the plain layers: `assign` is the assignment cx_round has just built its graphs for and found nothing on.  The compound
// edges of leader-balanced pairs are laid over the level-0 F matrix, the closure is squared again (k_cx_square), negative diagonal
// entries are unrolled -- a compound edge applies its two rows, an F edge its slot, every partition at most once -- and the
// realisations are scored exactly by K-eval.  Returns 1 and overwrites assign when one improved, 0 when none did, a negative
// KAO_ERR_* code on failure.  OFF unless KAO_CX_PAIRS=1: the enumeration is held to the oracle on the CPU (tests/test_host.py) and
// the whole round lifts the committed 300 x 2000 fixpoint from 14825 to the MILP optimum 14826 on the device (0.37 s, GPU call 33,
// tests/test_gpu_cycle.py), but kao_solve has not been measured with it (round 3 ran out of GPU minutes): not the default yet.
int cx_pairs_round(Cx &cx, uint16_t *assign, int32_t base, int32_t *new_obj, int32_t stats[8]) {
    const CxParams &q = cx.q;
    std::unordered_map<uint32_t, PairEdge> edges;
    int64_t st[4];
    int rc = pair_edges(cx.t, cx.A.data(), -2, edges, st);
    if (rc) return rc;
    if (edges.empty()) return 0;
    const size_t nn = (size_t)q.np * q.np;
    std::vector<int32_t> D0(nn);
    CX_TRY(hipStreamSynchronize(cx.stream));
    CX_TRY(hipMemcpy(D0.data(), cx.d_D[0][0], nn * 4, hipMemcpyDeviceToHost));
    if (!cx.have_paths) {   // the F edge keys (slot behind every level-0 edge); the closures on the host go stale below
        cx.hE[0].resize(nn);
        CX_TRY(hipMemcpy(cx.hE[0].data(), cx.d_E[0], nn * 8, hipMemcpyDeviceToHost));
    }
    std::vector<uint8_t> comp(nn, 0);
    size_t n_comp = 0;
    for (const auto &kv : edges) {
        const size_t x = kv.first / (uint32_t)q.B, z = kv.first % (uint32_t)q.B, idx = x * q.np + z;
        if (kv.second.cost < D0[idx]) { D0[idx] = kv.second.cost; comp[idx] = 1; ++n_comp; }
    }
    if (!n_comp) return 0;
    cx.have_paths = false;
    CX_TRY(hipMemcpyAsync(cx.d_D[0][0], D0.data(), nn * 4, hipMemcpyHostToDevice, cx.stream));
    const dim3 gs(q.np / 64, q.np / 64);
    for (int v = 1; v <= kCxLevels; ++v)
        if ((rc = cx.square(cx.d_D[0][v - 1], cx.d_D[0][v], cx.d_M[0][v]))) return rc;
    CX_TRY(hipGetLastError());
    std::vector<int32_t> dg((size_t)kCxLevels * q.B);
    for (int v = 1; v <= kCxLevels; ++v)
        CX_TRY(hipMemcpy2DAsync(&dg[(size_t)(v - 1) * q.B], 4, cx.d_D[0][v], ((size_t)q.np + 1) * 4, 4, (size_t)q.B, hipMemcpyDeviceToHost, cx.stream));
    CX_TRY(hipStreamSynchronize(cx.stream));
    // every level with negative diagonal entries contributes its (at most 64) most negative ones: the shortest cycles first
    std::vector<CxCand> cyc;
    for (int v = 1; v <= kCxLevels; ++v) {
        std::vector<CxCand> lv;
        for (int b = 0; b < q.B; ++b)
            if (dg[(size_t)(v - 1) * q.B + b] < 0) lv.push_back({-dg[(size_t)(v - 1) * q.B + b], 0, v, b});
        std::sort(lv.begin(), lv.end(), [](const CxCand &x, const CxCand &y) { return x.total != y.total ? x.total > y.total : x.c < y.c; });
        if (lv.size() > 64) lv.resize(64);
        cyc.insert(cyc.end(), lv.begin(), lv.end());
    }
    if (cyc.empty()) return 0;
    stats[4] += (int32_t)cyc.size();
    for (int v = 1; v <= kCxLevels; ++v) {
        cx.hM[0][v].resize(nn);
        CX_TRY(hipMemcpyAsync(cx.hM[0][v].data(), cx.d_M[0][v], nn * 2, hipMemcpyDeviceToHost, cx.stream));
    }
    CX_TRY(hipStreamSynchronize(cx.stream));
    std::vector<CxReal> reals;
    for (const CxCand &cd : cyc) {
        const int lev = cd.b;
        const int b = cd.c, m = cx.hM[0][lev][(size_t)b * q.np + b];
        std::vector<int> nodes;
        cx.path(0, b, m, lev - 1, nodes);
        cx.path(0, m, b, lev - 1, nodes);
        CxReal r;
        bool good = true;
        int s = b;
        for (int d : nodes) {
            const int s0 = s;
            s = d;
            if (s0 == d || s0 >= q.B || d >= q.B) continue;   // slack nodes carry no slot
            if (comp[(size_t)s0 * q.np + d]) {
                const PairEdge &e = edges[(uint32_t)s0 * (uint32_t)q.B + (uint32_t)d];
                const int ps[2] = {e.p, e.q};
                const uint16_t *rows[2] = {e.rowp, e.rowq};
                for (int i = 0; i < 2 && good; ++i) {
                    if (std::find(r.used.begin(), r.used.end(), ps[i]) != r.used.end()) { good = false; break; }
                    r.used.push_back(ps[i]);
                    r.rows.insert(r.rows.end(), rows[i], rows[i] + q.RF);
                }
            } else {
                const std::vector<int> one{d};
                good = cx.walk(r, 0, s0, one);
            }
            if (!good) break;
        }
        if (good && !r.used.empty()) reals.push_back(std::move(r));
    }
    if (reals.empty()) return 0;
    std::vector<const CxReal *> rs(reals.size());
    for (size_t i = 0; i < reals.size(); ++i) rs[i] = &reals[i];
    std::vector<int32_t> obj, viol;
    if ((rc = cx.eval_patched(rs, obj, viol))) return rc;
    stats[2] += (int32_t)reals.size();
    int best = -1;
    for (size_t i = 0; i < reals.size(); ++i)
        if (viol[i * 8] == 0 && obj[i] > base && (best < 0 || obj[i] > obj[(size_t)best])) best = (int)i;
    if (best < 0) return 0;
    ++stats[3];
    const size_t slots = (size_t)q.P * q.RF;
    std::memcpy(assign, cx.A.data(), slots * 2);
    const CxReal &w = reals[(size_t)best];
    for (size_t u = 0; u < w.used.size(); ++u) std::memcpy(&assign[(size_t)w.used[u] * q.RF], &w.rows[u * q.RF], (size_t)q.RF * 2);
    *new_obj = obj[(size_t)best];
    return 1;
}

}  // namespace

// KAO-CX from a feasible assignment: rounds until nothing improves, `max_rounds` or the deadline (seconds on now_s()'s clock,
// <= 0 = none).  stats: [0] rounds run, [1] rounds that improved, [2] realisations evaluated, [3] improving ones,
// [4] candidates priced > 0, [5] compounds merged, [6] objective before, [7] objective after.
struct CycleCtx { Cx cx; };

CycleCtx *cycle_open(const kao_topic *t, int *rc_out) {
    CycleCtx *c = new CycleCtx();
    const int rc = c->cx.open(t);
    if (rc_out) *rc_out = rc;
    if (rc) { delete c; return nullptr; }
    return c;
}
void cycle_close(CycleCtx *c) { delete c; }

int cycle_run(CycleCtx *c, uint16_t *assign, int32_t max_rounds, double deadline, int64_t *objective, int32_t stats[8],
              int (*poll)(void *), void *poll_arg, bool pairs) {
    int32_t local[8];
    if (!stats) stats = local;
    for (int i = 0; i < 8; ++i) stats[i] = 0;
    Cx &cx = c->cx;
    std::vector<const uint16_t *> one{assign};
    std::vector<int32_t> o0, v0;
    int rc = cx.eval(one, o0, v0);
    if (rc) return rc;
    if (v0[0] != 0) return api_fail(KAO_ERR_INVALID, "KAO-CX starts from a feasible assignment");
    int32_t cur = o0[0];
    stats[6] = cur;
    for (int r = 0; max_rounds <= 0 || r < max_rounds; ++r) {
        if (deadline > 0 && api_now_s() >= deadline) break;
        if (poll && (rc = poll(poll_arg))) return rc;
        int32_t next = cur;
        int got = cx_round(cx, assign, cur, &next, stats);
        ++stats[0];
        if (got < 0) return got;   // a KAO_ERR_* code (negative) from cx_round
        if (got == 0) {            // a fixpoint of the plain layers: the compound-edge layer (test hook KAO_CX_PAIRS=1, see cx_pairs_round)
            const char *pe = std::getenv("KAO_CX_PAIRS");   // read at every fixpoint: the tests switch it inside one process
            if (!(pairs && pe && pe[0] == '1')) break;
            got = cx_pairs_round(cx, assign, cur, &next, stats);
            if (got < 0) return got;
            if (got == 0) break;
        }
        ++stats[1];
        cur = next;
    }
    stats[7] = cur;
    if (objective) *objective = cur;
    return KAO_OK;
}

int cycle_improve(const kao_topic *t, uint16_t *assign, int32_t max_rounds, double deadline, int64_t *objective, int32_t stats[8]) {
    int rc = KAO_OK;
    CycleCtx *c = cycle_open(t, &rc);
    if (!c) return rc;
    rc = cycle_run(c, assign, max_rounds, deadline, objective, stats);
    cycle_close(c);
    return rc;
}

bool cycle_supported(const kao_topic *t) {
    return t && t->rf >= 2 && t->rf <= kCxMaxRF && t->rf_cur <= 8 && t->n_brokers + t->n_racks + 1 <= 2048;
}

}  // namespace kao

extern "C" {

int kao_improve_cycles(const kao_topic *t, uint16_t *assignment, int32_t max_rounds, int64_t *objective, int32_t stats[8]) {
    if (!t || !assignment) return kao::api_fail(KAO_ERR_INVALID, "null topic or assignment");
    return kao::cycle_improve(t, assignment, max_rounds, 0.0, objective, stats);
}

int kao_cycle_matrices(const kao_topic *t, const uint16_t *assignment, int32_t layer, int32_t level, int32_t *dist, int32_t *mid, uint32_t *slot) {
    using namespace kao;
    if (!t || !assignment || !dist || layer < 0 || layer >= kCxLayers || level < 0 || level > kCxLevels) return api_fail(KAO_ERR_INVALID, "bad arguments");
    Cx cx;
    int rc = cx.open(t);
    if (rc) return rc;
    if ((rc = cx.build(assignment))) return rc;
    const CxParams &q = cx.q;
    const size_t nn = (size_t)q.np * q.np;
    std::vector<int32_t> D(nn);
    CX_TRY(hipMemcpy(D.data(), cx.d_D[layer][level], nn * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < q.n; ++i) for (int j = 0; j < q.n; ++j) dist[(size_t)i * q.n + j] = D[(size_t)i * q.np + j];
    if (mid && level >= 1) {
        std::vector<uint16_t> M(nn);
        CX_TRY(hipMemcpy(M.data(), cx.d_M[layer][level], nn * 2, hipMemcpyDeviceToHost));
        for (int i = 0; i < q.n; ++i) for (int j = 0; j < q.n; ++j) mid[(size_t)i * q.n + j] = M[(size_t)i * q.np + j];
    }
    if (slot) {
        std::vector<unsigned long long> E(nn);
        CX_TRY(hipMemcpy(E.data(), cx.d_E[layer], nn * 8, hipMemcpyDeviceToHost));
        for (int i = 0; i < q.n; ++i) for (int j = 0; j < q.n; ++j) slot[(size_t)i * q.n + j] = (uint32_t)(E[(size_t)i * q.np + j] & 0xFFFFFFFFu);
    }
    return KAO_OK;
}

int kao_cycle_seeds(const kao_topic *t, const uint16_t *assignment, int32_t *table, int32_t *n_cfg) {
    using namespace kao;
    if (!t || !assignment) return api_fail(KAO_ERR_INVALID, "bad arguments");
    Cx cx;
    int rc = cx.open(t);
    if (rc) return rc;
    if (n_cfg) *n_cfg = cx.q.ncfg;
    if (!table) return KAO_OK;
    if ((rc = cx.build(assignment)) || (rc = cx.seeds())) return rc;
    for (size_t i = 0; i < cx.table.size(); ++i) { table[2 * i] = cx.table[i].x; table[2 * i + 1] = cx.table[i].y; }
    return KAO_OK;
}

}  // extern "C"
