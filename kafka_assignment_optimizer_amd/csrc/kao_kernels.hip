// kao_kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the Kafka partition-assignment solver.
//
//   k_search : parallel-restart local search "KAO-LS" (DESIGN.md section 4).  One wavefront owns one
//              restart.  Per iteration its 64 lanes either score random slots (tournament) and then scan
//              every target broker / partner slot for the winning slot, or sample their own proposals;
//              every neighbour is delta-evaluated for feasibility (C3,C4,C6,C7; C1,C2,C5 hold by
//              construction) and move cost against broker / rack tables staged in LDS; a DPP min-reduce
//              over the wavefront picks the move.  k_search<false> also keeps the assignment words in
//              LDS; k_search<true> leaves them in HBM/L2 for topics that do not fit.
//   k_bound  : Lagrangian dual bound "KAO-DB" (the optimality certificate): one workgroup per topic prices the
//              coupling rows, rebuilds the candidate pools (wavefront arg-max per rack), one lane per partition solves
//              the priced subproblem exactly, deflected level-controlled Polyak steps in integer fixed point.
//   k_eval   : full evaluation (objective README.md:145-146 and rows C1..C7 README.md:148-180) of
//              complete compact candidates streamed from HBM, one wavefront per candidate, ending in
//              the wavefront -> workgroup -> atomicMin reduce of the packed (violation, cost, id) key.
//
// Integer-only (no floating point on the device path); wave64 throughout; no MFMA (nothing here is a
// dense contraction).  The scalar CPU restatement used by the tests is oracle/kao_port.c.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "kao_device.h"
#include "kao_internal.h"

namespace kao {

// ------------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t fmix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}
__device__ __forceinline__ uint32_t xs32(uint32_t &s) {
    s ^= s << 13; s ^= s >> 17; s ^= s << 5;
    return s;
}
__device__ __forceinline__ uint32_t mulhi(uint32_t a, uint32_t b) { return __umulhi(a, b); }
__device__ __forceinline__ int band(int c, int lo, int hi) { return max(c - hi, 0) + max(lo - c, 0); }
// band(c+1)-band(c) and band(c-1)-band(c)
__device__ __forceinline__ int dinc(int c, int lo, int hi) { return (int)(c >= hi) - (int)(c < lo); }
__device__ __forceinline__ int ddec(int c, int lo, int hi) { return (int)(c <= lo) - (int)(c > hi); }

// A partition's replica slots as the kernels hold them: NW = 4 words (RF <= 4, one ds_read_b128) or 8 words (RF 5..8, two).
template <int NW> struct alignas(16) Part { uint32_t w[NW]; };
// slot k of a partition, as independent selects (keeps the compiler from building a switch)
__device__ __forceinline__ uint32_t sel4(const Part<4> &a, int k) {
    const uint32_t lo = (k & 2) ? a.w[2] : a.w[0];
    const uint32_t hi = (k & 2) ? a.w[3] : a.w[1];
    return (k & 1) ? hi : lo;
}
__device__ __forceinline__ uint32_t sel4(const Part<8> &a, int k) {
    const uint32_t q0 = (k & 4) ? a.w[4] : a.w[0], q1 = (k & 4) ? a.w[5] : a.w[1], q2 = (k & 4) ? a.w[6] : a.w[2], q3 = (k & 4) ? a.w[7] : a.w[3];
    const uint32_t lo = (k & 2) ? q2 : q0, hi = (k & 2) ? q3 : q1;
    return (k & 1) ? hi : lo;
}
template <int NW> __device__ __forceinline__ void set_slot(Part<NW> &a, int k, uint32_t v) {
#pragma unroll
    for (int i = 0; i < NW; ++i) a.w[i] = (i == k) ? v : a.w[i];
}
// per-lane LCG modulo 2^24: one v_mad_u32_u24 (only the low 24 bits of the state are ever read)
__device__ __forceinline__ uint32_t lcg24(uint32_t &s) {
    // s = (s & 0xFFFFFF) * 0x6D2B79 + 0x3C6EF3; forced to the full-rate 24-bit multiply-add (hipcc otherwise
    // picks the quarter-rate v_mul_lo_u32 for some call sites)
    asm("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(s) : "s"(0x6D2B79u), "v"(0x3C6EF3u));
    return s;
}
// uniform-ish draw on [0, n) from the high bits of a 24x24-bit product: one v_mul_hi_u32_u24.  n8 = n << 8.
__device__ __forceinline__ uint32_t rnd24(uint32_t &s, uint32_t n8) {
    const uint32_t v = lcg24(s);
    return (uint32_t)(((unsigned long long)(v & 0xFFFFFFu) * (unsigned long long)(n8 & 0xFFFFFFu)) >> 32);
}
// same draw for ranges that may exceed 65535 (partition indices): floor(v24 * n / 2^24) = mulhi(v24 << 8, n)
__device__ __forceinline__ uint32_t rnd24_wide(uint32_t &s, uint32_t n) {
    const uint32_t v = lcg24(s);
    return __umulhi((v & 0xFFFFFFu) << 8, n);
}
__device__ __forceinline__ uint32_t make_key(int lam, int S, int dV, int dObj, int lane) {
    int delta = __mul24(lam, dV) - __mul24(S, dObj);
    delta = min(max(delta, -kDBias), kDBias - 2);
    return ((uint32_t)(delta + kDBias) << 8) | (uint32_t)lane;
}
// the same keys with a price term dP (Lagrangian prices of the broker rows, already in key units) added to the cost
__device__ __forceinline__ uint32_t make_key_p(int lam, int S, int dV, int dObj, int dP, int lane) {
    int delta = __mul24(lam, dV) - __mul24(S, dObj) + dP;
    delta = min(max(delta, -kDBias), kDBias - 2);
    return ((uint32_t)(delta + kDBias) << 8) | (uint32_t)lane;
}
__device__ __forceinline__ uint32_t make_key_tie_p(int lam, int S, int dV, int dObj, int dP, uint32_t tie) {
    int delta = __mul24(lam, dV) - __mul24(S, dObj) + dP;
    delta = min(max(delta, -kDBias), kDBias - 2);
    return ((uint32_t)(delta + kDBias) << 8) | (tie & 0xFFu);
}
// packed search prices of one broker: low half = replica price a[b], high half = leader price l[b], key units
__device__ __forceinline__ int price_rep(uint32_t pr) { return (int)(short)(pr & 0xFFFFu); }
__device__ __forceinline__ int price_lead(uint32_t pr) { return (int)pr >> 16; }
// Price of one more (p_in) / one fewer (p_out) unit on a priced row whose count is c: the multiplier applies only where the
// count leaves or re-enters its band [lo, hi], i.e. exactly where the violation changes; inside a slack band a unit costs
// nothing (a plain linear term would push the counts of rows with a positive multiplier down to the lower band end).
__device__ __forceinline__ int p_in(int c, int lo, int hi, int price) { return ((c >= hi) | (c < lo)) ? price : 0; }
__device__ __forceinline__ int p_out(int c, int lo, int hi, int price) { return ((c > hi) | (c <= lo)) ? -price : 0; }
// packed broker weights (objective terms per replica / per leader on a broker, kao_topic.broker_w / broker_wl): low | high half
__device__ __forceinline__ int bw_of(uint32_t bw, bool lead) { return (int)(bw & 0xFFFFu) + (lead ? (int)(bw >> 16) : 0); }
// fixed point (kDualScale) -> key units (obj_scale per objective unit), rounded half up, clamped to 16 bits
__device__ __forceinline__ int price_units(int v, int S) { return min(max((S * v + kDualScale / 2) >> kDualLog2, -32767), 32767); }

template <int NW> __device__ __forceinline__ bool in4(const Part<NW> &a, uint32_t w) {
    bool r = false;
#pragma unroll
    for (int i = 0; i < NW; ++i) r |= a.w[i] == w;
    return r;
}
// replicas of the partition that sit in rack r (empty slots carry rack 0xFFFF and never match)
template <int NW> __device__ __forceinline__ int cnt4(const Part<NW> &a, uint32_t r) {
    int n = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) n += (int)((a.w[i] >> 16) == r);
    return n;
}

__device__ __forceinline__ uint32_t make_key_tie(int lam, int S, int dV, int dObj, uint32_t tie) {
    int delta = __mul24(lam, dV) - __mul24(S, dObj);
    delta = min(max(delta, -kDBias), kDBias - 2);
    return ((uint32_t)(delta + kDBias) << 8) | (tie & 0xFFu);
}

struct TopicRegs {  // wave-uniform copy of the fields the inner loop needs
    int P, RF, R, m, Bx;
    uint32_t magic;
    int rep_lo, rep_hi, lead_lo, lead_hi, rack_lo, rack_hi, prack_lo, prack_hi;
    int w00, w01, w10, w11;
};

// objective weight of broker word w on a partition whose current replicas are c, in new role nr
template <int NW> __device__ __forceinline__ int role_w2(const Part<NW> &c, uint32_t w, int wl, int wf) {
    bool fol = false;
#pragma unroll
    for (int i = 1; i < NW; ++i) fol |= c.w[i] == w;
    return (c.w[0] == w) ? wl : (fol ? wf : 0);
}
template <int NW> __device__ __forceinline__ int role_w(const TopicRegs &T, const Part<NW> &c, uint32_t w, int nr) {
    return role_w2(c, w, nr ? T.w01 : T.w00, nr ? T.w11 : T.w10);
}
// internal index -> LDS word (x | rack << 16); 0xFFFF -> empty
__device__ __forceinline__ uint32_t to_word(const TopicRegs &T, uint32_t x) {
    return x == 0xFFFFu ? kNoneW : (x | (mulhi(x, T.magic) << 16));
}
// a restart's state in HBM between launches (LDS path): NW x u16 internal indices per partition
template <int NW> __device__ __forceinline__ Part<NW> load_packed(const TopicRegs &T, const unsigned char *base, int p) {
    Part<NW> a;
    if (NW == 4) {
        const uint2 s = reinterpret_cast<const uint2 *>(base)[p];
        a.w[0] = to_word(T, s.x & 0xFFFFu); a.w[1] = to_word(T, s.x >> 16); a.w[2] = to_word(T, s.y & 0xFFFFu); a.w[3] = to_word(T, s.y >> 16);
    } else {
        const uint4 s = reinterpret_cast<const uint4 *>(base)[p];
        const uint32_t v[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
        for (int i = 0; i < NW / 2; ++i) { a.w[2 * i] = to_word(T, v[i & 3] & 0xFFFFu); a.w[2 * i + 1] = to_word(T, v[i & 3] >> 16); }
    }
    return a;
}
template <int NW> __device__ __forceinline__ void store_packed(unsigned char *base, int p, const Part<NW> &a) {
    if (NW == 4) reinterpret_cast<uint2 *>(base)[p] = make_uint2((a.w[0] & 0xFFFFu) | (a.w[1] << 16), (a.w[2] & 0xFFFFu) | (a.w[3] << 16));
    else reinterpret_cast<uint4 *>(base)[p] = make_uint4((a.w[0] & 0xFFFFu) | (a.w[1] << 16), (a.w[2] & 0xFFFFu) | (a.w[3] << 16),
                                                         (a.w[4 % NW] & 0xFFFFu) | (a.w[5 % NW] << 16), (a.w[6 % NW] & 0xFFFFu) | (a.w[7 % NW] << 16));
}
// sum over all R racks of band(#replicas of the partition in the rack)
template <int NW> __device__ __forceinline__ int part_rack_viol(const TopicRegs &T, const Part<NW> &a) {
    int s = 0, touched = 0;
#pragma unroll
    for (int k = 0; k < NW; ++k) {
        const uint32_t rk = a.w[k] >> 16;
        bool first = a.w[k] != kNoneW;
#pragma unroll
        for (int j = 0; j < k; ++j) first &= (a.w[j] >> 16) != rk;   // this rack has not been counted yet
        if (first) { s += band(cnt4(a, rk), T.prack_lo, T.prack_hi); touched++; }
    }
    return s + (T.R - touched) * T.prack_lo;  // band(0, lo, hi) == lo
}

// ------------------------------------------------------------------------------------------------
// K-search
// ------------------------------------------------------------------------------------------------
constexpr int kTeamRec = 12;   // ints per proposal record of a team (search_body, kTeam)
template <int NW> struct WaveLds {
    Part<NW> *A;  // [P] this restart's assignment, NW words per partition
    uint32_t *C;  // [Bx] replicas | leaders << 16 per broker
    uint16_t *W;  // [Bx] band state of every broker, derived from C and kept current with it (see band_fields); bit 15 = no candidate
    int *K;       // [krt] replicas per rack (krt = search_rack_tab(largest rack count of the launch group))
    int *RT;      // [krt] scratch: rack-dependent part of a REPLACE delta for the slot being scanned
};

// Band state of one broker, precomputed from its counter word c = replicas | leaders << 16 so that delta evaluation costs
// one v_bfe_i32 per row instead of two compares, a select and a subtract.  Replica row (C3) in bits 5:0, leader row (C4) in
// bits 11:6, each: signed 2-bit dinc = band(c + 1) - band(c), signed 2-bit ddec = band(c - 1) - band(c) (README.md:158-166),
// and the two flags that say where a search price applies (p_in / p_out).  Bit 15 marks an index that is no candidate: padding
// slots of the rack-major index space always, and during a REPLACE scan the brokers already in the partition (row C5,
// README.md:168-171).
constexpr int kWIncR = 0, kWDecR = 2, kWPinR = 4, kWPoutR = 5, kWIncL = 6, kWDecL = 8, kWPinL = 10, kWPoutL = 11;
__device__ __forceinline__ int wfld(uint32_t w, int off) { return __builtin_amdgcn_sbfe((int)w, (unsigned)off, 2u); }
__device__ __forceinline__ int wfldw(uint32_t w, int off, uint32_t width) { return __builtin_amdgcn_sbfe((int)w, (unsigned)off, width); }   // width 0 -> 0
__device__ __forceinline__ int wflag(uint32_t w, int off) { return __builtin_amdgcn_sbfe((int)w, (unsigned)off, 1u); }                      // all ones / 0
__device__ __forceinline__ int wflagw(uint32_t w, int off, uint32_t width) { return __builtin_amdgcn_sbfe((int)w, (unsigned)off, width); }
// The six state bits of one band row as a function of where the count c stands: a = clamp(c - lo, -1, 1), b = clamp(c - hi, -1, 1)
//   dinc = (c >= hi) - (c < lo) = (b >= 0) - (a < 0)        ddec = (c <= lo) - (c > hi) = (a <= 0) - (b > 0)
//   pin  = (c >= hi) | (c < lo)   (one more unit leaves / re-enters the band: where a price applies, p_in)
//   pout = (c > hi) | (c <= lo)   (one fewer unit, p_out)
// entry = dinc & 3 | (ddec & 3) << 2 | pin << 4 | pout << 5, nine entries of 6 bits indexed by 3 * (a + 1) + (b + 1) in one 64-bit constant.
constexpr unsigned long long band_entry_of(int a, int b) {
    const int di = (b >= 0 ? 1 : 0) - (a < 0 ? 1 : 0), dd = (a <= 0 ? 1 : 0) - (b > 0 ? 1 : 0);
    const int pin = (b >= 0 || a < 0) ? 1 : 0, pout = (b > 0 || a <= 0) ? 1 : 0;
    return (unsigned long long)((di & 3) | ((dd & 3) << 2) | (pin << 4) | (pout << 5));
}
constexpr unsigned long long band_table() {
    unsigned long long t = 0;
    for (int a = -1; a <= 1; ++a) for (int b = -1; b <= 1; ++b) t |= band_entry_of(a, b) << (6 * (3 * (a + 1) + (b + 1)));
    return t;
}
constexpr unsigned long long kBandTab = band_table();
__device__ __forceinline__ uint32_t band_entry(int c, int lo, int hi) {
    const int a = min(max(c - lo, -1), 1), b = min(max(c - hi, -1), 1);
    return (uint32_t)(kBandTab >> (uint32_t)((__mul24(a, 3) + b + 4) * 6)) & 63u;
}
__device__ __forceinline__ uint32_t band_fields(const TopicRegs &T, uint32_t c) {
    return band_entry((int)(c & 0xFFFFu), T.rep_lo, T.rep_hi) | (band_entry((int)(c >> 16), T.lead_lo, T.lead_hi) << 6);
}
constexpr uint32_t kWNoCand = 0x8000u;
// W[x] for every index of the topic (XR: rack of x, `inv` = padding)
// (`lane`, `stride`: a wavefront strides by 64; the wavefronts of a team stride together by the workgroup size)
template <int NW> __device__ __forceinline__ void rebuild_band_state(const TopicRegs &T, const WaveLds<NW> &L, const uint8_t *XR, uint32_t inv, int lane, int stride = 64) {
    for (int x = lane; x < ((T.Bx + 63) & ~63); x += stride) L.W[x] = (uint16_t)(XR[x] == inv ? kWNoCand : band_fields(T, L.C[x]));
}

// rebuild C and K from A (lanes stride partitions; LDS atomics)
// (a team calls it between two workgroup barriers and zeroes, then counts, with a barrier in between: `stride` > 64)
template <int NW> __device__ __forceinline__ void recount(const TopicRegs &T, const WaveLds<NW> &L, int lane, int stride, int krt) {
    for (int x = lane; x < ((T.Bx + 63) & ~63); x += stride) L.C[x] = 0;
    for (int r = lane; r < krt; r += stride) L.K[r] = 0;
    if (stride > 64) __syncthreads();
    for (int p = lane; p < T.P; p += stride) {
        const Part<NW> a = L.A[p];
#pragma unroll
        for (int k = 0; k < NW; ++k)
            if (a.w[k] != kNoneW) { atomicAdd(&L.C[a.w[k] & 0xFFFFu], k == 0 ? 0x10001u : 1u); atomicAdd(&L.K[a.w[k] >> 16], 1); }
    }
}

// total violation magnitude and objective of the state in LDS (C, K must be current)
// kTeam: the wavefronts of the workgroup split the passes and meet through `TS` (two ints per wavefront); every wavefront
// returns the totals.  Integer sums: the split changes no result.
template <int NW, bool kTeam = false> __device__ __forceinline__ void full_cost(const TopicRegs &T, const WaveLds<NW> &L, const Part<NW> *CUR, const int *RSZ,
                                                            int lane, int stride, int &V, int &obj, const uint32_t *BW = nullptr,
                                                            int *TS = nullptr, int wave = 0, int n_waves = 1) {
    int v = 0, o = 0;
    for (int p = lane; p < T.P; p += stride) {
        const Part<NW> a = L.A[p];
        const Part<NW> c = CUR[p];
#pragma unroll
        for (int k = 0; k < NW; ++k)
            if (a.w[k] != kNoneW) o += role_w(T, c, a.w[k], k == 0 ? 0 : 1);
        v += part_rack_viol(T, a);
    }
    for (int x = lane; x < T.Bx; x += stride) {
        const int r = (int)mulhi((uint32_t)x, T.magic);
        if (x - r * T.m < RSZ[r]) {
            const uint32_t c = L.C[x];
            v += band((int)(c & 0xFFFFu), T.rep_lo, T.rep_hi) + band((int)(c >> 16), T.lead_lo, T.lead_hi);
            if (BW) { const uint32_t bw = BW[x]; o += (int)(c & 0xFFFFu) * (int)(bw & 0xFFFFu) + (int)(c >> 16) * (int)(bw >> 16); }
        }
    }
    for (int r = lane; r < T.R; r += stride) v += band(L.K[r], T.rack_lo, T.rack_hi);
    V = wave_sum(v);
    obj = wave_sum(o);
    if (kTeam) {
        __syncthreads();   // (TS may still be read from the previous call)
        if ((lane & 63) == 0) { TS[2 * wave] = V; TS[2 * wave + 1] = obj; }
        __syncthreads();
        int tv = 0, to = 0;
        if ((lane & 63) < n_waves) { tv = TS[2 * (lane & 63)]; to = TS[2 * (lane & 63) + 1]; }
        V = wave_sum(tv);
        obj = wave_sum(to);
    }
}

template <int NW> __device__ __forceinline__ void snapshot(const TopicRegs &T, const WaveLds<NW> &L, const uint16_t *ext, uint16_t *best, int lane, int stride = 64) {
    for (int p = lane; p < T.P; p += stride) {
        const Part<NW> a = L.A[p];
        uint16_t *o = best + p * T.RF;
#pragma unroll
        for (int k = 0; k < NW; ++k)
            if (k < T.RF) o[k] = ext[a.w[k] & 0xFFFFu];
    }
}

// kGlobalA = false: the restart's assignment words and the topic's current-assignment words are staged in LDS
//                   (topics that fit: the fast path).
// kGlobalA = true : they stay in global memory (HBM / L2) -- 16 B per partition per restart, updated in place --
//                   and only the broker / rack tables live in LDS.  Same algorithm, same results; this is what
//                   lets a single 100k-partition topic run.
// kCurG = true  : (kGlobalA = false only; round 5) the restart's WORKING assignment words live in LDS, the topic's CURRENT-assignment
//                   words are read from global memory (one copy per topic, shared by every restart: it sits in L2).  Holding both in LDS
//                   costs 2 x 16 B per partition, so a restart fitted 160 KiB only up to ~4,900 partitions and 500 x 5000 ran the HBM
//                   path at 4.8 ms a launch; with the working words alone the limit is ~9,800.  Same arithmetic: the replay holds.
// kPriced = true : the cost of a move also carries Lagrangian PRICES of the coupling rows (K-bound's multipliers: replicas
//                   per broker / rack, leaders per broker) -- an augmented-Lagrangian search: with near-optimal prices the
//                   chain steps an improvement needs (objective down a little, violation unchanged) become neutral moves.
// NW              : replica words per partition -- 4 (RF and current RF <= 4) or 8 (up to 8 replicas).
// kWide          : the launch group holds topics of 512 replica slots or more: their tournament scores several slots per lane
//                   and is issued two slots per trip, the REPLACE scan two rounds per trip (instruction-level parallelism for
//                   the one-wavefront-per-SIMD regime of large topics; costs registers the small-topic instantiation keeps)
// kTeam = true  : (topics in global memory only; kernel k_team) the wavefronts of the workgroup are a TEAM on ONE restart: they
//                   share one set of counters and band states in LDS, every wavefront proposes its own move per iteration against
//                   the same frozen state, and a proposal is applied iff it is acceptable and shares no partition, broker or
//                   (when it changes rack totals) rack with an acceptable proposal of a lower-numbered wavefront -- disjoint
//                   moves commute, so violation and objective deltas add up.  A 30,000-partition topic gets W moves per
//                   latency-bound iteration instead of one (the depth large topics lack), deterministically (specification:
//                   oracle/kao_port.c::ls_run with team > 1, replayed bit for bit).
template <bool kGlobalA, bool kPriced, int NW, bool kWide, bool kTeam, bool kCurG = false>
__device__ __forceinline__ void search_body(unsigned char *smem, const SearchPools &pl, const SearchParams &prm) {
    static_assert(!kTeam || kGlobalA, "teams run topics that live in global memory");
    static_assert(!kCurG || !kGlobalA, "kCurG: the working assignment is in LDS");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n_waves = kTeam ? __builtin_amdgcn_readfirstlane((int)(blockDim.x >> 6)) : 1;   // team size W
    const int tid = kTeam ? (int)threadIdx.x : lane, nthr = kTeam ? (int)blockDim.x : 64;   // who strides the O(P) / O(B) passes
    const int2 bm = pl.block_map[blockIdx.x];
    const TopicDev *TD = pl.topics + bm.x;

    TopicRegs T;
    T.P = TD->P; T.RF = TD->RF; T.R = TD->R; T.m = TD->m; T.Bx = TD->Bx; T.magic = TD->magic;
    T.rep_lo = TD->rep_lo; T.rep_hi = TD->rep_hi; T.lead_lo = TD->lead_lo; T.lead_hi = TD->lead_hi;
    T.rack_lo = TD->rack_lo; T.rack_hi = TD->rack_hi; T.prack_lo = TD->prack_lo; T.prack_hi = TD->prack_hi;
    T.w00 = TD->w00; T.w01 = TD->w01; T.w10 = TD->w10; T.w11 = TD->w11;
    // the broker band ends are operands of per-lane compares only: held in VGPRs.  The SGPR file is full (106 + 142 spilled
    // in the plain instantiation) and every spilled scalar costs a v_readlane -- a VALU slot, the unit this kernel is bound
    // by -- where it is used; VGPRs are plentiful (68 of the 72 that keep 7 waves per SIMD).
    asm volatile("" : "+v"(T.rep_lo), "+v"(T.rep_hi), "+v"(T.lead_lo), "+v"(T.lead_hi));

    // ---- LDS carve: [CUR uint4[maxP]]* [RSZ int[krt]] [XR u8[Bx rounded to 64]] then per wave
    //      [A uint4[maxP]]* [C u32[Bx rounded to 64]] [W u16[same]] [K int[krt]] [RT int[krt]]        (* only when !kGlobalA)
    //      krt = search_rack_tab(largest rack count of the launch group): the racks plus one entry for the padding marker
    const int a_bytes = kGlobalA ? 0 : prm.maxP * NW * 4;              // a restart's working words (per wave)
    const int cur_bytes = (kGlobalA || kCurG) ? 0 : prm.maxP * NW * 4;  // the topic's current-assignment words (shared by the workgroup)
    const int bx64 = (prm.maxBx + 63) & ~63;
    const int c_bytes = bx64 * 4;
    const int krt = search_rack_tab(prm.maxR);
    int *RSZ = reinterpret_cast<int *>(smem + cur_bytes);
    uint8_t *XR = smem + cur_bytes + krt * 4;  // rack of internal index x, inv = krt - 1 (never a rack) = padding slot / beyond Bx
    const uint32_t inv = (uint32_t)krt - 1u;
    uint32_t *PR = reinterpret_cast<uint32_t *>(smem + cur_bytes + krt * 4 + bx64);  // [bx64] packed prices (kPriced only)
    const bool hbw = kPriced && prm.bw != 0;   // the launch group carries broker weights (their table is carved only then)
    const int pr_bytes = kPriced ? (hbw ? 2 : 1) * c_bytes + krt * 4 : 0;
    int *PG = reinterpret_cast<int *>(smem + cur_bytes + krt * 4 + bx64 + c_bytes);  // [krt] rack prices (kPriced only)
    uint32_t *BW = reinterpret_cast<uint32_t *>(smem + cur_bytes + krt * 4 + bx64 + c_bytes + krt * 4);  // [bx64] broker weights (kPriced only)
    // per wave: [A] [C] [W] [K] [RT]; a team shares ONE [C] [W] [K], then one [RT] per wavefront and the proposal records
    unsigned char *wb = smem + cur_bytes + krt * 4 + bx64 + pr_bytes + (kTeam ? 0 : wave * (a_bytes + c_bytes + c_bytes / 2 + krt * 8));
    const Part<NW> *cur_words = reinterpret_cast<const Part<NW> *>(pl.cur_pool + TD->cur_off);  // host-prepared words x | rack << 16 (0xFFFFFFFF = none); cur_off counts words
    const Part<NW> *CUR;
    if (kGlobalA || kCurG) CUR = cur_words; else CUR = reinterpret_cast<const Part<NW> *>(smem);
    WaveLds<NW> L;
    L.C = reinterpret_cast<uint32_t *>(wb + a_bytes);
    L.W = reinterpret_cast<uint16_t *>(wb + a_bytes + c_bytes);
    L.K = reinterpret_cast<int *>(wb + a_bytes + c_bytes + c_bytes / 2);
    L.RT = L.K + krt + (kTeam ? wave * krt : 0);
    // team: proposal records, two buffers (iteration parity) x W x kTeamRec ints, then W partial sums x 2
    int *TR = L.K + krt + n_waves * krt;
    int *TS = TR + 2 * 16 * kTeamRec;

    // ---- stage the rack sizes / rack-of-index table (and, when it fits, the current-assignment words) ----
    if (!kGlobalA && !kCurG) {
        Part<NW> *cur_lds = reinterpret_cast<Part<NW> *>(smem);
        for (int p = threadIdx.x; p < T.P; p += blockDim.x) cur_lds[p] = cur_words[p];
    }
    for (int r = threadIdx.x; r < krt; r += blockDim.x) {
        RSZ[r] = r < T.R ? pl.rsz_pool[TD->rsz_off + r] : 0;
        if (kPriced) PG[r] = r < T.R ? price_units(pl.price_pool[TD->price_off + 2 * TD->B + r], prm.obj_scale) : 0;
    }
    __syncthreads();
    for (int x = threadIdx.x; x < ((T.Bx + 63) & ~63); x += blockDim.x) {
        const uint32_t r = mulhi((uint32_t)x, T.magic);
        const bool valid = x < T.Bx && (int)((uint32_t)x - r * (uint32_t)T.m) < RSZ[r < (uint32_t)krt ? r : 0];
        XR[x] = valid ? (uint8_t)r : (uint8_t)inv;
        if (kPriced) {  // prices of broker x in key units: replica price a[b] | leader price l[b] << 16
            uint32_t pr = 0;
            if (valid) {
                const int32_t *pp = pl.price_pool + TD->price_off;
                const int b = pl.ext_pool[TD->ext_off + x];
                pr = ((uint32_t)price_units(pp[b], prm.obj_scale) & 0xFFFFu) | ((uint32_t)price_units(pp[TD->B + b], prm.obj_scale) << 16);
            }
            PR[x] = pr;
            if (hbw) BW[x] = (valid && TD->has_bw) ? pl.bw_pool[TD->bw_off + x] : 0u;
        }
    }
    __syncthreads();

    const int rho = bm.y + (kTeam ? 0 : wave);
    if (rho >= TD->n_restarts) return;  // no block-level barrier below this point (a team is one restart: all or none)
    const int g = TD->restart_base + rho;
    // restart state in HBM: packed 4 x u16 per partition (LDS path, loaded / stored around the launch) or the
    // working words themselves, 16 B per partition, updated in place (global path)
    unsigned char *state_packed = pl.state_pool + TD->state_off + (uint64_t)rho * T.P * (NW * 2);
    if (kGlobalA) L.A = reinterpret_cast<Part<NW> *>(pl.state_pool + TD->state_off) + (uint64_t)rho * T.P;
    else L.A = reinterpret_cast<Part<NW> *>(wb);
    uint16_t *best = pl.best_pool + TD->best_off + (uint64_t)rho * T.P * T.RF;
    const uint16_t *ext = pl.ext_pool + TD->ext_off;
    const uint32_t slo = TD->seed_lo, shi = TD->seed_hi;
    const int S = prm.obj_scale;

    int best_obj, accepted;
    if (prm.init) {
        // surviving current replicas stay in their slots (init == 2: k_init has seeded the state and filled its holes)
        if (prm.init == 1)
        for (int p = tid; p < T.P; p += nthr) {
            Part<NW> c = CUR[p];
#pragma unroll
            for (int k = 1; k < NW; ++k)
                if (k >= T.RF) c.w[k] = kNoneW;
            L.A[p] = c;
        }
        best_obj = -1; accepted = 0;
    } else {
        best_obj = pl.restart_info[g * 4 + 0];
        accepted = pl.restart_info[g * 4 + 3];
        // Elite rule: a restart whose best feasible objective trails the topic's best (as of the previous step) re-seeds
        // its state from that assignment with probability 1/2 (hash of seed, restart, launch); the elite's own restart and
        // the restarts that tie with it keep going, so the population stays diverse.
        bool reseed = false;
        if (prm.elite) {
            const unsigned long long ek = pl.elite_key[bm.x];
            const int e_obj = (int)kObjCap - (int)((ek >> 20) & 0xFFFFFFull);
            reseed = ek != ~0ull && (ek >> 44) == 0 && (int)(ek & 0xFFFFFull) != rho && best_obj < e_obj &&
                     (fmix32(slo ^ ((uint32_t)rho * 0x9E3779B1u) ^ (prm.launch * 0x85EBCA77u) ^ 0xE117Eu) & 1u);
        }
        if (reseed) {
            const uint16_t *ea = pl.elite_assign + TD->win_off;
            const uint16_t *io = pl.int_pool + TD->int_off;
            for (int p = tid; p < T.P; p += nthr) {
                Part<NW> w;
#pragma unroll
                for (int k = 0; k < NW; ++k) w.w[k] = k < T.RF ? to_word(T, io[ea[p * T.RF + k]]) : kNoneW;
                L.A[p] = w;
            }
        } else if (!kGlobalA) {
            for (int p = lane; p < T.P; p += 64) L.A[p] = load_packed<NW>(T, state_packed, p);
        }
    }
    if (kGlobalA) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // own stores visible to every lane's loads
    if (kTeam) __syncthreads();
    recount(T, L, tid, nthr, krt);
    if (kTeam) __syncthreads();

    if (prm.init == 1) {
        // ---- hole filling by best insertion, holes in (p,k) order.  Partitions are inspected 64 at a time (one per
        //      lane); only those with a hole are visited, in ascending order. ----
        // Two passes: leader holes of all partitions first, then follower holes (leaders are the scarcer resource).
        // Round 4: the partitions that have a hole are listed by the host (TopicDev::hole_off: static, they depend on the current
        // assignment only) -- inspecting all P partitions twice, 64 per trip with a global load each, was most of the 70-ms first
        // launch of a 100,000-partition topic, holes or not.  Same holes, same order.
        const uint32_t *HL = pl.cur_pool + TD->hole_off;
        if (!kTeam || wave == 0)   // (a team: its first wavefront fills the holes, in the same order as a single one)
        for (int pass = 0; pass < 2; ++pass) {
            const uint32_t n_holes = HL[pass], *hl = HL + 2 + (pass ? HL[0] : 0u);
            // the next hole's rows are loaded while this one is scanned (a pass lists every partition once, so the rows of the next
            // hole cannot be written by this one): config 5 as one topic has 15,000 holes, each two global round trips apart
            int p_n = n_holes ? (int)hl[0] : 0;
            Part<NW> a_n = L.A[p_n], c_n = CUR[p_n];
            for (uint32_t hi = 0; hi < n_holes; ++hi) {
                const int p = p_n;
                Part<NW> a = a_n;  // same address in every lane: broadcast
                const Part<NW> c = c_n;
                if (hi + 1 < n_holes) { p_n = (int)hl[hi + 1]; a_n = L.A[p_n]; c_n = CUR[p_n]; }
#pragma unroll
                for (int k = 0; k < NW; ++k) {
                    if (k >= T.RF) break;
                    if ((k == 0) != (pass == 0)) continue;  // this pass handles the other kind of slot
                    if (a.w[k] != kNoneW) continue;  // wave-uniform
                    // best insertion: every valid broker not in the partition, 64 per round (lane = internal index)
                    const uint32_t hmix = slo ^ fmix32(shi + (uint32_t)rho * 0x9E3779B1u + (uint32_t)(p * NW + k) * 0x27D4EB2Fu + 0x5BD1E995u + prm.gen * 0x632BE5ABu);
                    const int wl = k == 0 ? T.w00 : T.w01, wf = k == 0 ? T.w10 : T.w11;
                    uint32_t key = kKeyNull, xw_l = kNoneW;
#pragma unroll 2
                    for (int base = 0; base < T.Bx; base += 64) {
                        const uint32_t x = (uint32_t)(base + lane);
                        const uint32_t r = XR[x];
                        const uint32_t xw = x | (r << 16);
                        const bool okx = (r != inv) & !in4(a, xw);
                        const uint32_t cn = L.C[x];
                        const uint32_t rk = r;   // (padding lanes, no candidates anyway, read the spare entry krt - 1)
                        int dV = dinc((int)(cn & 0xFFFFu), T.rep_lo, T.rep_hi) + dinc(L.K[rk], T.rack_lo, T.rack_hi) +
                                 dinc(cnt4(a, r), T.prack_lo, T.prack_hi);
                        if (k == 0) dV += dinc((int)(cn >> 16), T.lead_lo, T.lead_hi);
                        const uint32_t tie = fmix32(hmix + x * 0x165667B1u) >> 24;
                        uint32_t keyx;
                        if (kPriced) {
                            const uint32_t prx = PR[x];
                            int dP = p_in((int)(cn & 0xFFFFu), T.rep_lo, T.rep_hi, price_rep(prx)) + p_in(L.K[rk], T.rack_lo, T.rack_hi, PG[rk]);
                            if (k == 0) dP += p_in((int)(cn >> 16), T.lead_lo, T.lead_hi, price_lead(prx));
                            keyx = okx ? make_key_tie_p(prm.lam_max, S, dV, role_w2(c, xw, wl, wf) + (hbw ? bw_of(BW[x], k == 0) : 0), dP, tie) : kKeyNull;
                        }
                        else keyx = okx ? make_key_tie(prm.lam_max, S, dV, role_w2(c, xw, wl, wf), tie) : kKeyNull;
                        if (keyx < key) { key = keyx; xw_l = xw; }
                    }
                    const uint32_t kmin = wave_umin(key);
                    const unsigned long long bal = __ballot(key == kmin);
                    const uint32_t xw_win = (uint32_t)__builtin_amdgcn_readlane((int)xw_l, __ffsll((long long)bal) - 1);  // ties: lowest lane
                    a.w[k] = xw_win;
                    if (lane == 0) {
                        reinterpret_cast<uint32_t *>(&L.A[p])[k] = xw_win;
                        L.C[xw_win & 0xFFFFu] += (k == 0) ? 0x10001u : 1u;
                        L.K[xw_win >> 16] += 1;
                    }
                }
            }
        }
        if (kGlobalA) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        if (kTeam) __syncthreads();
    }

    int V, obj;
    full_cost<NW, kTeam>(T, L, CUR, RSZ, tid, nthr, V, obj, hbw ? BW : nullptr, TS, wave, n_waves);
    if (V == 0 && obj > best_obj) { best_obj = obj; snapshot(T, L, ext, best, tid, nthr); }
    rebuild_band_state(T, L, XR, inv, tid, nthr);     // W from the counters; kept current by every accepted move below
    if (lane == 0) L.RT[krt - 1] = 0;                 // the spare entry padding lanes read
    if (kTeam) __syncthreads();

    // ---- per-lane RNG stream of this launch (LCG mod 2^24, re-keyed every launch) ----
    uint32_t rng = fmix32(slo ^ fmix32(shi + (uint32_t)rho * 0x9E3779B1u + prm.launch * 0x85EBCA77u + (uint32_t)tid * 0xC2B2AE3Du));   // (team: wavefront w's lanes are streams 64 w .. 64 w + 63)

    const int plog = TD->period_log2 + (rho & 3);
    const uint32_t pmask = (1u << plog) - 1u;
    const uint32_t lrange = (uint32_t)(prm.lam_max - prm.lam_min + 1);
    const uint32_t RF8 = (uint32_t)T.RF << 8, R8 = (uint32_t)T.R << 8, m8 = (uint32_t)T.m << 8;  // all < 65536

    const int T_tour = min(64, max(4, (T.P * T.RF) >> 2));  // lanes taking part in the slot tournament
    uint32_t tour_off = lane < T_tour ? 0u : kKeyNull;       // OR-ed into a lane's tournament key (a select would put a branch between the slots of a trip)
    asm volatile("" : "+v"(tour_off));
    const int GA = min(16, max(1, (T.P * T.RF) >> 8));       // random slots scored per lane
    const int x_rounds_full = (T.P + 63) >> 6;
    const bool x_windowed = x_rounds_full > 8;               // EXCHANGE scans at most 8 rounds of 64 partitions

    // the launch parameters the loop needs, as scalars of their own: read straight from `prm` they stay one 8-dword register
    // tuple (the kernarg load) that the allocator spills and restores WHOLE -- three times per iteration, 24 v_readlane
    int lam_lo = prm.lam_min, lam_hi = prm.lam_max;
    uint32_t n_iters = prm.iters, it_base = prm.launch * prm.iters;
    uint32_t scan_two = TD->P * TD->RF <= prm.scan2_max ? 1u : 0u;   // REPLACE scan over two tournament slots (topics up to kScanTwoSlots replica slots)
    {   // through a VGPR and back: a plain scalar copy is coalesced with the tuple again
        uint32_t v0 = (uint32_t)lam_lo, v1 = (uint32_t)lam_hi, v2 = n_iters, v3 = it_base, v4 = scan_two;
        asm volatile("" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4));
        lam_lo = (int)__builtin_amdgcn_readfirstlane(v0); lam_hi = (int)__builtin_amdgcn_readfirstlane(v1);
        n_iters = __builtin_amdgcn_readfirstlane(v2); it_base = __builtin_amdgcn_readfirstlane(v3);
        scan_two = __builtin_amdgcn_readfirstlane(v4);
    }
    for (uint32_t i = 0; i < n_iters; ++i) {
        const uint32_t it = it_base + i;
        const int type = (int)((0x1210u >> ((it & 7u) * 2u)) & 3u);  // pattern R R X R L R X R
        const uint32_t ph = it & pmask;
        // no oscillation before the restart has been feasible once (best_obj < 0): the penalty stays at lam_max
        const int lam = best_obj < 0 ? lam_hi : min(lam_hi, lam_lo + (int)((2u * ph * lrange) >> plog));
        // REPLACE alternates, in blocks of 8 iterations, between "scan" (one slot, every broker) and "sample"
        // (every lane its own slot, 4 brokers); EXCHANGE always scans; LEADER-SWAP always samples
        const bool sampled = (type == 2) || (type == 0 && ((it >> 3) & 1u));

        // this lane's best proposal of the iteration
        uint32_t key = kKeyNull;
        int dV = 0, dObj = 0, p = 0, k = 0, q = 0, j = 0;
        uint32_t uw = 0, vw = 0;
        uint32_t kmin;
        int win;

        // Every violation delta of the broker rows (C3, C4) below comes from the brokers' band state W (band_fields): a signed
        // 2-bit field per (row, direction) instead of two compares, a select and a subtract on the counter word; the prices
        // of the priced instantiation apply where the matching "count leaves / re-enters its band" flag is set.
        if (sampled) {
            p = (int)rnd24_wide(rng, (uint32_t)T.P);
            const Part<NW> a = L.A[p];
            const Part<NW> c = CUR[p];
            if (type == 0) {  // REPLACE (p,k) <- x_g: 2 candidates of any rack, 2 of the old broker's rack
                k = (int)rnd24(rng, RF8);
                uw = sel4(a, k);
                const uint32_t ro = uw >> 16;
                const bool lead = k == 0;
                const uint32_t lw = lead ? 2u : 0u;   // width of a leader field: a zero-width extract yields 0 for follower slots
                const int wl = lead ? T.w00 : T.w01, wf = lead ? T.w10 : T.w11;
                const int g_old = role_w2(c, uw, wl, wf) + (hbw ? bw_of(BW[uw & 0xFFFFu], lead) : 0);
                const uint32_t wo = L.W[uw & 0xFFFFu];
                const int dV_old = wfld(wo, kWDecR) + wfldw(wo, kWDecL, lw);
                const int dV_rack_old = ddec(L.K[ro], T.rack_lo, T.rack_hi) + ddec(cnt4(a, ro), T.prack_lo, T.prack_hi);
                const int rsz_ro = RSZ[ro];
                int dP_old = 0, dP_rack_old = 0;
                if (kPriced) {
                    const uint32_t pro = PR[uw & 0xFFFFu];
                    dP_old = -(wflag(wo, kWPoutR) & price_rep(pro));
                    if (lead) dP_old -= wflag(wo, kWPoutL) & price_lead(pro);
                    dP_rack_old = p_out(L.K[ro], T.rack_lo, T.rack_hi, PG[ro]);
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    uint32_t r, jj;
                    bool okg;
                    if (g < 2) {
                        r = rnd24(rng, R8);
                        jj = rnd24(rng, m8);
                        okg = (int)jj < RSZ[r];
                    } else {
                        r = ro;
                        jj = rnd24(rng, (uint32_t)rsz_ro << 8);
                        okg = true;
                    }
                    const uint32_t x = __umul24(r, (uint32_t)T.m) + jj;
                    const uint32_t xw = x | (r << 16);
                    okg = okg && !in4(a, xw);
                    const uint32_t wn = L.W[x];
                    int dVg = dV_old + wfld(wn, kWIncR) + wfldw(wn, kWIncL, lw);
                    if (g < 2) {
                        if (r != ro)
                            dVg += dV_rack_old + dinc(L.K[r], T.rack_lo, T.rack_hi) + dinc(cnt4(a, r), T.prack_lo, T.prack_hi);
                    }
                    const int dObjg = role_w2(c, xw, wl, wf) + (hbw ? bw_of(BW[x], lead) : 0) - g_old;
                    uint32_t keyg;
                    if (kPriced) {
                        const uint32_t prx = PR[x];
                        int dPg = dP_old + (wflag(wn, kWPinR) & price_rep(prx));
                        if (lead) dPg += wflag(wn, kWPinL) & price_lead(prx);
                        if (g < 2 && r != ro) dPg += dP_rack_old + p_in(L.K[r], T.rack_lo, T.rack_hi, PG[r]);
                        keyg = okg ? make_key_p(lam, S, dVg, dObjg, dPg, lane) : kKeyNull;
                    }
                    else keyg = okg ? make_key(lam, S, dVg, dObjg, lane) : kKeyNull;
                    if (keyg < key) { key = keyg; vw = xw; dV = dVg; dObj = dObjg; }
                }
            } else {  // LEADER SWAP inside p: slot 0 <-> slot k, every k = 1..RF-1 is a candidate
                uw = a.w[0];
                const int u_lead = role_w2(c, uw, T.w00, T.w10), u_fol = role_w2(c, uw, T.w01, T.w11);
                const uint32_t wu = L.W[uw & 0xFFFFu];
                const int dV_u = wfld(wu, kWDecL);
                const int dP_u = kPriced ? -(wflag(wu, kWPoutL) & price_lead(PR[uw & 0xFFFFu])) : 0;
#pragma unroll
                for (int kk = 1; kk < NW; ++kk) {
                    if (kk >= T.RF) break;
                    const uint32_t xw = a.w[kk];
                    const int dObjg = role_w2(c, xw, T.w00, T.w10) + u_fol - u_lead - role_w2(c, xw, T.w01, T.w11) +
                                      (hbw ? (int)(BW[xw & 0xFFFFu] >> 16) - (int)(BW[uw & 0xFFFFu] >> 16) : 0);
                    const uint32_t wx = L.W[xw & 0xFFFFu];
                    const int dVg = dV_u + wfld(wx, kWIncL);
                    uint32_t keyg;
                    if (kPriced) keyg = make_key_p(lam, S, dVg, dObjg, dP_u + (wflag(wx, kWPinL) & price_lead(PR[xw & 0xFFFFu])), lane);
                    else keyg = make_key(lam, S, dVg, dObjg, lane);
                    if (keyg < key) { key = keyg; vw = xw; k = kk; dV = dVg; dObj = dObjg; }
                }
            }
            kmin = wave_umin(key);  // wavefront min-scan over the lanes' best proposals
            win = (int)(kmin & 63u);
        } else {
            // ---- phase A: tournament over T_tour random slots; lowest removal score wins the iteration ----
            // Score one random slot: the cost of taking its replica out under the current penalty.  REPLACE: the replica
            // leaves its broker and (at best) its rack.  EXCHANGE: broker and rack totals do not change; only the
            // partition's own rack spread (C7) and who leads (a leader slot may shed a leader, a follower slot may
            // gain one) can improve.  `type` is wave-uniform, so only one branch is ever executed.
            uint32_t keyA, oldw_l;
            int pl_, kl_, g_old_l, dvo_l = 0, dvr_l = 0;
            // `ty` is the (wave-uniform) move type as a compile-time constant: each instantiation is straight-line code, so the
            // two slots of a trip below interleave -- a big topic runs one wavefront per SIMD and the LDS round trips of one
            // slot (A[p], then W[x] / K[rack]) are hidden behind the other slot's, not behind other wavefronts
            // (p_o comes in: the lane's first slot is in a random partition, its further slots in the partitions that follow it
            //  cyclically -- on topics that live in HBM the slots of a lane then share two cache lines instead of touching 16)
            auto score_words = [&](auto ty, uint32_t &key_o, const Part<NW> &al, const Part<NW> &cl, int k_o, uint32_t &oldw_o, int &g_o, int &dvo_o, int &dvr_o) {
                constexpr int TY = decltype(ty)::value;
                oldw_o = sel4(al, k_o);
                const uint32_t rol = oldw_o >> 16;
                const bool leadl = k_o == 0;
                g_o = role_w2(cl, oldw_o, leadl ? T.w00 : T.w01, leadl ? T.w10 : T.w11);
                if (hbw && TY == 0) g_o += bw_of(BW[oldw_o & 0xFFFFu], leadl);   // a REPLACE also gives up the broker's own weight
                const uint32_t wo = L.W[oldw_o & 0xFFFFu];
                const int dv7 = ddec(cnt4(al, rol), T.prack_lo, T.prack_hi);
                int sc;
                if (TY == 0) {
                    dvo_o = wfld(wo, kWDecR) + wfldw(wo, kWDecL, leadl ? 2u : 0u);
                    dvr_o = ddec(L.K[rol], T.rack_lo, T.rack_hi) + dv7;
                    sc = dvo_o + min(dvr_o, 0);
                } else {
                    const int dvl = wfld(wo, leadl ? kWDecL : kWIncL);
                    sc = min(dv7, 0) + min(dvl, 0);
                }
                if (kPriced) {
                    int dPs = 0;
                    if (TY == 0) {
                        const uint32_t pro = PR[oldw_o & 0xFFFFu];
                        dPs = -(wflag(wo, kWPoutR) & price_rep(pro));
                        dPs -= wflagw(wo, kWPoutL, leadl ? 1u : 0u) & price_lead(pro);
                    }
                    key_o = make_key_p(lam, S, sc, -g_o, dPs, lane) | tour_off;
                }
                else key_o = make_key(lam, S, sc, -g_o, lane) | tour_off;   // lanes outside the tournament: all ones = kKeyNull
            };
            auto score_slot = [&](auto ty, uint32_t &key_o, int p_o, int &k_o, uint32_t &oldw_o, int &g_o, int &dvo_o, int &dvr_o) {
                k_o = (int)rnd24(rng, RF8);
                const Part<NW> al = L.A[p_o];
                const Part<NW> cl = CUR[p_o];
                score_words(ty, key_o, al, cl, k_o, oldw_o, g_o, dvo_o, dvr_o);
            };
            // large topics: up to 16 slots per lane, the lane keeps its best; two slots per trip.  Draw and comparison order
            // are those of a one-at-a-time loop.
            auto tournament = [&](auto ty) {
                const int p0 = (int)rnd24_wide(rng, (uint32_t)T.P);
                auto part = [&](int ga) { const int q2 = p0 + ga; return q2 < T.P ? q2 : q2 - T.P; };   // GA <= 16 <= P whenever GA > 1
                pl_ = p0;
                score_slot(ty, keyA, p0, kl_, oldw_l, g_old_l, dvo_l, dvr_l);
                int ga = 1;
                for (; kWide && ga + 1 < GA; ga += 2) {
                    uint32_t kg0, ow0, kg1, ow1;
                    int kk0, gg0, d10 = 0, d20 = 0, kk1, gg1, d11 = 0, d21 = 0;
                    const int pg0 = part(ga), pg1 = part(ga + 1);
                    score_slot(ty, kg0, pg0, kk0, ow0, gg0, d10, d20);
                    score_slot(ty, kg1, pg1, kk1, ow1, gg1, d11, d21);
                    if (kg0 < keyA) { keyA = kg0; pl_ = pg0; kl_ = kk0; oldw_l = ow0; g_old_l = gg0; dvo_l = d10; dvr_l = d20; }
                    if (kg1 < keyA) { keyA = kg1; pl_ = pg1; kl_ = kk1; oldw_l = ow1; g_old_l = gg1; dvo_l = d11; dvr_l = d21; }
                }
                for (; ga < GA; ++ga) {
                    uint32_t kg, ow;
                    int kk, gg, d1 = 0, d2 = 0;
                    const int pg = part(ga);
                    score_slot(ty, kg, pg, kk, ow, gg, d1, d2);
                    if (kg < keyA) { keyA = kg; pl_ = pg; kl_ = kk; oldw_l = ow; g_old_l = gg; dvo_l = d1; dvr_l = d2; }
                }
            };
            // Topics that live in global memory (round 4): every draw of the iteration is independent of what is loaded, so all
            // of them come first, then ALL the loads of the iteration -- the 2 x 16 tournament words of the lane and, for an
            // EXCHANGE, the partner words of the first XB rounds -- are issued back to back, and only then scored: one global round
            // trip per iteration where the slot-by-slot form took 8 for the tournament, one more for the winner's words and one per
            // partner round (a restart is one wavefront per SIMD: nothing else hides that latency).  The lane's 16 loads of
            // consecutive partitions also hit the same two or three cache lines while they are still in the L1.  Draw and
            // comparison order are those of the slot-by-slot loop (same spec, same replay); slots g >= GA of a small topic are
            // loaded from the lane's first partition and masked out of the comparison.
            Part<NW> a_l, c_l;   // the words of the lane's best slot: the winner's are broadcast, not re-read
            constexpr int XB = !kGlobalA ? 1 : ((NW == 4 && !kTeam) ? 8 : 4);   // (a team runs two wavefronts per SIMD: 256 registers each)
            Part<NW> xb[XB], xcb[XB];
            int q0 = 0;
            auto x_partner = [&](int rd, int &qq, bool &okq) {   // partition lane `lane` looks at in partner round rd
                qq = q0 + rd * 64 + lane; okq = true;
                if (qq >= T.P) { if (x_windowed) qq -= T.P; else okq = false; }
                return min(qq, T.P - 1);
            };
            auto tournament_global = [&](auto ty) {
                constexpr int TY = decltype(ty)::value;
                constexpr int TB = (NW == 4 && !kTeam) ? 16 : 8;   // slots per batch: 2 x NW x TB registers of loads in flight
                const int p0 = (int)rnd24_wide(rng, (uint32_t)T.P);
                auto part = [&](int ga) { const int q2 = p0 + ga; return q2 < T.P ? q2 : q2 - T.P; };
                int kk[16];
#pragma unroll
                for (int g = 0; g < 16; ++g) { kk[g] = 0; if (g < GA) kk[g] = (int)rnd24(rng, RF8); }
                if (TY == 1) {
                    const int q_draw = (int)rnd24_wide(rng, (uint32_t)T.P);   // every lane draws; lane 0's value places the window
                    q0 = x_windowed ? __builtin_amdgcn_readfirstlane(q_draw) : 0;
#pragma unroll
                    for (int r2 = 0; r2 < XB; ++r2) { int qq; bool okq; const int qc = x_partner(r2, qq, okq); xb[r2] = L.A[qc]; xcb[r2] = CUR[qc]; }
                }
#pragma unroll
                for (int g0 = 0; g0 < 16; g0 += TB) {
                    if (g0 > 0 && g0 >= GA) break;   // wave-uniform
                    Part<NW> tal[TB], tcl[TB];
                    int tp[TB];
#pragma unroll
                    for (int g = 0; g < TB; ++g) { tp[g] = (g0 + g) < GA ? part(g0 + g) : p0; tal[g] = L.A[tp[g]]; tcl[g] = CUR[tp[g]]; }
#pragma unroll
                    for (int g = 0; g < TB; ++g) {
                        uint32_t kg, ow;
                        int gg, d1 = 0, d2 = 0;
                        score_words(ty, kg, tal[g], tcl[g], kk[g0 + g], ow, gg, d1, d2);
                        if (g0 + g > 0) kg |= (g0 + g) < GA ? 0u : kKeyNull;
                        if (g0 + g == 0 || kg < keyA) { keyA = kg; pl_ = tp[g]; kl_ = kk[g0 + g]; oldw_l = ow; g_old_l = gg; dvo_l = d1; dvr_l = d2; a_l = tal[g]; c_l = tcl[g]; }
                    }
                }
            };
            if constexpr (kGlobalA) { if (type == 0) tournament_global(std::integral_constant<int, 0>{}); else tournament_global(std::integral_constant<int, 1>{}); }
            else if (type == 0) tournament(std::integral_constant<int, 0>{}); else tournament(std::integral_constant<int, 1>{});
            const int wA1 = (int)(wave_umin(keyA) & 63u);
            // REPLACE scan (round 4): the slots of the tournament's TWO best lanes are scanned, the winner's first; the better of
            // the two moves is the proposal (ties: the first slot's).  Loop head, penalty, acceptance and bookkeeping are paid once
            // for twice the neighbours (they were two thirds of the instruction stream at 500 brokers, docs/notes_r03.md section 6).
            int wA2 = -1;
            if (type == 0 && scan_two) {   // (large topics scan one slot: their iterations are what they are short of)
                const uint32_t k2 = wave_umin(lane == wA1 ? kKeyNull : keyA);
                wA2 = k2 == kKeyNull ? -1 : (int)(k2 & 63u);   // (no other lane takes part: one slot)
            }
            uint32_t b_kmin = kKeyNull, b_uw = 0, b_vw = 0;
            int b_win = 0, b_p = 0, b_k = 0, b_dV = 0, b_dObj = 0;
#pragma nounroll
            for (int si = 0; si < (wA2 >= 0 ? 2 : 1); ++si) {
            const int wA = si == 0 ? wA1 : wA2;
            p = __builtin_amdgcn_readlane(pl_, wA);
            k = __builtin_amdgcn_readlane(kl_, wA);
            uw = (uint32_t)__builtin_amdgcn_readlane((int)oldw_l, wA);
            const int g_old = __builtin_amdgcn_readlane(g_old_l, wA);
            const int dV_old = __builtin_amdgcn_readlane(dvo_l, wA);
            const int dV_rack_old = __builtin_amdgcn_readlane(dvr_l, wA);
            Part<NW> a, c;
            if constexpr (kGlobalA) {
#pragma unroll
                for (int i2 = 0; i2 < NW; ++i2) { a.w[i2] = (uint32_t)__builtin_amdgcn_readlane((int)a_l.w[i2], wA); c.w[i2] = (uint32_t)__builtin_amdgcn_readlane((int)c_l.w[i2], wA); }
            } else {
                a = L.A[p];   // same address in every lane: LDS broadcast
                c = CUR[p];
            }
            const bool lead = k == 0;  // wave-uniform
            const uint32_t ro = uw >> 16;
            if (type == 0) {
                // ---- phase B (REPLACE): every target broker for slot (p,k), 64 per round -- the band-state scan.  A candidate
                //      costs one W read (band deltas + where its rack's entry of RT lives), one RT read, two bit-field
                //      extracts, the cost and the key; the per-lane running minimum is a plain v_min_u32 because the key carries
                //      the round number below the tie bits; only the winner's details are recomputed afterwards.  Same
                //      candidates, same keys, same winner as the scalar restatement (oracle/kao_port.c).
                {   // rack-dependent part of the delta, racks strided over the lanes
                    const int dP_rack_old = kPriced ? p_out(L.K[ro], T.rack_lo, T.rack_hi, PG[ro]) : 0;
                    for (int r = lane; r < T.R; r += 64) {
                        int v = ((uint32_t)r != ro) ? dV_rack_old + dinc(L.K[r], T.rack_lo, T.rack_hi) + dinc(cnt4(a, (uint32_t)r), T.prack_lo, T.prack_hi) : 0;
                        if (kPriced)  // rack part of the price delta rides in the upper 24 bits (the violation delta is within -8..8)
                            v = (v & 0xFF) | ((((uint32_t)r != ro) ? dP_rack_old + p_in(L.K[r], T.rack_lo, T.rack_hi, PG[r]) : 0) * 256);
                        L.RT[r] = v;
                    }
                }
                const int wl = lead ? T.w00 : T.w01, wf = lead ? T.w10 : T.w11;
                // lane i < NW looks after slot i of the partition: (1) the broker it holds is no candidate (row C5) -- for the
                // duration of the scan its W entry points at the reserved RT entry; (2) current replica i, when displaced (in c,
                // not in a), is the only broker with a non-zero objective weight here: the round it falls into is scored with weights
                const int li = lane & (NW - 1);
                const uint32_t ai = sel4(a, li), ci = sel4(c, li);
                // (a team shares W: nothing may be marked there -- the rounds that hold a broker of the partition take the slow
                //  path below, like the rounds with a displaced current replica, and drop it by comparison)
                const bool holds = (lane < NW) & (ai != kNoneW);
                uint32_t w_keep = 0;
                if (!kTeam && holds) {
                    w_keep = L.W[ai & 0xFFFFu];
                    L.W[ai & 0xFFFFu] = (uint16_t)(w_keep | kWNoCand);
                }
                int ar[NW];
                {
                    const int ar_l = (kTeam && holds) ? (int)((ai & 0xFFFFu) >> 6) : -1;
#pragma unroll
                    for (int i2 = 0; i2 < NW; ++i2) ar[i2] = __builtin_amdgcn_readlane(ar_l, i2);
                }
                const bool hm_l = (lane < NW) & (ci != kNoneW) & !in4(a, ci);
                const int mr_l = hm_l ? (int)((ci & 0xFFFFu) >> 6) : -1;
                int mr[NW];
#pragma unroll
                for (int i2 = 0; i2 < NW; ++i2) mr[i2] = __builtin_amdgcn_readlane(mr_l, i2);
                bool has_missing = false;
#pragma unroll
                for (int i2 = 0; i2 < NW; ++i2) has_missing |= mr[i2] >= 0;
                // cost + bias of a candidate = lam * (its own delta + dV_old) + S * g_old (+ prices) (- S * weight of a displaced
                // current replica): everything that does not depend on the candidate is one wave-uniform constant
                int K0 = __mul24(S, g_old) + kDBias + __mul24(lam, dV_old);
                if (kPriced) {
                    const uint32_t pro = PR[uw & 0xFFFFu];
                    const uint32_t wo = L.W[uw & 0xFFFFu];
                    K0 -= wflag(wo, kWPoutR) & price_rep(pro);
                    if (lead) K0 -= wflag(wo, kWPoutL) & price_lead(pro);
                }
                const uint32_t lw = lead ? 2u : 0u, lf = lead ? 1u : 0u;   // widths of the leader fields / flags: zero for follower slots
                uint32_t bestA = kKeyNull;   // (cost + bias) << 16 | tie << 8 | round within the chunk of 256 rounds
                int chunkA = 0;
                // one round: 64 candidates x = base + lane, `rd` = round within the chunk of 256; `with_w` (compile time): the
                // round holds a displaced current replica of the partition and is scored with objective weights
                auto scan_round = [&](auto with_w, int base, int rd) -> uint32_t {
                    const uint32_t st = lcg24(rng);   // tie bits = bits 8..15 of the draw, as make_key_tie(lcg24 >> 8)
                    const int w = (int)(short)L.W[base + lane];   // sign-extended: bit 15 fills the upper half
                    const int rt = L.RT[XR[base + lane]];
                    int dsc;
                    if (kPriced) {   // RT entry: violation delta in the low byte, rack price above it
                        const uint32_t prx = PR[base + lane];
                        const int dVx = wfld(w, kWIncR) + wfldw(w, kWIncL, lw) + (int)(signed char)(rt & 0xFF);
                        dsc = __mul24(lam, dVx) + K0 + (rt >> 8) + (wflag(w, kWPinR) & price_rep(prx)) + (wflagw(w, kWPinL, lf) & price_lead(prx));
                        if (hbw) dsc -= __mul24(S, bw_of(BW[base + lane], lead));
                    } else {
                        const int dVx = wfld(w, kWIncR) + wfldw(w, kWIncL, lw) + rt;
                        dsc = __mul24(lam, dVx) + K0;
                    }
                    uint32_t member = 0;
                    if (decltype(with_w)::value) {
                        const uint32_t x = (uint32_t)(base + lane);
                        const uint32_t xw = x | ((uint32_t)XR[x] << 16);
                        dsc -= __mul24(S, role_w2(c, xw, wl, wf));
                        if (kTeam && in4(a, xw)) member = 0xFFFF0000u;   // row C5: already in the partition
                    }
                    dsc = min(max(dsc, 0), 2 * kDBias - 2);
                    // a "no candidate" index gets a cost field of 0xFFFF: above every real cost (<= 2 * kDBias - 2) and never accepted
                    return ((uint32_t)dsc << 16) | (st & 0xFF00u) | (uint32_t)rd | ((uint32_t)w & 0xFFFF0000u) | member;
                };
                auto is_weighted = [&](int rdg) {   // wave-uniform
                    bool wgt = false;
#pragma unroll
                    for (int i2 = 0; i2 < NW; ++i2) wgt |= (mr[i2] == rdg) | (kTeam && ar[i2] == rdg);
                    return wgt;
                };
                for (int cb = 0; cb < T.Bx; cb += 16384) {
                    uint32_t bestc = kKeyNull;
                    const int cend = min(T.Bx, cb + 16384);
                    int rd = 0, base = cb;
                    if (kWide) {   // two rounds per trip: straight-line code, the LDS round trips of one round hide behind the other's
                        for (; base + 64 < cend; base += 128, rd += 2) {
                            const int rdg = (cb >> 6) + rd;
                            if (is_weighted(rdg) || is_weighted(rdg + 1)) {   // rare: at most NW rounds of a scan carry weights
                                bestc = min(bestc, is_weighted(rdg) ? scan_round(std::true_type{}, base, rd) : scan_round(std::false_type{}, base, rd));
                                bestc = min(bestc, is_weighted(rdg + 1) ? scan_round(std::true_type{}, base + 64, rd + 1) : scan_round(std::false_type{}, base + 64, rd + 1));
                            } else {
                                const uint32_t k0 = scan_round(std::false_type{}, base, rd);
                                const uint32_t k1 = scan_round(std::false_type{}, base + 64, rd + 1);
                                bestc = min(bestc, min(k0, k1));
                            }
                        }
                    }
                    if (kWide || kTeam) {
                        for (; base < cend; base += 64, ++rd)
                            bestc = min(bestc, is_weighted((cb >> 6) + rd) ? scan_round(std::true_type{}, base, rd) : scan_round(std::false_type{}, base, rd));
                    } else {
                        // Round 5: the rounds between two weighted ones run in a loop of their own.  Asking every round
                        // "is it one of the NW weighted ones" was 11 of its 22 scalar instructions -- and the two-way
                        // body cost two register copies of the generator state and an address add per round on top of
                        // the 12 vector instructions a candidate round needs.  Same rounds, same order, same draws.
                        const int n_rd = (cend - cb + 63) >> 6, rd0 = cb >> 6;
                        while (rd < n_rd) {
                            int nxt = n_rd;   // the next weighted round of this chunk at or after rd
#pragma unroll
                            for (int i2 = 0; i2 < NW; ++i2) {
                                const int m = mr[i2] - rd0;
                                if (m >= rd) nxt = min(nxt, m);
                            }
                            for (; rd + 1 < nxt; rd += 2, base += 128) {   // two rounds share the address arithmetic and one v_min3_u32
                                const uint32_t k0 = scan_round(std::false_type{}, base, rd);
                                const uint32_t k1 = scan_round(std::false_type{}, base + 64, rd + 1);
                                bestc = min(bestc, min(k0, k1));
                            }
                            if (rd < nxt) { bestc = min(bestc, scan_round(std::false_type{}, base, rd)); ++rd; base += 64; }
                            if (rd < n_rd) { bestc = min(bestc, scan_round(std::true_type{}, base, rd)); ++rd; base += 64; }
                        }
                    }
                    if ((bestc >> 8) < (bestA >> 8)) { bestA = bestc; chunkA = cb; }   // strict: ties stay with the earlier round
                }
                if (!kTeam && holds) L.W[ai & 0xFFFFu] = (uint16_t)w_keep;
                key = bestA >> 8;   // (cost + bias) << 8 | tie: the key format of every other move type
                kmin = wave_umin(key);
                const unsigned long long bal = __ballot(key == kmin);
                win = __ffsll((long long)bal) - 1;  // ties inside the wave go to the lowest lane
                if ((int)(kmin >> 8) - kDBias <= 0) {   // will be accepted: the winner's move, wave-uniform
                    const uint32_t bw_ = (uint32_t)__builtin_amdgcn_readlane((int)bestA, win);
                    const uint32_t xs = (uint32_t)__builtin_amdgcn_readlane(chunkA, win) + ((bw_ & 255u) << 6) + (uint32_t)win;
                    const uint32_t rs = XR[xs];
                    const uint32_t ws = L.W[xs];
                    const int rts = L.RT[rs];
                    vw = xs | (rs << 16);
                    dV = wfld(ws, kWIncR) + wfldw(ws, kWIncL, lw) + dV_old + (kPriced ? (int)(signed char)(rts & 0xFF) : rts);
                    dObj = -g_old + (has_missing ? role_w2(c, vw, wl, wf) : 0) + (hbw ? bw_of(BW[xs], lead) : 0);
                }
                if (si == 0 || kmin < b_kmin) { b_kmin = kmin; b_win = win; b_p = p; b_k = k; b_uw = uw; b_vw = vw; b_dV = dV; b_dObj = dObj; }
            } else {
                // ---- phase B (EXCHANGE): every partner slot (q,j) for slot (p,k), 64 partitions per round ----
                const int nrp = lead ? 0 : 1;
                const int cnt_a_ru = cnt4(a, ro);
                const uint32_t wu = L.W[uw & 0xFFFFu];
                const int pl_u = kPriced ? price_lead(PR[uw & 0xFFFFu]) : 0;
                const int bwl_u = hbw ? (int)(BW[uw & 0xFFFFu] >> 16) : 0;
                // every lane draws; lane 0's value places the window when the topic has more than 512 partitions
                // (topics in global memory: drawn -- at the same place of the stream -- and its first partner words loaded by tournament_global)
                if (!kGlobalA) {
                    const int q_draw = (int)rnd24_wide(rng, (uint32_t)T.P);
                    q0 = x_windowed ? __builtin_amdgcn_readfirstlane(q_draw) : 0;
                }
                const int x_rounds = x_windowed ? 8 : x_rounds_full;
                auto x_round = [&](int rd, const Part<NW> &b, const Part<NW> &cb) {
                    const uint32_t tie0 = lcg24(rng) >> 8;
                    int qq; bool okq;
                    x_partner(rd, qq, okq);
                    okq = okq & (qq != p);
                    const bool u_in_b = in4(b, uw);
                    // independent of the partner slot j: what u would be worth in q, and q's replicas in u's rack
                    const int u_in_q_lead = role_w2(cb, uw, T.w00, T.w10), u_in_q_fol = role_w2(cb, uw, T.w01, T.w11);
                    const int cnt_b_ro = cnt4(b, ro);
#pragma unroll
                    for (int jj = 0; jj < NW; ++jj) {
                        if (jj >= T.RF) break;
                        const uint32_t v = b.w[jj];
                        const bool ok = okq & (v != uw) & !in4(a, v) & !u_in_b;
                        const int nrq = jj != 0;
                        int dObjx = role_w(T, c, v, nrp) + (jj == 0 ? u_in_q_lead : u_in_q_fol) - g_old - role_w(T, cb, v, nrq);
                        int dVx = 0, dPx = 0;
                        if (lead != (jj == 0)) {  // wave-uniform: exactly one of the two slots is a leader slot
                            const uint32_t wv = L.W[v & 0xFFFFu];
                            dVx += lead ? (wfld(wu, kWDecL) + wfld(wv, kWIncL)) : (wfld(wv, kWDecL) + wfld(wu, kWIncL));
                            if (kPriced) {  // the leader moves u -> v or v -> u
                                if (hbw) { const int dbl = (int)(BW[v & 0xFFFFu] >> 16) - bwl_u; dObjx += lead ? dbl : -dbl; }
                                const int plv = price_lead(PR[v & 0xFFFFu]);
                                dPx = lead ? ((wflag(wv, kWPinL) & plv) - (wflag(wu, kWPoutL) & pl_u))
                                           : ((wflag(wu, kWPinL) & pl_u) - (wflag(wv, kWPoutL) & plv));
                            }
                        }
                        const uint32_t rv = v >> 16;
                        if (rv != ro)
                            dVx += ddec(cnt_a_ru, T.prack_lo, T.prack_hi) + dinc(cnt4(a, rv), T.prack_lo, T.prack_hi) +
                                   ddec(cnt4(b, rv), T.prack_lo, T.prack_hi) + dinc(cnt_b_ro, T.prack_lo, T.prack_hi);
                        uint32_t keyx;
                        if (kPriced) keyx = ok ? make_key_tie_p(lam, S, dVx, dObjx, dPx, tie0 + (uint32_t)jj * 0x55u) : kKeyNull;
                        else keyx = ok ? make_key_tie(lam, S, dVx, dObjx, tie0 + (uint32_t)jj * 0x55u) : kKeyNull;
                        if (keyx < key) { key = keyx; vw = v; q = qq; j = jj; dV = dVx; dObj = dObjx; }
                    }
                };
                if constexpr (kGlobalA) {
#pragma unroll
                    for (int r2 = 0; r2 < XB; ++r2)
                        if (r2 < x_rounds) x_round(r2, xb[r2], xcb[r2]);
                    for (int rb = XB; rb < x_rounds; rb += XB) {   // (8 words per partition: the second half of the rounds)
                        Part<NW> yb[XB], ycb[XB];
#pragma unroll
                        for (int r2 = 0; r2 < XB; ++r2) { int qq; bool okq; const int qc = x_partner(rb + r2, qq, okq); yb[r2] = L.A[qc]; ycb[r2] = CUR[qc]; }
#pragma unroll
                        for (int r2 = 0; r2 < XB; ++r2)
                            if (rb + r2 < x_rounds) x_round(rb + r2, yb[r2], ycb[r2]);
                    }
                } else {
                    for (int rd = 0; rd < x_rounds; ++rd) {
                        int qq; bool okq;
                        const int qc = x_partner(rd, qq, okq);
                        const Part<NW> b = L.A[qc];
                        const Part<NW> cb = CUR[qc];
                        x_round(rd, b, cb);
                    }
                }
                kmin = wave_umin(key);
                const unsigned long long bal = __ballot(key == kmin);
                win = __ffsll((long long)bal) - 1;  // ties inside the wave go to the lowest lane
            }
            }   // (the two scan slots)
            if (type == 0) { kmin = b_kmin; win = b_win; p = b_p; k = b_k; uw = b_uw; vw = b_vw; dV = b_dV; dObj = b_dObj; }
        }
        if (!kTeam) {
            if (kmin == kKeyNull) continue;
            if ((int)(kmin >> 8) - kDBias > 0) continue;  // accept only non-worsening moves (cost under current lam)
        }
        bool mine = true;   // team: this wavefront's proposal is applied
        int *rec = TR + ((i & 1u) * 16 + (uint32_t)wave) * kTeamRec;
        if (kTeam) {
            // ---- the team's proposals meet: record = {acceptable, p, q, broker out, broker in, rack out, rack in, dV, dObj, applied} ----
            const bool ok = kmin != kKeyNull && (int)(kmin >> 8) - kDBias <= 0;
            if (lane == win) {
                const uint32_t ro_ = uw >> 16, rn_ = vw >> 16;
                const bool racks = ok && type == 0 && ro_ != rn_;   // only a REPLACE across racks reads and changes rack totals
                rec[0] = ok ? 1 : 0; rec[1] = p; rec[2] = type == 1 ? q : p;
                rec[3] = (int)(uw & 0xFFFFu); rec[4] = (int)(vw & 0xFFFFu);
                rec[5] = racks ? (int)ro_ : 0xFFFF; rec[6] = racks ? (int)rn_ : 0xFFFF;
                rec[7] = dV; rec[8] = dObj;
            }
            __syncthreads();
            bool clash = false;
            if (lane < wave) {   // lane l looks at wavefront l's record: only lower-numbered wavefronts can block this one
                const int *o = TR + ((i & 1u) * 16 + (uint32_t)lane) * kTeamRec;
                if (o[0]) {
                    const int mp = rec[1], mq = rec[2], mb0 = rec[3], mb1 = rec[4], mr0 = rec[5], mr1 = rec[6];
                    clash = (o[1] == mp) | (o[1] == mq) | (o[2] == mp) | (o[2] == mq) | (o[3] == mb0) | (o[3] == mb1) | (o[4] == mb0) | (o[4] == mb1);
                    if (o[5] != 0xFFFF && mr0 != 0xFFFF) clash |= (o[5] == mr0) | (o[5] == mr1) | (o[6] == mr0) | (o[6] == mr1);
                }
            }
            mine = ok && __ballot(clash) == 0ull;
        }

        if (lane == win && mine) {  // the winning lane applies its own proposal
            uint32_t *ap = reinterpret_cast<uint32_t *>(&L.A[p]);
            if (type == 0) {
                const uint32_t d = (k == 0) ? 0x10001u : 1u;
                L.C[uw & 0xFFFFu] -= d;
                L.C[vw & 0xFFFFu] += d;
                if (!kTeam || (uw >> 16) != (vw >> 16)) {   // (team: moves inside one rack do not own its total -- two of them may run at once)
                    L.K[uw >> 16] -= 1;
                    L.K[vw >> 16] += 1;
                }
                ap[k] = vw;
            } else if (type == 1) {
                uint32_t *bp = reinterpret_cast<uint32_t *>(&L.A[q]);
                if ((k == 0) != (j == 0)) {
                    const uint32_t lose = (k == 0) ? uw : vw, gain = (k == 0) ? vw : uw;
                    L.C[lose & 0xFFFFu] -= 0x10000u;
                    L.C[gain & 0xFFFFu] += 0x10000u;
                }
                ap[k] = vw;
                bp[j] = uw;
            } else {
                L.C[uw & 0xFFFFu] -= 0x10000u;
                L.C[vw & 0xFFFFu] += 0x10000u;
                ap[0] = vw;
                ap[k] = uw;
            }
        }
        if (mine) {   // band state of the two brokers whose counters may have changed: lane 0 the old broker, lane 1 the new one
            const uint32_t xo = (uint32_t)__builtin_amdgcn_readlane((int)uw, win) & 0xFFFFu, xn = (uint32_t)__builtin_amdgcn_readlane((int)vw, win) & 0xFFFFu;
            if (lane < 2) {
                const uint32_t xx = lane ? xn : xo;
                L.W[xx] = (uint16_t)band_fields(T, L.C[xx]);
            }
        }
        if (kGlobalA) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // the winner's stores before the next loads
        if (kTeam) {
            if (lane == 0) rec[9] = mine ? 1 : 0;
            __syncthreads();
            int dVs = 0, dOs = 0, na = 0;
            if (lane < n_waves) {
                const int *o = TR + ((i & 1u) * 16 + (uint32_t)lane) * kTeamRec;
                if (o[9]) { dVs = o[7]; dOs = o[8]; na = 1; }
            }
            V += wave_sum(dVs);
            obj += wave_sum(dOs);
            accepted += wave_sum(na);
        } else {
            V += __builtin_amdgcn_readlane(dV, win);
            obj += __builtin_amdgcn_readlane(dObj, win);
            accepted++;
        }
        if (V == 0 && obj > best_obj) { best_obj = obj; snapshot(T, L, ext, best, tid, nthr); }
    }

    // ---- end of launch: verify the incremental bookkeeping against a from-scratch recount ----
    if (kTeam) __syncthreads();
    recount(T, L, tid, nthr, krt);
    if (kTeam) __syncthreads();
    int V2, obj2;
    full_cost<NW, kTeam>(T, L, CUR, RSZ, tid, nthr, V2, obj2, hbw ? BW : nullptr, TS, wave, n_waves);
    if ((V2 != V || obj2 != obj) && tid == 0) atomicAdd(pl.drift, 1);
    if (!kGlobalA)
        for (int p = lane; p < T.P; p += 64) store_packed<NW>(state_packed, p, L.A[p]);
    if (tid == 0) {
        pl.restart_info[g * 4 + 0] = best_obj;
        pl.restart_info[g * 4 + 1] = V2;
        pl.restart_info[g * 4 + 2] = obj2;
        pl.restart_info[g * 4 + 3] = accepted;
    }
}

#ifndef KAO_WPE_SMALL
#define KAO_WPE_SMALL 6
#endif
#ifndef KAO_WPE_WIDE
#define KAO_WPE_WIDE 6
#endif
template <bool kGlobalA, bool kPriced, int NW, bool kWide> constexpr int search_min_waves() { return kGlobalA ? 1 : (kWide ? KAO_WPE_WIDE : KAO_WPE_SMALL); }
template <bool kGlobalA, bool kPriced, int NW, bool kWide>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(search_min_waves<kGlobalA, kPriced, NW, kWide>(), 8))) void k_search(SearchPools pl, SearchParams prm) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    search_body<kGlobalA, kPriced, NW, kWide, false>(smem, pl, prm);
}
// working assignment in LDS, current assignment from global memory / L2 (kCurG; ~4,900 .. 9,800 partitions: always wide)
template <bool kPriced, int NW>
__global__ __launch_bounds__(256) void k_search_curg(SearchPools pl, SearchParams prm) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    search_body<false, kPriced, NW, true, false, true>(smem, pl, prm);
}
// a team of up to 8 wavefronts per restart (topics in global memory; see search_body)
template <bool kPriced, int NW>
__global__ __launch_bounds__(NW == 8 ? 256 : 512) void k_team(SearchPools pl, SearchParams prm) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    search_body<true, kPriced, NW, true, true>(smem, pl, prm);
}

// ------------------------------------------------------------------------------------------------
// K-init (round 6): the hole filling of an initialising launch, one WORKGROUP per restart -- topics in global memory only
// ------------------------------------------------------------------------------------------------
// search_body fills the holes with ONE wavefront per restart: every hole scans all Bx brokers, 64 per round, and the holes are a chain
// (each insertion moves the counters the next one reads) -- 15,000 holes x 16 rounds = 70 ms for config 5 as one topic, on a quarter
// of the chip's SIMDs at best.  Here the rounds of a hole are dealt to the W wavefronts of the workgroup (round j to wavefront j mod W);
// every wavefront reduces its rounds to one record (key, lane, round, word), the records meet in LDS behind ONE workgroup barrier per hole
// and every wavefront takes the same minimum in the order of the single-wavefront scan: lowest key, then lowest lane, then -- inside a
// lane -- the earliest round.  Same holes, same order, same winners -- the replays (oracle/kao_port.c::ls_init) hold bit for bit; k_search / k_team
// then run with prm.init = 2 (state seeded and filled, everything else as in an initialising launch).
constexpr int kInitWaves = 16;
size_t init_lds_bytes(int maxBx, int maxR, bool priced, bool bw) {
    const size_t bx64 = (size_t)((maxBx + 63) & ~63), krt = (size_t)search_rack_tab(maxR);
    return krt * 4 + bx64 + (priced ? (bw ? 2 : 1) * bx64 * 4 + krt * 4 : 0) + bx64 * 4 + krt * 4 + (size_t)kInitWaves * krt * 4 + 3 * 8;
}
template <bool kPriced, int NW>
__global__ __launch_bounds__(64 * kInitWaves) void k_init(SearchPools pl, SearchParams prm, int per_block) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n_waves = __builtin_amdgcn_readfirstlane((int)(blockDim.x >> 6));
    const int tid = (int)threadIdx.x, nthr = (int)blockDim.x;
    const int2 bm = pl.block_map[blockIdx.x / (unsigned)per_block];
    const TopicDev *TD = pl.topics + bm.x;
    const int rho = bm.y + (int)(blockIdx.x % (unsigned)per_block);
    if (rho >= TD->n_restarts) return;   // (the whole workgroup)

    TopicRegs T;
    T.P = TD->P; T.RF = TD->RF; T.R = TD->R; T.m = TD->m; T.Bx = TD->Bx; T.magic = TD->magic;
    T.rep_lo = TD->rep_lo; T.rep_hi = TD->rep_hi; T.lead_lo = TD->lead_lo; T.lead_hi = TD->lead_hi;
    T.rack_lo = TD->rack_lo; T.rack_hi = TD->rack_hi; T.prack_lo = TD->prack_lo; T.prack_hi = TD->prack_hi;
    T.w00 = TD->w00; T.w01 = TD->w01; T.w10 = TD->w10; T.w11 = TD->w11;

    // LDS: [RSZ int[krt]] [XR u8[bx64]] ([PR u32[bx64]] [PG int[krt]] ([BW u32[bx64]])) [C u32[bx64]] [K int[krt]] [KW int[W][krt]] [BEST u64[3]]
    const int bx64 = (prm.maxBx + 63) & ~63, krt = search_rack_tab(prm.maxR);
    const bool hbw = kPriced && prm.bw != 0;
    const uint32_t inv = (uint32_t)krt - 1u;
    unsigned char *q = smem;
    int *RSZ = reinterpret_cast<int *>(q); q += krt * 4;
    uint8_t *XR = q; q += bx64;
    uint32_t *PR = reinterpret_cast<uint32_t *>(q); if (kPriced) q += bx64 * 4;
    int *PG = reinterpret_cast<int *>(q); if (kPriced) q += krt * 4;
    uint32_t *BW = reinterpret_cast<uint32_t *>(q); if (hbw) q += bx64 * 4;
    WaveLds<NW> L;
    L.C = reinterpret_cast<uint32_t *>(q); q += bx64 * 4;
    L.K = reinterpret_cast<int *>(q); q += krt * 4;
    L.W = nullptr; L.RT = nullptr;
    int *KW = reinterpret_cast<int *>(q); q += kInitWaves * krt * 4;   // every wavefront's own copy of the rack counts
    unsigned long long *BEST = reinterpret_cast<unsigned long long *>(q);   // the hole's winner over the wavefronts (LDS atomic min), three in rotation

    for (int r = tid; r < krt; r += nthr) {
        RSZ[r] = r < T.R ? pl.rsz_pool[TD->rsz_off + r] : 0;
        if (kPriced) PG[r] = r < T.R ? price_units(pl.price_pool[TD->price_off + 2 * TD->B + r], prm.obj_scale) : 0;
    }
    __syncthreads();
    for (int x = tid; x < ((T.Bx + 63) & ~63); x += nthr) {
        const uint32_t r = mulhi((uint32_t)x, T.magic);
        const bool valid = x < T.Bx && (int)((uint32_t)x - r * (uint32_t)T.m) < RSZ[r < (uint32_t)krt ? r : 0];
        XR[x] = valid ? (uint8_t)r : (uint8_t)inv;
        if (kPriced) {
            uint32_t pr = 0;
            if (valid) {
                const int32_t *pp = pl.price_pool + TD->price_off;
                const int b = pl.ext_pool[TD->ext_off + x];
                pr = ((uint32_t)price_units(pp[b], prm.obj_scale) & 0xFFFFu) | ((uint32_t)price_units(pp[TD->B + b], prm.obj_scale) << 16);
            }
            PR[x] = pr;
            if (hbw) BW[x] = (valid && TD->has_bw) ? pl.bw_pool[TD->bw_off + x] : 0u;
        }
    }
    const Part<NW> *CUR = reinterpret_cast<const Part<NW> *>(pl.cur_pool + TD->cur_off);
    L.A = reinterpret_cast<Part<NW> *>(pl.state_pool + TD->state_off) + (uint64_t)rho * T.P;
    const uint32_t slo = TD->seed_lo, shi = TD->seed_hi;
    const int S = prm.obj_scale;
    for (int p = tid; p < T.P; p += nthr) {   // surviving current replicas stay in their slots
        Part<NW> c = CUR[p];
#pragma unroll
        for (int k = 1; k < NW; ++k)
            if (k >= T.RF) c.w[k] = kNoneW;
        L.A[p] = c;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __syncthreads();
    recount(T, L, tid, nthr, krt);
    __syncthreads();
    // Who reads what during the fill: C[x] only the wavefront that scans x's round, K[r] every wavefront.  So the winner's C entry is
    // bumped by the wavefront that owns it, the rack counts live in one copy per wavefront, and a hole costs ONE workgroup barrier.
    if (n_waves > 1) {
        for (int r = lane; r < krt; r += 64) KW[wave * krt + r] = L.K[r];
        L.K = KW + wave * krt;
        if (tid < 3) BEST[tid] = ~0ull;
        __syncthreads();
    }

    const uint32_t *HL = pl.cur_pool + TD->hole_off;
    const int n_rounds = (T.Bx + 63) >> 6;
    int par = 0;   // which BEST this hole uses
    for (int pass = 0; pass < 2; ++pass) {   // leader holes of all partitions first, then follower holes
        const uint32_t n_holes = HL[pass], *hl = HL + 2 + (pass ? HL[0] : 0u);
        // The rows of 64 holes are fetched at once, one hole per lane, and handed out by v_readlane: a pass lists every partition once, so
        // no row of the batch is written before its turn.  (Fetching them one hole ahead left two dependent global round trips, list entry
        // then row, per hole: 1.5 us, the whole fill once the scan was spread over the workgroup.)
        for (uint32_t h0 = 0; h0 < n_holes; h0 += 64) {
          const uint32_t nb = min(64u, n_holes - h0);
          const uint32_t p_l = hl[h0 + ((uint32_t)lane < nb ? (uint32_t)lane : 0u)];
          const Part<NW> a_l = L.A[p_l], c_l = CUR[p_l];
          for (uint32_t hi = 0; hi < nb; ++hi) {
            const int p = __builtin_amdgcn_readlane((int)p_l, (int)hi);
            Part<NW> a, c;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                a.w[w] = (uint32_t)__builtin_amdgcn_readlane((int)a_l.w[w], (int)hi);
                c.w[w] = (uint32_t)__builtin_amdgcn_readlane((int)c_l.w[w], (int)hi);
            }
#pragma unroll
            for (int k = 0; k < NW; ++k) {
                if (k >= T.RF) break;
                if ((k == 0) != (pass == 0)) continue;
                if (a.w[k] != kNoneW) continue;   // uniform over the workgroup
                const uint32_t hmix = slo ^ fmix32(shi + (uint32_t)rho * 0x9E3779B1u + (uint32_t)(p * NW + k) * 0x27D4EB2Fu + 0x5BD1E995u + prm.gen * 0x632BE5ABu);
                const int wl = k == 0 ? T.w00 : T.w01, wf = k == 0 ? T.w10 : T.w11;
                uint32_t key = kKeyNull, xw_l = kNoneW;
                for (int j = wave; j < n_rounds; j += n_waves) {
                    const uint32_t x = (uint32_t)(j * 64 + lane);
                    const uint32_t r = XR[x];
                    const uint32_t xw = x | (r << 16);
                    const bool okx = (r != inv) & !in4(a, xw);
                    const uint32_t cn = L.C[x];
                    int dV = dinc((int)(cn & 0xFFFFu), T.rep_lo, T.rep_hi) + dinc(L.K[r], T.rack_lo, T.rack_hi) +
                             dinc(cnt4(a, r), T.prack_lo, T.prack_hi);
                    if (k == 0) dV += dinc((int)(cn >> 16), T.lead_lo, T.lead_hi);
                    const uint32_t tie = fmix32(hmix + x * 0x165667B1u) >> 24;
                    uint32_t keyx;
                    if (kPriced) {
                        const uint32_t prx = PR[x];
                        int dP = p_in((int)(cn & 0xFFFFu), T.rep_lo, T.rep_hi, price_rep(prx)) + p_in(L.K[r], T.rack_lo, T.rack_hi, PG[r]);
                        if (k == 0) dP += p_in((int)(cn >> 16), T.lead_lo, T.lead_hi, price_lead(prx));
                        keyx = okx ? make_key_tie_p(prm.lam_max, S, dV, role_w2(c, xw, wl, wf) + (hbw ? bw_of(BW[x], k == 0) : 0), dP, tie) : kKeyNull;
                    }
                    else keyx = okx ? make_key_tie(prm.lam_max, S, dV, role_w2(c, xw, wl, wf), tie) : kKeyNull;
                    if (keyx < key) { key = keyx; xw_l = xw; }   // (equal keys: the earlier round stays)
                }
                // this wavefront's winner: lowest key, then lowest lane (then, inside the lane, the earliest round: above)
                uint32_t kmin = wave_umin(key);
                const unsigned long long bal = __ballot(key == kmin);
                const int l_win = __ffsll((long long)bal) - 1;
                uint32_t xw_win = (uint32_t)__builtin_amdgcn_readlane((int)xw_l, l_win);
                if (n_waves > 1) {
                    // across the wavefronts the same order, (key, lane, round), packed with the rack into one 64-bit word: LDS atomic min.
                    // Three words in rotation: the one two holes ahead is reset behind this hole's barrier (everybody has read it before).
                    if (lane == 0) atomicMin(&BEST[par], ((unsigned long long)kmin << 32) | ((unsigned long long)l_win << 24) | ((unsigned long long)((xw_win & 0xFFFFu) >> 6) << 8) | (xw_win >> 16 & 0xFFu));
                    // LDS traffic only is ordered here: __syncthreads() would also wait for the winner's global store of the previous hole
                    // (vmcnt(0): a round trip to L2 per hole, ~1 us -- it was most of the fill); nobody reads those rows before the pass ends
                    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                    const unsigned long long bw = BEST[par];
                    const int nxt = par == 2 ? 0 : par + 1;
                    if (tid == 0) BEST[nxt == 2 ? 0 : nxt + 1] = ~0ull;
                    par = nxt;
                    const uint32_t lo = (uint32_t)bw;
                    xw_win = (((lo >> 8) & 0xFFFFu) << 6 | (lo >> 24)) | ((lo & 0xFFu) << 16);
                }
                a.w[k] = xw_win;
                if (lane == 0) {
                    const int owner = (int)(((xw_win & 0xFFFFu) >> 6) & (uint32_t)(n_waves - 1));   // (W is a power of two)
                    if (wave == owner) {
                        reinterpret_cast<uint32_t *>(&L.A[p])[k] = xw_win;
                        L.C[xw_win & 0xFFFFu] += (k == 0) ? 0x10001u : 1u;
                    }
                    L.K[xw_win >> 16] += 1;   // own copy
                }
            }
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");   // pass 1 reads the rows pass 0 wrote
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// K-eval
// ------------------------------------------------------------------------------------------------
// NE = replica slots handled per partition: 4 (RF and current RF <= 4) or 8.
// kCoop = false: one wavefront per candidate (4 candidates in flight per workgroup) -- batches that fill the device.
// kCoop = true : the WHOLE workgroup evaluates one candidate, its four wavefronts striding the partitions over one shared set
//                of LDS counters -- few large candidates (a 30,000-partition topic has 256 restarts; KAO-CX scores <= 513
//                realisations): one wavefront per candidate left 3 of 4 SIMDs idle and took 469 dependent trips per candidate.
//                All sums are integers, so the split changes no result.
// RFT > 0 (round 6): every topic of the launch has replication factor RFT -- the slot loops run RFT times without the `k >= RF` guards and the
//                C7 compare square is RFT x RFT instead of NE x NE (RF 3 in four slots: 9 of 16); RFT = 0: RF is read per topic.
template <int NE, bool kCoop, int RFT = 0>
__global__ __launch_bounds__(256) void k_eval(EvalPools pl) {
    constexpr int RFE = RFT ? RFT : NE;      // slots the loops visit
    constexpr bool kLds = RFT > 0;           // the RF-uniform instantiation is launched only with the current assignment staged in LDS (launch_eval)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
    unsigned long long *wave_key = reinterpret_cast<unsigned long long *>(smem_all);  // [kWaves], 32 B
    unsigned char *smem = smem_all + 32;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int4 bm = pl.block_map[blockIdx.x];
    const TopicDev *TD = pl.topics + bm.x;
    const int B = TD->B, R = TD->R, P = TD->P, RF = RFT ? RFT : TD->RF, rf_cur = TD->rf_cur;
    const int rep_lo = TD->rep_lo, rep_hi = TD->rep_hi, lead_lo = TD->lead_lo, lead_hi = TD->lead_hi;
    const int rack_lo = TD->rack_lo, rack_hi = TD->rack_hi, prack_lo = TD->prack_lo, prack_hi = TD->prack_hi;
    const int w00 = TD->w00, w01 = TD->w01, w10 = TD->w10, w11 = TD->w11;
    const bool cur_lds = kLds || pl.cur_in_lds != 0;
    // C7 of a partition whose three replicas sit on one / two / three racks (RFT == 3)
    const int c7_one = band(3, prack_lo, prack_hi) + (R - 1) * prack_lo;
    const int c7_two = band(2, prack_lo, prack_hi) + band(1, prack_lo, prack_hi) + (R - 2) * prack_lo;
    const int c7_three = 3 * band(1, prack_lo, prack_hi) + (R - 3) * prack_lo;

    // ---- LDS carve: [wave_key 32 B] [RACK u8[maxB~]] [CURD u16[maxP][NE]] then per wave [C u32[maxB~]] [K int[256]]
    const int r_bytes = (pl.maxB + 15) & ~15;
    const int d_bytes = cur_lds ? pl.maxP * NE * 2 : 0;  // huge topics read the current assignment from global memory
    const int c_bytes = (pl.maxB * 4 + 15) & ~15;
    uint8_t *RACK = smem;
    uint16_t *CURD = reinterpret_cast<uint16_t *>(smem + r_bytes);
    unsigned char *wb = smem + r_bytes + ((d_bytes + 15) & ~15) + (kCoop ? 0 : wave) * (c_bytes + kRackTab * 4);
    int *red = reinterpret_cast<int *>(smem + r_bytes + ((d_bytes + 15) & ~15) + kWaves * (c_bytes + kRackTab * 4));   // [kWaves][8] (kCoop)
    const int tid = kCoop ? (int)threadIdx.x : lane, tstride = kCoop ? 256 : 64;
    uint32_t *C = reinterpret_cast<uint32_t *>(wb);
    int *K = reinterpret_cast<int *>(wb + c_bytes);

    // ---- stage the broker->rack table and the current assignment (padded to NE slots with 0xFFFF) ----
    for (int b = threadIdx.x; b < B; b += 256) RACK[b] = pl.rackof_pool[TD->rackof_off + b];
    const uint16_t *curd = pl.curd_pool + TD->curd_off;
    const uint32_t *bwd = TD->has_bw ? pl.bwd_pool + TD->bwd_off : nullptr;   // broker weights, dense index (global memory / L2)
    if (cur_lds)
        for (int i = threadIdx.x; i < P * NE; i += 256) {
            const int p = i / NE, k = i - p * NE;
            CURD[i] = k < rf_cur ? curd[(size_t)p * rf_cur + k] : (uint16_t)0xFFFFu;
        }
    __syncthreads();

    unsigned long long my_key = ~0ull;
    const int nB4 = (B + 3) >> 2;  // counters zeroed 16 bytes per lane per store (C is 16-byte aligned and padded)
    // Rack counters: the 64 lanes of a wavefront hit only R addresses, so one LDS atomic per replica would serialise (10 racks:
    // ~5 lanes per address).  They are privatised per 16-lane row -- 4 copies inside the same 256-entry table when R <= 64 --
    // added without return value, and the band rows C6 are evaluated from the totals in one pass at the end.
    const int KR = (R + 15) & ~15;
    const int kcopy = 4 * KR <= kRackTab ? (lane >> 4) * KR : 0;
    const bool k4 = 4 * KR <= kRackTab;
    const bool big = !kLds && P * RF > 65535;   // only then can a 16-bit per-broker counter overflow (the RF-3 instantiation runs with the current assignment in LDS: at most 20,480 partitions x 3)
    for (int ci = bm.y + (kCoop ? 0 : wave); ci < bm.y + bm.z; ci += (kCoop ? 1 : kWaves)) {
        const uint16_t *cand = pl.cand + TD->best_off + (uint64_t)ci * P * RF;
        for (int b4 = tid; b4 < nB4; b4 += tstride) reinterpret_cast<uint4 *>(C)[b4] = make_uint4(0, 0, 0, 0);
        for (int r = tid; r < (k4 ? 4 * KR : KR); r += tstride) K[r] = 0;      // (only the entries the atomics below and the C6 pass touch)
        if (kCoop) __syncthreads();
        // Broker band violations are accumulated from the value each LDS atomic RETURNS: adding a replica to a
        // broker whose count was c changes band(c) by (c >= hi) - (c < lo), and sum_b band(0) = B*lo, so
        // no pass over all brokers is needed.  Packed partial sums: low half = #(old >= hi), high = #(old < lo).
        int obj = 0;
        uint32_t s12 = 0;  // v1 | v2 << 16
        // C3 / C4 are COUNTS of lanes (old count at or above the upper end, below the lower end): each is a compare into a scalar pair and a
        // population count, accumulated in scalar registers -- no per-lane sum, no wavefront reduction (round 6, last: they were two of the six
        // words of wave_sum6 and five vector instructions per slot)
        int n3hi = 0, n3lo = 0, n4hi = 0, n4lo = 0;
        uint32_t s57 = 0;  // v5 | v7 << 16
        bool ovf = false;
        for (int p = tid; p < P; p += tstride) {
            const uint16_t *ap = cand + (size_t)p * RF;  // a wavefront reads 64*RF consecutive u16: coalesced
            uint32_t bk[NE], rk[NE], ck[NE];
            uint32_t cw[NE / 2];   // the partition's current replicas, two u16 per word: one ds_read_b64 / b128 when staged in LDS
            if (cur_lds) {
                if (NE == 4) { const uint2 v = reinterpret_cast<const uint2 *>(CURD)[p]; cw[0] = v.x; cw[1] = v.y; }
                else { const uint4 v = reinterpret_cast<const uint4 *>(CURD)[p]; cw[0] = v.x; cw[1] = v.y; cw[2 % (NE / 2)] = v.z; cw[3 % (NE / 2)] = v.w; }
            }
#pragma unroll
            for (int k = 0; k < NE; ++k) {
                bk[k] = k < RF ? (uint32_t)ap[k < RF ? k : 0] : 0xFFFFu;
                rk[k] = 0xFFu;
                ck[k] = cur_lds ? ((k & 1) ? cw[k >> 1] >> 16 : cw[k >> 1] & 0xFFFFu)
                                      : (k < rf_cur ? (uint32_t)curd[(size_t)p * rf_cur + (k < rf_cur ? k : 0)] : 0xFFFFu);
            }
            int missing = 0;
            // One slot of the trip (KAO_EVAL_SLOT).  VALID is the lane's own "this slot holds a broker", or the literal true when every lane of
            // the trip has one (the usual case, tested once per trip with a ballot): the body then has no per-lane branch and the counts sit in
            // wave-uniform control flow.  Otherwise the counts are taken where the lanes have met again -- the scalar accumulators live in every
            // lane's copy of the loop state, and a lane that sat out a slot would miss its counts (lane 0, whose copy is read in the end, is in
            // every trip: partitions ascend with the lane).
#define KAO_EVAL_SLOT(VALID) do { \
                uint32_t oc = 0; \
                if (!(VALID)) ++missing; \
                else { \
                    rk[k] = RACK[b]; \
                    oc = atomicAdd(&C[b], k == 0 ? 0x10001u : 1u); \
                    __hip_atomic_fetch_add(&K[kcopy + rk[k]], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); \
                    if (big) ovf |= (oc & 0xFFFFu) == 0xFFFFu; \
                    bool fol = false, dup = false; \
                    _Pragma("unroll") for (int j = 1; j < NE; ++j) fol |= ck[j] == b; \
                    _Pragma("unroll") for (int j = 0; j < k; ++j) dup |= bk[j] == b; \
                    obj += (ck[0] == b) ? (k == 0 ? w00 : w01) : (fol ? (k == 0 ? w10 : w11) : 0); \
                    if (bwd) { const uint32_t bw = bwd[b]; obj += (int)(bw & 0xFFFFu) + (k == 0 ? (int)(bw >> 16) : 0); } \
                    s57 += (uint32_t)dup;  /* C5: f+l <= 1 (an earlier slot holds the same broker) */ \
                } \
                const int cr = (int)(oc & 0xFFFFu); \
                n3hi += wave_count((VALID) & (cr >= rep_hi)); n3lo += wave_count((VALID) & (cr < rep_lo));         /* C3 */ \
                if (k == 0) { \
                    const int cl = (int)(oc >> 16); \
                    n4hi += wave_count((VALID) & (cl >= lead_hi)); n4lo += wave_count((VALID) & (cl < lead_lo));   /* C4 */ \
                } \
            } while (0)
            bool any_empty = false;
#pragma unroll
            for (int k = 0; k < RFE; ++k) {
                if (!RFT && k >= RF) break;
                any_empty |= bk[k] >= (uint32_t)B;
            }
            const bool trip_full = __ballot(any_empty) == 0ull;      // wave-uniform: no lane of this trip has an empty / out-of-range slot
            if (trip_full) {
#pragma unroll
                for (int k = 0; k < RFE; ++k) {
                    if (!RFT && k >= RF) break;
                    const uint32_t b = bk[k];
                    KAO_EVAL_SLOT(true);
                }
            } else {
#pragma unroll
                for (int k = 0; k < RFE; ++k) {
                    if (!RFT && k >= RF) break;
                    const uint32_t b = bk[k];
                    const bool valid = b < (uint32_t)B;
                    KAO_EVAL_SLOT(valid);
                }
            }
#undef KAO_EVAL_SLOT
            s12 += (uint32_t)missing + ((uint32_t)(bk[0] >= (uint32_t)B) << 16);  // C1: sum_b (f+l) = RF ; C2: exactly one leader
            // C7: replicas per partition per rack, over all R racks (each rack counted at its first slot)
            if (RFT == 3 && trip_full) {
                // three filled slots (no lane of this trip has an empty one: wave-uniform) fall on one, two or three racks -- the row's value
                // in each case is a constant of the topic (c7_one / c7_two / c7_three: the sums the general loop below would form)
                const bool e01 = rk[0] == rk[1], e02 = rk[0] == rk[2], e12 = rk[1] == rk[2];
                s57 += (uint32_t)((e01 & e12) ? c7_one : ((e01 | e02 | e12) ? c7_two : c7_three)) << 16;
            } else {
                int touched = 0, s7 = 0;
#pragma unroll
                for (int k = 0; k < RFE; ++k) {
                    bool first = rk[k] != 0xFFu;
                    int cnt = 0;
#pragma unroll
                    for (int j = 0; j < RFE; ++j) { cnt += (int)(rk[j] == rk[k]); if (j < k) first &= rk[j] != rk[k]; }
                    if (first) { s7 += band(cnt, prack_lo, prack_hi); touched++; }
                }
                s57 += (uint32_t)(s7 + (R - touched) * prack_lo) << 16;
            }
        }
        // C6 from the rack totals (the wavefront's own LDS operations complete in order: no barrier needed; the cooperating
        // wavefronts of kCoop meet at one)
        if (kCoop) __syncthreads();
        int s6 = 0;
        for (int r = tid; r < R; r += tstride) {
            const int tot = k4 ? K[r] + K[KR + r] + K[2 * KR + r] + K[3 * KR + r] : K[r];
            s6 += band(tot, rack_lo, rack_hi);
        }
        if (big && __ballot(ovf) != 0ull && pl.overflow && lane == 0) atomicOr(pl.overflow, 1);
        int v1, v2, v3, v4, v5, v6, v7;   // (v3, v4 without their constants B * lo until the partial sums have met)
        v3 = __builtin_amdgcn_readfirstlane(n3hi - n3lo); v4 = __builtin_amdgcn_readfirstlane(n4hi - n4lo);      // lane 0's copy (see above): wave-uniform from here on
        if (P * RF <= 32767) {  // packed halves cannot carry: every count is at most P*RF -- the four words are summed in one go (wave_sum4)
            int t[4] = {obj, (int)s12, (int)s57, s6};
            wave_sum4(t);
            const uint32_t t12 = (uint32_t)t[1], t57 = (uint32_t)t[2];
            obj = t[0]; v6 = t[3];
            v1 = (int)(t12 & 0xFFFFu); v2 = (int)(t12 >> 16);
            v5 = (int)(t57 & 0xFFFFu); v7 = (int)(t57 >> 16);
        } else {  // huge topic: per-lane halves still fit 16 bits, the wavefront totals do not -> sum them unpacked
            obj = wave_sum(obj);
            v1 = wave_sum((int)(s12 & 0xFFFFu)); v2 = wave_sum((int)(s12 >> 16));
            v5 = wave_sum((int)(s57 & 0xFFFFu)); v7 = wave_sum((int)(s57 >> 16));
            v6 = wave_sum(s6);
        }
        if (kCoop) {   // the four wavefronts' partial sums meet in LDS; every wavefront reads the totals
            if (lane == 0) { int *q = red + wave * 8; q[0] = obj; q[1] = v1; q[2] = v2; q[3] = v3; q[4] = v4; q[5] = v5; q[6] = v6; q[7] = v7; }
            __syncthreads();
            obj = v1 = v2 = v3 = v4 = v5 = v6 = v7 = 0;
            for (int w = 0; w < kWaves; ++w) {
                const int *q = red + w * 8;
                obj += q[0]; v1 += q[1]; v2 += q[2]; v3 += q[3]; v4 += q[4]; v5 += q[5]; v6 += q[6]; v7 += q[7];
            }
            __syncthreads();   // the counters and `red` are reused by the next candidate
        }
        v3 += B * rep_lo; v4 += B * lead_lo;
        const int v0 = v1 + v2 + v3 + v4 + v5 + v6 + v7;
        const int out = bm.w + (ci - bm.y);
        if (lane == 0 && (!kCoop || wave == 0)) {
            if (pl.objective) pl.objective[out] = obj;
            if (pl.violations) {
                int4 *vo = reinterpret_cast<int4 *>(pl.violations + (size_t)out * 8);
                vo[0] = make_int4(v0, v1, v2, v3);
                vo[1] = make_int4(v4, v5, v6, v7);
            }
        }
        const unsigned long long key = ((unsigned long long)min(v0, 0xFFFFF) << 44) |
                                       ((unsigned long long)(kObjCap - (uint32_t)min(obj, (int)kObjCap)) << 20) |
                                       (unsigned long long)(ci & 0xFFFFF);
        my_key = key < my_key ? key : my_key;
    }
    if (pl.best_key) {  // workgroup reduce of the wave-uniform keys, one atomicMin per workgroup per topic
        if (lane == 0) wave_key[wave] = my_key;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long k = wave_key[0];
            for (int w = 1; w < kWaves; ++w) k = wave_key[w] < k ? wave_key[w] : k;
            if (k != ~0ull) atomicMin(pl.best_key + bm.x, k);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K-gather: winners -> contiguous read-back buffers
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_gather(const TopicDev *topics, const unsigned long long *keys, const uint16_t *best_pool,
                                               const int32_t *viol, uint16_t *win_assign, int32_t *win_viol) {
    const TopicDev *TD = topics + blockIdx.x;
    const unsigned long long key = keys[blockIdx.x];
    if (key == ~0ull) return;
    const int rho = (int)(key & 0xFFFFFull);
    if (rho == (int)kExternalRestart) {  // the topic's best came from another GPU (kao_solve_multi): win_assign already holds it
        if (threadIdx.x < 8) win_viol[blockIdx.x * 8 + threadIdx.x] = 0;  // only feasible assignments are exchanged
        return;
    }
    const int n = TD->P * TD->RF;
    const uint16_t *src = best_pool + TD->best_off + (uint64_t)rho * n;
    uint16_t *dst = win_assign + TD->win_off;
    for (int i = threadIdx.x; i < n; i += 64) dst[i] = src[i];
    if (threadIdx.x < 8) win_viol[blockIdx.x * 8 + threadIdx.x] = viol[(size_t)(TD->restart_base + rho) * 8 + threadIdx.x];
}

// After the min-allreduce of the packed best keys across GPUs (kao_solve_multi, replicated topics): where another GPU's key
// beats the local one, adopt it with the reserved restart id kExternalRestart (its assignment arrives by broadcast).
__global__ __launch_bounds__(64) void k_adopt_global(unsigned long long *keys, const unsigned long long *glob, int n) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const unsigned long long g = glob[i];
    if (g < keys[i] && (g >> 44) == 0) keys[i] = g | (unsigned long long)kExternalRestart;
}

// ------------------------------------------------------------------------------------------------
// K-canon: canonical tie-break among equal-objective feasible assignments (kao_canonicalize)
// ------------------------------------------------------------------------------------------------
// Scanning partitions and slots in order, every NEWLY placed replica (its broker is not a current replica of the
// partition) moves to the lowest DENSE broker index that keeps the assignment feasible; repeated to a fixpoint.
// Such a move never changes the objective (neither broker carries weight on that partition) and, the state being
// feasible, it stays feasible iff the move's violation delta is 0 -- so this is the REPLACE scan of k_search with
// "delta == 0" as the filter and the dense index as the key.  One wavefront; the assignment and current-assignment
// words stay in global memory (any topic size); broker / rack tables in LDS.  status = {input feasible, #moves}.
template <int NW>
__global__ __launch_bounds__(64) void k_canon(const TopicDev *TD, const Part<NW> *cur_words, const uint16_t *ext, const int32_t *rsz,
                                              Part<NW> *A, int maxBx, int32_t *status) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    TopicRegs T;
    T.P = TD->P; T.RF = TD->RF; T.R = TD->R; T.m = TD->m; T.Bx = TD->Bx; T.magic = TD->magic;
    T.rep_lo = TD->rep_lo; T.rep_hi = TD->rep_hi; T.lead_lo = TD->lead_lo; T.lead_hi = TD->lead_hi;
    T.rack_lo = TD->rack_lo; T.rack_hi = TD->rack_hi; T.prack_lo = TD->prack_lo; T.prack_hi = TD->prack_hi;
    T.w00 = TD->w00; T.w01 = TD->w01; T.w10 = TD->w10; T.w11 = TD->w11;
    const int bx64 = (maxBx + 63) & ~63;
    int *RSZ = reinterpret_cast<int *>(smem);
    uint8_t *XR = smem + kRackTab * 4;
    WaveLds<NW> L;
    L.A = A;
    L.C = reinterpret_cast<uint32_t *>(smem + kRackTab * 4 + bx64);
    L.K = reinterpret_cast<int *>(smem + kRackTab * 4 + bx64 + bx64 * 4);
    L.RT = L.K;  // unused here
    L.W = reinterpret_cast<uint16_t *>(L.C);   // unused here
    for (int r = lane; r < kRackTab; r += 64) RSZ[r] = r < T.R ? rsz[r] : 0;
    __syncthreads();
    for (int x = lane; x < ((T.Bx + 63) & ~63); x += 64) {
        const uint32_t r = mulhi((uint32_t)x, T.magic);
        XR[x] = (x < T.Bx && (int)((uint32_t)x - r * (uint32_t)T.m) < RSZ[r < (uint32_t)kRackTab ? r : 0]) ? (uint8_t)r : (uint8_t)0xFF;
    }
    __syncthreads();
    recount(T, L, lane, 64, kRackTab);
    int V, obj;
    full_cost(T, L, cur_words, RSZ, lane, 64, V, obj);
    if (V != 0) {  // only feasible assignments are polished
        if (lane == 0) { status[0] = 0; status[1] = 0; }
        return;
    }
    int moves = 0;
    bool changed = true;
    while (changed) {
        changed = false;
        for (int pbase = 0; pbase < T.P; pbase += 64) {
            bool has_new = false;
            if (pbase + lane < T.P) {
                const Part<NW> al = L.A[pbase + lane];
                const Part<NW> cl = cur_words[pbase + lane];
#pragma unroll
                for (int k = 0; k < NW; ++k) has_new |= (k < T.RF) & !in4(cl, al.w[k]);
            }
            unsigned long long todo = __ballot(has_new);
            while (todo) {
                const int p = pbase + __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                Part<NW> a = L.A[p];
                const Part<NW> c = cur_words[p];
#pragma unroll
                for (int k = 0; k < NW; ++k) {
                    if (k >= T.RF) break;
                    const uint32_t uw = a.w[k];
                    if (in4(c, uw)) continue;  // a retained current replica stays where it is (wave-uniform)
                    const uint32_t old_dense = ext[uw & 0xFFFFu];
                    const uint32_t ro = uw >> 16;
                    const bool lead = k == 0;
                    const uint32_t co = L.C[uw & 0xFFFFu];
                    int dV_old = ddec((int)(co & 0xFFFFu), T.rep_lo, T.rep_hi);
                    if (lead) dV_old += ddec((int)(co >> 16), T.lead_lo, T.lead_hi);
                    const int dV_rack_old = ddec(L.K[ro], T.rack_lo, T.rack_hi) + ddec(cnt4(a, ro), T.prack_lo, T.prack_hi);
                    uint32_t key = kKeyNull;
                    for (int base = 0; base < T.Bx; base += 64) {
                        const uint32_t x = (uint32_t)(base + lane);
                        const uint32_t r = XR[x];
                        const uint32_t xw = x | (r << 16);
                        bool ok = (r != 0xFFu) && !in4(a, xw) && !in4(c, xw);
                        const uint32_t dense = ok ? (uint32_t)ext[x] : 0xFFFFu;
                        ok = ok & (dense < old_dense);
                        const uint32_t cn = L.C[x];
                        int dV = dV_old + dinc((int)(cn & 0xFFFFu), T.rep_lo, T.rep_hi);
                        if (lead) dV += dinc((int)(cn >> 16), T.lead_lo, T.lead_hi);
                        if (r != ro) dV += dV_rack_old + dinc(L.K[r & 255u], T.rack_lo, T.rack_hi) + dinc(cnt4(a, r), T.prack_lo, T.prack_hi);
                        const uint32_t kx = (ok & (dV == 0)) ? ((dense << 16) | x) : kKeyNull;
                        key = min(key, kx);
                    }
                    const uint32_t kmin = wave_umin(key);
                    if (kmin == kKeyNull) continue;
                    const uint32_t xn = kmin & 0xFFFFu;
                    const uint32_t rn = XR[xn];
                    const uint32_t xw_new = xn | (rn << 16);
                    a.w[k] = xw_new;
                    if (lane == 0) {
                        const uint32_t d = lead ? 0x10001u : 1u;
                        reinterpret_cast<uint32_t *>(&L.A[p])[k] = xw_new;
                        L.C[uw & 0xFFFFu] -= d;
                        L.C[xn] += d;
                        L.K[ro] -= 1;
                        L.K[rn] += 1;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                    changed = true;
                    ++moves;
                }
            }
        }
    }
    if (lane == 0) { status[0] = 1; status[1] = moves; }
}

// ------------------------------------------------------------------------------------------------
// launch wrappers
// ------------------------------------------------------------------------------------------------
size_t search_lds_bytes(int maxP, int maxBx, int waves, bool global_a, bool priced, int nw, bool bw, int maxR, int team, bool cur_global) {
    const size_t a = global_a ? 0 : (size_t)maxP * 4 * (size_t)nw, bx64 = ((size_t)maxBx + 63) & ~(size_t)63, krt = (size_t)search_rack_tab(maxR);
    const size_t shared = (cur_global ? 0 : a) + krt * 4 + bx64 + (priced ? (bw ? 2 : 1) * bx64 * 4 + krt * 4 : 0);
    if (team > 0)   // one set of counters / band states / rack totals, one RT per wavefront, the proposal records, the partial sums
        return shared + bx64 * 6 + krt * 4 + (size_t)team * krt * 4 + 2 * 16 * kTeamRec * 4 + 16 * 2 * 4;
    return shared + (size_t)waves * (a + bx64 * 6 + krt * 8);
}
size_t eval_lds_bytes(int maxP, int maxB, bool cur_in_lds, int ne) {
    const size_t r = ((size_t)maxB + 15) & ~(size_t)15, d = cur_in_lds ? ((size_t)maxP * 2 * (size_t)ne + 15) & ~(size_t)15 : 0;
    const size_t c = ((size_t)maxB * 4 + 15) & ~(size_t)15;
    return 32 + r + d + kWaves * (c + kRackTab * 4) + kWaves * 8 * 4;   // (+ the partial sums of the cooperative mode)
}

// largest dynamic-LDS size each kernel has been enabled for, per device (function attributes are per device)
static int g_attr_eval_dev[kAttrDevices] = {0};

template <bool kGlobalA, bool kPriced, int NW, bool kWide>
static void launch_search_t(const SearchPools &pools, const SearchParams &prm, int n_blocks, int waves, size_t lds, int &attr, hipStream_t st) {
    if ((int)lds > attr) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_search<kGlobalA, kPriced, NW, kWide>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((k_search<kGlobalA, kPriced, NW, kWide>), dim3(n_blocks), dim3(64 * waves), lds, st, pools, prm);
}
template <bool kGlobalA, bool kPriced, int NW>
static void launch_search_w(const SearchPools &pools, const SearchParams &prm, int n_blocks, int waves, size_t lds, int &attr, bool wide, hipStream_t st) {
    if (wide || kGlobalA) launch_search_t<kGlobalA, kPriced, NW, true>(pools, prm, n_blocks, waves, lds, attr, st);   // (topics in global memory are always wide)
    else launch_search_t<kGlobalA, kPriced, NW, kGlobalA>(pools, prm, n_blocks, waves, lds, attr, st);
}

template <bool kPriced, int NW>
static void launch_team_t(const SearchPools &pools, const SearchParams &prm, int n_blocks, int team, size_t lds, int &attr, hipStream_t st) {
    if ((int)lds > attr) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_team<kPriced, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((k_team<kPriced, NW>), dim3(n_blocks), dim3(64 * team), lds, st, pools, prm);
}

void launch_search(const SearchPools &pools, const SearchParams &prm, int n_blocks, int waves, bool global_a, bool priced, int nw, void *stream, int team) {
    const bool curg = prm.cur_global != 0 && !global_a;
    const size_t lds = search_lds_bytes(prm.maxP, prm.maxBx, waves, global_a, priced, nw, prm.bw != 0, prm.maxR, team, curg);
    // largest dynamic-LDS size each of the instantiations has been enabled for, per device
    static int attr[kAttrDevices][24] = {{0}};
    const bool wide = prm.wide != 0;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (curg) {   // working assignment in LDS, current assignment from L2
        int &ca = attr[attr_slot()][20 + (priced ? 2 : 0) + (nw == 8 ? 1 : 0)];
        const void *fn = nw == 8 ? (priced ? reinterpret_cast<const void *>(k_search_curg<true, 8>) : reinterpret_cast<const void *>(k_search_curg<false, 8>))
                                 : (priced ? reinterpret_cast<const void *>(k_search_curg<true, 4>) : reinterpret_cast<const void *>(k_search_curg<false, 4>));
        if ((int)lds > ca) { (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); ca = (int)lds; }
        if (nw == 8) { if (priced) hipLaunchKernelGGL((k_search_curg<true, 8>), dim3(n_blocks), dim3(64 * waves), lds, st, pools, prm); else hipLaunchKernelGGL((k_search_curg<false, 8>), dim3(n_blocks), dim3(64 * waves), lds, st, pools, prm); }
        else { if (priced) hipLaunchKernelGGL((k_search_curg<true, 4>), dim3(n_blocks), dim3(64 * waves), lds, st, pools, prm); else hipLaunchKernelGGL((k_search_curg<false, 4>), dim3(n_blocks), dim3(64 * waves), lds, st, pools, prm); }
        return;
    }
    if (team > 0) {   // one block per restart, `team` wavefronts each (topics in global memory only)
        int &ta = attr[attr_slot()][16 + (priced ? 2 : 0) + (nw == 8 ? 1 : 0)];
        if (nw == 8) { if (priced) launch_team_t<true, 8>(pools, prm, n_blocks, team, lds, ta, st); else launch_team_t<false, 8>(pools, prm, n_blocks, team, lds, ta, st); }
        else { if (priced) launch_team_t<true, 4>(pools, prm, n_blocks, team, lds, ta, st); else launch_team_t<false, 4>(pools, prm, n_blocks, team, lds, ta, st); }
        if ((int)lds > ta) ta = (int)lds;
        return;
    }
    int &a = attr[attr_slot()][(wide || global_a ? 8 : 0) + (global_a ? 4 : 0) + (priced ? 2 : 0) + (nw == 8 ? 1 : 0)];
    if (nw == 8) {
        if (global_a && priced) launch_search_w<true, true, 8>(pools, prm, n_blocks, waves, lds, a, wide, st);
        else if (global_a) launch_search_w<true, false, 8>(pools, prm, n_blocks, waves, lds, a, wide, st);
        else if (priced) launch_search_w<false, true, 8>(pools, prm, n_blocks, waves, lds, a, wide, st);
        else launch_search_w<false, false, 8>(pools, prm, n_blocks, waves, lds, a, wide, st);
    } else {
        if (global_a && priced) launch_search_w<true, true, 4>(pools, prm, n_blocks, waves, lds, a, wide, st);
        else if (global_a) launch_search_w<true, false, 4>(pools, prm, n_blocks, waves, lds, a, wide, st);
        else if (priced) launch_search_w<false, true, 4>(pools, prm, n_blocks, waves, lds, a, wide, st);
        else launch_search_w<false, false, 4>(pools, prm, n_blocks, waves, lds, a, wide, st);
    }
    if ((int)lds > a) a = (int)lds;
}

// K-init for one launch group of topics in global memory: `n_blocks` block-map entries of `per_block` restarts each.  False when the
// tables do not fit (the caller leaves prm.init = 1: k_search fills the holes itself).
bool launch_init(const SearchPools &pools, const SearchParams &prm, int n_blocks, int per_block, bool priced, int nw, void *stream) {
    const size_t lds = init_lds_bytes(prm.maxBx, prm.maxR, priced, prm.bw != 0);
    if (lds > 160 * 1024) return false;
    static int attr[kAttrDevices][4] = {{0}};
    int &a = attr[attr_slot()][(priced ? 2 : 0) + (nw == 8 ? 1 : 0)];
    const void *fn = nw == 8 ? (priced ? reinterpret_cast<const void *>(k_init<true, 8>) : reinterpret_cast<const void *>(k_init<false, 8>))
                             : (priced ? reinterpret_cast<const void *>(k_init<true, 4>) : reinterpret_cast<const void *>(k_init<false, 4>));
    if ((int)lds > a && lds > 64 * 1024) { (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); a = (int)lds; }
    // (config 5 as one topic, 15,000 holes x 16 rounds, per K-init: 59.2 / 21.8 / 17.3 / 18.0 ms with 1 / 4 / 8 / 16 wavefronts -- beyond two rounds
    // per wavefront the hole's fixed part, ~200 instructions of reduce / exchange / update behind one another, is what is left)
    int waves = std::min(8, std::max(1, (prm.maxBx + 63) / 64));
    if (const char *e = std::getenv("KAO_INIT_WAVES")) waves = std::min(kInitWaves, std::max(1, std::atoi(e)));   // measurement hook (0: see kao_session_step)
    while (waves & (waves - 1)) waves &= waves - 1;   // a power of two (the kernel's owner-of-a-round mask)
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)(n_blocks * per_block)), block((unsigned)(64 * waves));
    if (nw == 8) { if (priced) hipLaunchKernelGGL((k_init<true, 8>), grid, block, lds, st, pools, prm, per_block); else hipLaunchKernelGGL((k_init<false, 8>), grid, block, lds, st, pools, prm, per_block); }
    else { if (priced) hipLaunchKernelGGL((k_init<true, 4>), grid, block, lds, st, pools, prm, per_block); else hipLaunchKernelGGL((k_init<false, 4>), grid, block, lds, st, pools, prm, per_block); }
    return true;
}

void launch_eval(const EvalPools &pools, int n_blocks, int ne, void *stream) {
    const size_t lds = eval_lds_bytes(pools.maxP, pools.maxB, pools.cur_in_lds != 0, ne);
    int &g_attr_eval = g_attr_eval_dev[attr_slot()];
    if ((int)lds > g_attr_eval) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_eval<4, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_eval<4, false, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_eval<8, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_eval<4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_eval<8, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        g_attr_eval = (int)lds;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (pools.coop) {
        if (ne == 8) hipLaunchKernelGGL((k_eval<8, true>), dim3(n_blocks), dim3(256), lds, st, pools);
        else hipLaunchKernelGGL((k_eval<4, true>), dim3(n_blocks), dim3(256), lds, st, pools);
    } else {
        if (ne == 8) hipLaunchKernelGGL((k_eval<8, false>), dim3(n_blocks), dim3(256), lds, st, pools);
        else if (pools.rf_uniform == 3 && pools.cur_in_lds) hipLaunchKernelGGL((k_eval<4, false, 3>), dim3(n_blocks), dim3(256), lds, st, pools);
        else hipLaunchKernelGGL((k_eval<4, false>), dim3(n_blocks), dim3(256), lds, st, pools);
    }
}

void launch_gather(const TopicDev *topics, int n_topics, const unsigned long long *keys, const uint16_t *best_pool,
                   const int32_t *viol, uint16_t *win_assign, int32_t *win_viol, void *stream) {
    hipLaunchKernelGGL(k_gather, dim3(n_topics), dim3(64), 0, static_cast<hipStream_t>(stream), topics, keys, best_pool, viol,
                       win_assign, win_viol);
}

void launch_adopt_global(unsigned long long *keys, const unsigned long long *glob, int n, void *stream) {
    hipLaunchKernelGGL(k_adopt_global, dim3((n + 63) / 64), dim3(64), 0, static_cast<hipStream_t>(stream), keys, glob, n);
}

size_t canon_lds_bytes(int maxBx) {
    const size_t bx64 = ((size_t)maxBx + 63) & ~(size_t)63;
    return kRackTab * 4 + bx64 + bx64 * 4 + kRackTab * 4;
}

void launch_canon(const TopicDev *topic, const uint32_t *cur_words, const uint16_t *ext, const int32_t *rsz, uint32_t *A, int maxBx,
                  int nw, int32_t *status, void *stream) {
    const size_t lds = canon_lds_bytes(maxBx);
    if (nw == 8) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_canon<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k_canon<8>, dim3(1), dim3(64), lds, static_cast<hipStream_t>(stream), topic, reinterpret_cast<const Part<8> *>(cur_words), ext, rsz,
                           reinterpret_cast<Part<8> *>(A), maxBx, status);
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_canon<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k_canon<4>, dim3(1), dim3(64), lds, static_cast<hipStream_t>(stream), topic, reinterpret_cast<const Part<4> *>(cur_words), ext, rsz,
                           reinterpret_cast<Part<4> *>(A), maxBx, status);
    }
}

}  // namespace kao
