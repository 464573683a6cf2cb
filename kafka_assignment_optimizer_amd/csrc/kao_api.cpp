// kao_api.cpp -- host side of libkao.so: the C ABI declared in include/kao.h.
//
// Everything that computes runs in the gfx950 kernels of kao_kernels.hip; this file only prepares
// instances (dense -> rack-major internal broker index), owns the device pools, launches, and reads
// results back.  There is deliberately no CPU evaluation or search path here: if the HIP device is
// missing every compute entry point fails with KAO_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types and prototypes only: librccl.so is loaded on first use (kao_solve_multi), see Rccl below
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/kao.h"
#include "kao_internal.h"

using namespace kao;

namespace {

thread_local std::string g_err;
int g_device = -1;           // the process default (kao_init)
bool g_init = false;
thread_local int t_device = -1;  // per-thread override: kao_solve_multi drives several devices from one process
constexpr int kMaxDevices = 64;
int g_num_cu_of[kMaxDevices] = {0};

int cur_device() { return t_device >= 0 ? t_device : g_device; }
int num_cu(int device) {
    if (device < 0 || device >= kMaxDevices) return 256;
    if (!g_num_cu_of[device]) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || v <= 0) v = 256;
        g_num_cu_of[device] = v;
    }
    return g_num_cu_of[device];
}

int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}
#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(KAO_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));       \
    } while (0)

double now_s() {
    using namespace std::chrono;
    return duration<double>(steady_clock::now().time_since_epoch()).count();
}

void floor_ceil(int64_t num, int64_t den, int32_t &lo, int32_t &hi) {
    lo = (int32_t)(num / den);
    hi = (int32_t)((num + den - 1) / den);
}

int validate(const kao_topic *t) {
    if (!t) return fail(KAO_ERR_INVALID, "null topic");
    if (t->n_brokers < 1 || t->n_brokers > 65534) return fail(KAO_ERR_INVALID, "n_brokers out of range");
    if (t->n_racks < 1) return fail(KAO_ERR_INVALID, "n_racks < 1");
    if (t->n_racks > KAO_MAX_RACKS) return fail(KAO_ERR_UNSUPPORTED, "more than 255 racks");
    if (t->n_partitions < 1) return fail(KAO_ERR_INVALID, "n_partitions < 1");
    if (t->rf < 1 || t->rf_cur < 1) return fail(KAO_ERR_INVALID, "rf < 1");
    if (t->rf > KAO_MAX_RF || t->rf_cur > KAO_MAX_RF) return fail(KAO_ERR_UNSUPPORTED, "replication factor > 8");
    if (t->rf > t->n_brokers) return fail(KAO_ERR_INVALID, "rf > n_brokers");
    if (!t->rack_of || !t->current) return fail(KAO_ERR_INVALID, "null rack_of/current");
    for (int b = 0; b < t->n_brokers; ++b)
        if (t->rack_of[b] >= t->n_racks) return fail(KAO_ERR_INVALID, "rack_of entry >= n_racks");
    int wmax = 0;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j) {
            if (t->w[i][j] < 0 || t->w[i][j] > 1023) return fail(KAO_ERR_INVALID, "objective weights must be 0..1023");
            wmax = std::max(wmax, t->w[i][j]);
        }
    int bwmax = 0, bwlmax = 0;
    for (int b = 0; b < t->n_brokers; ++b) {
        const int v = t->broker_w ? t->broker_w[b] : 0, vl = t->broker_wl ? t->broker_wl[b] : 0;
        if (v < 0 || v > 1023 || vl < 0 || vl > 1023) return fail(KAO_ERR_INVALID, "broker weights must be 0..1023");
        bwmax = std::max(bwmax, v); bwlmax = std::max(bwlmax, vl);
    }
    if ((int64_t)t->n_partitions * t->rf * (wmax + bwmax) + (int64_t)t->n_partitions * bwlmax > 0xFFFFFE)
        return fail(KAO_ERR_UNSUPPORTED, "objective can exceed 24 bits (n_partitions * rf * largest weight)");
    for (int p = 0; p < t->n_partitions; ++p) {  // one LP variable per (broker, partition) (README.md:146): a broker cannot be listed twice
        const uint16_t *c = t->current + (size_t)p * t->rf_cur;
        for (int k = 1; k < t->rf_cur; ++k)
            for (int j = 0; j < k; ++j)
                if (c[k] == c[j] && c[k] < t->n_brokers)
                    return fail(KAO_ERR_INVALID, "current assignment lists a broker twice in partition " + std::to_string(p));
    }
    if ((int64_t)t->n_partitions * t->rf > 4000000) return fail(KAO_ERR_UNSUPPORTED, "more than 4,000,000 replicas in one topic");
    if ((int64_t)t->n_partitions * t->rf_cur > 65535) {  // per-broker counters are 16 + 16 bits (replicas | leaders)
        std::vector<int32_t> cnt((size_t)t->n_brokers, 0);
        for (int64_t i = 0, n = (int64_t)t->n_partitions * t->rf_cur; i < n; ++i)
            if (t->current[i] < t->n_brokers && ++cnt[t->current[i]] > 65535)
                return fail(KAO_ERR_UNSUPPORTED, "current assignment puts more than 65,535 replicas on one broker (16-bit per-broker counters)");
    }
    if (((int64_t)t->n_partitions * t->rf + t->n_brokers - 1) / t->n_brokers > 30000)
        return fail(KAO_ERR_UNSUPPORTED, "more than 30,000 replicas per broker on average (16-bit per-broker counters)");
    return KAO_OK;
}

void derive_bounds(const kao_topic *t, int32_t o[8]) {
    const int64_t B = t->n_brokers, R = t->n_racks, P = t->n_partitions, RF = t->rf;
    floor_ceil(P * RF, B, o[0], o[1]);  // C3 README.md:158-161
    floor_ceil(P, B, o[2], o[3]);       // C4 README.md:163-166
    floor_ceil(P * RF, R, o[4], o[5]);  // C6 README.md:173-176
    floor_ceil(RF, R, o[6], o[7]);      // C7 README.md:178-180
    const int32_t ov[8] = {t->rep_lo, t->rep_hi, t->lead_lo, t->lead_hi, t->rack_lo, t->rack_hi, t->prack_lo, t->prack_hi};
    for (int i = 0; i < 8; ++i)
        if (ov[i] >= 0) o[i] = ov[i];
}

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

int prepare(const kao_topic *t, uint64_t seed, PreparedTopic &pt) {
    int rc = validate(t);
    if (rc) return rc;
    const int B = t->n_brokers, R = t->n_racks, P = t->n_partitions;
    int32_t bd[8];
    derive_bounds(t, bd);
    pt.rack_size.assign(R, 0);
    for (int b = 0; b < B; ++b) pt.rack_size[t->rack_of[b]]++;
    int m = 0;
    for (int r = 0; r < R; ++r) m = std::max(m, pt.rack_size[r]);
    m = std::max(m, 2);  // floor(2^32/m)+1 must fit 32 bits: single-broker racks get a stride of 2
    const int64_t Bx = (int64_t)R * m;
    if (Bx > 65534) return fail(KAO_ERR_UNSUPPORTED, "racks x largest-rack exceeds the 16-bit internal index");
    pt.int_of.assign(B, 0);
    pt.ext_of.assign((size_t)Bx, (uint16_t)KAO_NONE);
    std::vector<int> fill(R, 0);
    for (int b = 0; b < B; ++b) {  // dense order inside each rack is preserved
        const int r = t->rack_of[b];
        const int x = r * m + fill[r]++;
        pt.int_of[b] = (uint16_t)x;
        pt.ext_of[x] = (uint16_t)b;
    }
    const int nw = (t->rf > kRFP || t->rf_cur > kRFP) ? kMaxRF : kRFP;   // replica words per partition in K-search / K-canon / K-eval
    pt.cur_int.assign((size_t)P * nw, (uint16_t)KAO_NONE);
    for (int p = 0; p < P; ++p)
        for (int k = 0; k < t->rf_cur; ++k) {
            const unsigned b = t->current[(size_t)p * t->rf_cur + k];
            if (b < (unsigned)B) pt.cur_int[(size_t)p * nw + k] = pt.int_of[b];
        }
    pt.rack_of.assign(t->rack_of, t->rack_of + B);
    pt.cur_dense.assign(t->current, t->current + (size_t)P * t->rf_cur);
    TopicDev &d = pt.d;
    d.P = P; d.RF = t->rf; d.R = R; d.m = m; d.Bx = (int32_t)Bx;
    d.magic = (uint32_t)(0x100000000ull / (uint64_t)m) + 1u;
    d.rep_lo = bd[0]; d.rep_hi = bd[1]; d.lead_lo = bd[2]; d.lead_hi = bd[3];
    d.rack_lo = bd[4]; d.rack_hi = bd[5]; d.prack_lo = bd[6]; d.prack_hi = bd[7];
    d.w00 = t->w[0][0]; d.w01 = t->w[0][1]; d.w10 = t->w[1][0]; d.w11 = t->w[1][1];
    d.seed_lo = (uint32_t)seed; d.seed_hi = (uint32_t)(seed >> 32);
    d.B = B; d.rf_cur = t->rf_cur; d.nw = nw;
    d.has_bw = (t->broker_w || t->broker_wl) ? 1 : 0;
    if (d.has_bw) {
        pt.bw_dense.assign((size_t)B, 0);
        pt.bw_int.assign((size_t)Bx, 0);
        for (int b = 0; b < B; ++b) {
            const uint32_t v = (uint32_t)(t->broker_w ? t->broker_w[b] : 0) | ((uint32_t)(t->broker_wl ? t->broker_wl[b] : 0) << 16);
            pt.bw_dense[(size_t)b] = v;
            pt.bw_int[pt.int_of[(size_t)b]] = v;
        }
    }
    return KAO_OK;
}

// Necessary conditions checked by counting; empty string = not provably infeasible (kao_check_infeasible).
std::string infeasible_reason(const kao_topic *t) {
    const int64_t B = t->n_brokers, R = t->n_racks, P = t->n_partitions, RF = t->rf, n = P * RF;
    int32_t bd[8];
    derive_bounds(t, bd);
    const int64_t rep_lo = bd[0], rep_hi = bd[1], lead_lo = bd[2], lead_hi = bd[3], rack_lo = bd[4], rack_hi = bd[5], prack_lo = bd[6], prack_hi = bd[7];
    std::vector<int64_t> rs((size_t)R, 0);
    for (int b = 0; b < t->n_brokers; ++b) rs[t->rack_of[b]]++;
    if (RF > B) return "rf > brokers";
    if (!(B * rep_lo <= n && n <= B * rep_hi)) return "replicas per broker band cannot hold P*RF replicas";
    if (!(B * lead_lo <= P && P <= B * lead_hi)) return "leaders per broker band cannot hold P leaders";
    if (!(R * rack_lo <= n && n <= R * rack_hi)) return "replicas per rack band cannot hold P*RF replicas";
    int64_t sum_hi = 0, sum_lo = 0, spread = 0;
    for (int r = 0; r < t->n_racks; ++r) {
        const int64_t hi = std::min(rack_hi, std::min(rs[(size_t)r] * rep_hi, P * std::min(prack_hi, rs[(size_t)r])));
        const int64_t lo = std::max(rack_lo, std::max(rs[(size_t)r] * rep_lo, P * prack_lo));
        if (lo > hi) return "rack " + std::to_string(r) + ": needs at least " + std::to_string(lo) + " replicas but can hold at most " + std::to_string(hi);
        sum_hi += hi; sum_lo += lo;
        spread += std::min(prack_hi, rs[(size_t)r]);
        if (prack_lo > rs[(size_t)r]) return "per-partition rack floor cannot be met";
    }
    if (sum_hi < n) return "rack capacities sum below P*RF";
    if (sum_lo > n) return "rack floors sum above P*RF";
    if (spread < RF) return "a partition cannot spread RF replicas over the racks";
    if (R * prack_lo > RF) return "per-partition rack floor cannot be met";
    return "";
}

// Best value one partition can collect from a kept set: its current leader (if kept) and n_fol kept current
// followers, at most rf replicas, exactly one leader; coupling rows (C3, C4, C6, C7) ignored.
int64_t partition_value(const kao_topic *t, bool lead_kept, int n_fol, bool leader_may_lead = true) {
    const int slots = t->rf - 1;
    const int wLL = t->w[0][0], wLF = t->w[0][1], wFL = t->w[1][0], wFF = t->w[1][1];
    auto fol_sum = [&](int n_ff, bool old_leader) {  // best `slots` follower gains among n_ff x wFF (+ wLF)
        int64_t v = 0;
        int left = slots;
        const bool lf_first = old_leader && wLF > wFF;
        if (lf_first && left > 0 && wLF > 0) { v += wLF; --left; }
        const int take = std::min(left, n_ff);
        if (wFF > 0 && take > 0) { v += (int64_t)take * wFF; left -= take; }
        if (old_leader && !lf_first && left > 0 && wLF > 0) v += wLF;
        return v;
    };
    int64_t best = fol_sum(n_fol, lead_kept);                                         // a new broker leads
    if (lead_kept && leader_may_lead) best = std::max(best, wLL + fol_sum(n_fol, false));  // current leader stays leader
    if (n_fol) best = std::max(best, wFL + fol_sum(n_fol - 1, lead_kept));            // a current follower is promoted
    return best;
}

// Upper bound on the objective (kao_upper_bound): every partition keeps its best surviving replicas in
// their best roles, minus the cheapest way to perform the evictions / leader changes that EVERY feasible
// assignment must perform.  f_p(K) = partition_value of a kept subset K.  With s_b / s_r / s_(p,r) the
// surviving replicas per broker / rack / (partition, rack) cell, at least
//   k = max( sum_b (s_b - rep_hi)+, sum_r (s_r - rack_hi)+, sum_cells (s - prack_hi)+,
//            n_surv + sum_b (rep_lo - s_b)+ - P*RF,  n_surv + sum_r (rack_lo - s_r)+ - P*RF )
// replicas cannot be kept (one eviction lowers one broker, one rack and one cell count; lower bands need
// arrivals and only P*RF - kept slots can take them).  g_p(j) = f_p(all) - max_{|K| = n_p - j} f_p(K); the
// total loss is at least the k smallest marginals of the lower convex envelopes of the g_p.  Brokers with
// more current leaders than lead_hi force leader changes, each costing f_p(all) - f_p(leader not leading).
// Both losses may hit the same partitions, so the larger is subtracted; the result is then capped by the
// per-broker capacity bound below.  Restated (and checked against the exact optimum) in
// oracle/kao_oracle.py::upper_bound_forced / upper_bound_broker.
int64_t upper_bound(const kao_topic *t) {
    const int B = t->n_brokers, R = t->n_racks, P = t->n_partitions, RF = t->rf;
    int32_t bd[8];
    derive_bounds(t, bd);
    const int rep_lo = bd[0], rep_hi = bd[1], lead_hi = bd[3], rack_lo = bd[4], rack_hi = bd[5], prack_hi = bd[7];
    // Everything per partition depends only on (current leader survives, number of surviving followers): 2 x 4 combos.
    struct Combo { int64_t f_all = 0, lead_loss = 0; int n_marg = 0; int64_t marg[KAO_MAX_RF + 1] = {0}; int64_t count = 0; } combo[2 * KAO_MAX_RF];
    for (int la = 0; la < 2; ++la)
        for (int n_fol = 0; n_fol < KAO_MAX_RF; ++n_fol) {
            Combo &c = combo[la * KAO_MAX_RF + n_fol];
            const bool lead_alive = la != 0;
            const int n_p = n_fol + la;
            c.f_all = partition_value(t, lead_alive, n_fol);
            int64_t g[KAO_MAX_RF + 1];
            for (int j = 0; j <= n_p; ++j) {  // cheapest loss of evicting j replicas (followers are interchangeable)
                int64_t best = -1;
                for (int drop_lead = 0; drop_lead <= la; ++drop_lead) {
                    const int df = j - drop_lead;
                    if (df >= 0 && df <= n_fol) best = std::max(best, partition_value(t, lead_alive && !drop_lead, n_fol - df));
                }
                g[j] = c.f_all - best;
            }
            int hx[KAO_MAX_RF + 1]; int64_t hy[KAO_MAX_RF + 1]; int hn = 0;  // lower convex envelope of (j, g[j])
            for (int j = 0; j <= n_p; ++j) {
                hx[hn] = j; hy[hn] = g[j]; ++hn;
                while (hn >= 3 && (hy[hn - 2] - hy[hn - 3]) * (hx[hn - 1] - hx[hn - 3]) >= (hy[hn - 1] - hy[hn - 3]) * (hx[hn - 2] - hx[hn - 3])) {
                    hx[hn - 2] = hx[hn - 1]; hy[hn - 2] = hy[hn - 1]; --hn;
                }
            }
            for (int i = 0; i + 1 < hn; ++i)
                for (int x = hx[i] + 1; x <= hx[i + 1]; ++x) {  // integer floor of the envelope stays a lower bound
                    const int64_t dy = hy[i + 1] - hy[i], dx = hx[i + 1] - hx[i];
                    const int64_t prev = hy[i] + (dy * (x - 1 - hx[i])) / dx, now = hy[i] + (dy * (x - hx[i])) / dx;
                    c.marg[c.n_marg++] = now - prev;
                }
            if (lead_alive) {
                const int64_t alt = std::max(partition_value(t, false, n_fol), partition_value(t, true, n_fol, false));
                c.lead_loss = std::max<int64_t>(0, c.f_all - alt);
            }
        }
    std::vector<int> s_b((size_t)B, 0), s_r((size_t)R, 0), lead_b((size_t)B, 0);
    std::vector<int> nl_b((size_t)B, 0);  // surviving current LEADER replicas per broker (followers = s_b - nl_b)
    std::vector<int> touched;             // brokers holding at least one surviving replica
    int64_t total = 0, n_surv = 0, cell_excess = 0, parts_with_survivor = 0;
    for (int p = 0; p < P; ++p) {
        const uint16_t *c = t->current + (size_t)p * t->rf_cur;
        const bool lead_alive = c[0] < (unsigned)B;
        int n_fol = 0, racks[KAO_MAX_RF], n_in = 0;
        for (int k = 0; k < t->rf_cur; ++k) {
            if (c[k] >= (unsigned)B) continue;
            if (k > 0) ++n_fol;
            if (s_b[c[k]]++ == 0) touched.push_back((int)c[k]);
            s_r[t->rack_of[c[k]]]++;
            racks[n_in++] = t->rack_of[c[k]];
        }
        n_surv += n_in;
        parts_with_survivor += n_in > 0;
        for (int i = 0; i < n_in; ++i) {  // cells: count each rack once
            bool first = true;
            int cnt = 0;
            for (int j = 0; j < n_in; ++j) { if (racks[j] == racks[i]) { ++cnt; if (j < i) first = false; } }
            if (first) cell_excess += std::max(0, cnt - prack_hi);
        }
        Combo &cb = combo[(lead_alive ? KAO_MAX_RF : 0) + n_fol];
        total += cb.f_all;
        cb.count++;
        if (lead_alive) { nl_b[c[0]]++; lead_b[c[0]]++; }
    }
    int64_t ex_b = 0, ex_r = 0, need_b = 0, need_r = 0;
    need_b = (int64_t)std::max(0, rep_lo) * ((int64_t)B - (int64_t)touched.size());  // untouched brokers hold nothing
    for (int b : touched) { ex_b += std::max(0, s_b[(size_t)b] - rep_hi); need_b += std::max(0, rep_lo - s_b[(size_t)b]); }
    for (int r = 0; r < R; ++r) { ex_r += std::max(0, s_r[r] - rack_hi); need_r += std::max(0, rack_lo - s_r[r]); }
    const int64_t slots = (int64_t)P * RF;
    int64_t k = std::max<int64_t>({ex_b, ex_r, cell_excess, n_surv + need_b - slots, n_surv + need_r - slots, 0});
    // the k smallest marginals over all partitions, taken combo by combo (value, multiplicity)
    std::vector<std::pair<int64_t, int64_t>> vm;
    for (const Combo &c : combo)
        for (int i = 0; i < c.n_marg; ++i)
            if (c.count) vm.emplace_back(c.marg[i], c.count);
    std::sort(vm.begin(), vm.end());
    int64_t evict_loss = 0, left = k;
    for (const auto &e : vm) {
        if (left <= 0) break;
        const int64_t take = std::min(left, e.second);
        evict_loss += take * e.first;
        left -= take;
    }
    int64_t lead_loss = 0;
    bool over_led = false;
    for (int b : touched) over_led |= lead_b[(size_t)b] > lead_hi;
    if (over_led) {  // rare: collect the per-partition losses only for brokers holding too many current leaders
        std::vector<std::pair<int, int64_t>> lead_losses;
        for (int p = 0; p < P; ++p) {
            const uint16_t *c = t->current + (size_t)p * t->rf_cur;
            if (c[0] >= (unsigned)B || lead_b[c[0]] <= lead_hi) continue;
            int n_fol = 0;
            for (int q = 1; q < t->rf_cur; ++q) n_fol += c[q] < (unsigned)B;
            lead_losses.emplace_back((int)c[0], combo[KAO_MAX_RF + n_fol].lead_loss);
        }
        std::sort(lead_losses.begin(), lead_losses.end());
        for (size_t i = 0; i < lead_losses.size();) {
            size_t j = i;
            while (j < lead_losses.size() && lead_losses[j].first == lead_losses[i].first) ++j;
            const int ex = lead_b[(size_t)lead_losses[i].first] - lead_hi;
            for (size_t q = i; q < j && (int)(q - i) < ex; ++q) lead_loss += lead_losses[q].second;  // sorted by loss within a broker
            i = j;
        }
    }
    // Per-broker capacity bound with a global cap on leading survivors.  A broker keeps at most rep_hi of its
    // surviving replicas and at most lead_hi of them lead; a replica that leads is worth w[cur_role][0], one that
    // follows w[cur_role][1] (one-leader-per-partition and rack rows relaxed).  v_b(L) = best value on broker b with
    // at most L survivors leading.  Brokers with fewer than lead_lo survivors must receive lead_lo - s_b NEW leaders,
    // so at most Lcap = min(#partitions with a survivor, P - sum_b (lead_lo - s_b)+) partitions keep a surviving
    // replica as leader: bound = sum_b v_b(0) + the Lcap largest marginals of the upper concave envelopes of the v_b.
    // Charges forced evictions AND forced leader changes together.
    const int wLL = t->w[0][0], wLF = t->w[0][1], wFL = t->w[1][0], wFF = t->w[1][1];
    const int lead_lo = bd[2];
    int64_t lcap = P - (int64_t)std::max(0, lead_lo) * ((int64_t)B - (int64_t)touched.size());
    for (int b : touched) lcap -= std::max(0, lead_lo - s_b[(size_t)b]);
    lcap = std::min<int64_t>(lcap, parts_with_survivor);
    // brokers with the same (surviving leaders, surviving followers) share v_b: evaluate each distinct pair once
    // (only brokers that hold a surviving replica are visited; all others are the kind (0, 0))
    std::vector<std::pair<std::pair<int, int>, int64_t>> kinds;  // ((n_l, n_f), number of brokers)
    kinds.push_back({{0, 0}, (int64_t)B - (int64_t)touched.size()});
    for (int b : touched) {
        const std::pair<int, int> key{nl_b[(size_t)b], s_b[(size_t)b] - nl_b[(size_t)b]};
        size_t i = 0;
        while (i < kinds.size() && kinds[i].first != key) ++i;
        if (i == kinds.size()) kinds.push_back({key, 0});
        kinds[i].second++;
    }
    int64_t broker_base = 0;
    std::vector<std::pair<int64_t, int64_t>> lead_marg;  // (marginal value, multiplicity)
    std::vector<int64_t> v, hx, hy;
    for (const auto &kind : kinds) {
        const int n_l = kind.first.first, n_f = kind.first.second;
        const int64_t mult = kind.second;
        const int lmax = std::min(std::min(lead_hi, rep_hi), n_l + n_f);
        v.assign((size_t)lmax + 1, -1);
        for (int x = 0; x <= std::min(n_l, lmax); ++x)
            for (int y = 0; y <= std::min(n_f, lmax - x); ++y) {
                int64_t val = (int64_t)x * wLL + (int64_t)y * wFL;
                int slots = rep_hi - x - y;
                const int ga = n_l - x, gb = n_f - y;  // ga replicas worth wLF as followers, gb worth wFF
                const int hi_w = std::max(wLF, wFF), lo_w = std::min(wLF, wFF);
                const int hi_n = wLF >= wFF ? ga : gb, lo_n = wLF >= wFF ? gb : ga;
                const int t1 = std::min(hi_n, slots);
                if (hi_w > 0) val += (int64_t)t1 * hi_w;
                slots -= t1;
                if (lo_w > 0) val += (int64_t)std::min(lo_n, slots) * lo_w;
                v[(size_t)(x + y)] = std::max(v[(size_t)(x + y)], val);
            }
        for (int i = 1; i <= lmax; ++i) v[(size_t)i] = std::max(v[(size_t)i], v[(size_t)i - 1]);  // "at most L leading"
        broker_base += v[0] * mult;
        hx.clear(); hy.clear();  // upper concave envelope of (L, v[L]) -> non-increasing marginals
        for (int i = 0; i <= lmax; ++i) {
            hx.push_back(i); hy.push_back(v[(size_t)i]);
            while (hx.size() >= 3) {
                const size_t n = hx.size();
                if ((hy[n - 2] - hy[n - 3]) * (hx[n - 1] - hx[n - 3]) <= (hy[n - 1] - hy[n - 3]) * (hx[n - 2] - hx[n - 3])) {
                    hx[n - 2] = hx[n - 1]; hy[n - 2] = hy[n - 1]; hx.pop_back(); hy.pop_back();
                } else break;
            }
        }
        for (size_t i = 0; i + 1 < hx.size(); ++i)
            for (int64_t x = hx[i] + 1; x <= hx[i + 1]; ++x) {  // ceil of the running total keeps it an upper bound
                const int64_t dy = hy[i + 1] - hy[i], dx = hx[i + 1] - hx[i];
                auto up = [&](int64_t k) { const int64_t num = dy * k; return hy[i] + (num >= 0 ? (num + dx - 1) / dx : -((-num) / dx)); };
                const int64_t m = up(x - hx[i]) - up(x - 1 - hx[i]);
                if (m > 0) lead_marg.emplace_back(m, mult);
            }
    }
    std::sort(lead_marg.begin(), lead_marg.end(), [](const std::pair<int64_t, int64_t> &p, const std::pair<int64_t, int64_t> &q) { return p.first > q.first; });
    int64_t broker_bound = broker_base, cap_left = std::max<int64_t>(lcap, 0);
    for (const auto &e : lead_marg) {
        if (cap_left <= 0) break;
        const int64_t take = std::min(cap_left, e.second);
        broker_bound += take * e.first;
        cap_left -= take;
    }
    return std::min(total - std::max(evict_loss, lead_loss), broker_bound);
}

// Closed-form bound of a topic that may carry broker weights: the weights are bounded term by term -- the band rows allow at
// most min(rep_hi, P) replicas and min(lead_hi, P) leaders on a broker, P*RF replicas and P leaders in all, so the
// weight part is at most the greedy fill of those capacities in descending weight order.
int64_t upper_bound_w(const kao_topic *t) {
    int64_t ub = upper_bound(t);
    if (!t->broker_w && !t->broker_wl) return ub;
    int32_t bd[8];
    derive_bounds(t, bd);
    for (int kind = 0; kind < 2; ++kind) {
        const int32_t *w = kind == 0 ? t->broker_w : t->broker_wl;
        if (!w) continue;
        std::vector<int> v(w, w + t->n_brokers);
        std::sort(v.begin(), v.end(), std::greater<int>());
        int64_t left = kind == 0 ? (int64_t)t->n_partitions * t->rf : t->n_partitions;
        const int64_t per = std::min<int64_t>(kind == 0 ? bd[1] : bd[3], t->n_partitions);
        for (int x : v) {
            if (left <= 0) break;
            const int64_t take = std::min(left, per);
            ub += take * x;
            left -= take;
        }
    }
    return ub;
}

// Neighbours delta-evaluated by ONE restart over iterations [it0, it0+iters) (kao_kernels.hip, KAO-LS):
// move pattern R R X R L R X R; REPLACE scans all B brokers of one slot in even blocks of 8 iterations and
// samples 64 lanes x 4 brokers in odd blocks; EXCHANGE scans all P*RF partner slots (a window of 512 partitions
// when P > 512); LEADER-SWAP 64 x (RF-1).
uint64_t neighbours_in_range(uint32_t it0, uint32_t iters, int rf, int n_brokers, int n_partitions) {
    static const uint8_t pat[8] = {0, 0, 1, 0, 2, 0, 1, 0};
    uint64_t n = 0;
    for (uint32_t i = 0; i < iters; ++i) {
        const uint32_t it = it0 + i;
        const int type = pat[it & 7];
        if (type == 0) n += ((it >> 3) & 1u) ? 256ull : (uint64_t)n_brokers;
        else if (type == 1) n += (uint64_t)std::min(n_partitions, n_partitions > 512 ? 512 : n_partitions) * (uint64_t)rf;
        else n += 64ull * (uint64_t)(rf > 1 ? rf - 1 : 0);
    }
    return n;
}

// Sawtooth period by topic size: one ramp should span about 2 * P * RF iterations (every slot gets a chance to move
// while the penalty is low).  Measured on a drifted 2000-partition topic (optimum 14812): 2^8 -> 14777, 2^11 -> 14794,
// 2^14 -> 14801..14806; small topics keep the 2^8 they were tuned with.
int auto_period_log2(int P, int RF) {
    int64_t n = 2 * (int64_t)P * RF;
    int lg = 0;
    while (n > 1) { n >>= 1; ++lg; }
    return std::min(16, std::max(8, lg));
}

// K-bound limits: 19 B of LDS per broker + 72 B per rack; 32-bit headroom of the priced values (weights x 4096,
// P*RF subgradients)
bool dual_supported(const kao_topic *t) {
    if (t->rf > kRFP || t->rf_cur > kRFP) return false;   // K-bound's per-lane subproblem holds 4 replicas
    if (t->broker_w || t->broker_wl) return false;        // K-bound prices the README rows only
    if (bound_lds_bytes(t->n_brokers, 0, t->n_racks, false) > 160 * 1024) return false;
    const int64_t n = (int64_t)t->n_partitions * t->rf;
    if (n > 131072) return false;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            if (t->w[i][j] < 0 || t->w[i][j] > 255) return false;
    return true;
}

template <typename T>
int dev_alloc_copy(T **dst, const std::vector<T> &src) {
    *dst = nullptr;
    const size_t n = std::max<size_t>(src.size(), 1);
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(dst), n * sizeof(T)));
    if (!src.empty()) HIP_TRY(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return KAO_OK;
}

// Physical workgroup order: the dispatcher places workgroup b on XCD b % 8 (observed, used for L2
// affinity only), so the workgroups of one topic -- which read the same tables -- are dealt to one XCD.
template <typename Item>
std::vector<Item> xcd_order(const std::vector<Item> &items, const std::vector<int> &topic_of) {
    const size_t n = items.size();
    std::vector<std::vector<size_t>> q(8);
    for (size_t i = 0; i < n; ++i) q[(size_t)topic_of[i] % 8].push_back(i);
    std::vector<size_t> head(8, 0);
    std::vector<Item> out;
    out.reserve(n);
    for (size_t b = 0; b < n; ++b) {
        size_t x = b % 8;
        if (head[x] >= q[x].size()) {  // this XCD's queue ran dry: steal from the longest remaining one
            size_t bestx = 0, bestlen = 0;
            for (size_t y = 0; y < 8; ++y)
                if (q[y].size() - head[y] > bestlen) { bestlen = q[y].size() - head[y]; bestx = y; }
            x = bestx;
            out.push_back(items[q[x].back()]);
            q[x].pop_back();
            continue;
        }
        out.push_back(items[q[x][head[x]++]]);
    }
    return out;
}

}  // namespace

// =================================================================================================
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
    std::vector<char> topic_infeasible;  // per topic: proven infeasible by counting (kao_check_infeasible)
    std::vector<char> dual_ok;           // per topic: within K-bound's limits
    std::vector<int64_t> h_dual_target;  // staging for kao_session_bound_step
    std::vector<int32_t> h_dual_ids;
    std::vector<int2> h_wide_map;        // sliced K-bound: {topic, slice} per workgroup (staging, like h_dual_ids)
    uint64_t wide_ctl_i32 = 0, wide_map_i32 = 0;   // int32 offsets of the control blocks / the map inside d_dual
    std::vector<int32_t> dual_flags, dual_iters;
    int total_restarts = 0;
    // Topics are bucketed by LDS footprint into launch groups (a 3000-partition topic must not impose its LDS carve
    // and its 2 waves per workgroup on 200 small topics); one K-search + one K-eval launch per group per step.
    struct LaunchGroup {
        int maxP = 0, maxBx = 0, maxB = 0;
        int waves = kWaves;  // restarts per K-search workgroup: 4, 2 or 1 -- the largest whose LDS carve fits 160 KiB
        int nw = kRFP;       // replica words per partition of the group's topics: 4 or 8 (template instantiation)
        bool global_a = false;   // topic too large for LDS: assignment + current words stay in global memory
        bool cur_in_lds = true;  // K-eval stages the current assignment in LDS (false: reads it from global)
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
    size_t dual_bytes = 0;
    hipStream_t stream_bound = nullptr;   // K-bound runs beside K-search on its own stream (it occupies one CU per topic)
    hipEvent_t ev_bound0 = nullptr, ev_bound1 = nullptr, ev_search = nullptr;
    bool bound_inflight = false;
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
    // profiling
    std::vector<hipEvent_t> ev;  // triples
    int ev_pending = 0;
    double ms_search = 0, ms_eval = 0;
    uint64_t eval_bytes_per_launch = 0;
    uint64_t delta_total = 0, search_bytes_total = 0;
};

namespace {

constexpr int kEvRing = 32;

// hipMalloc / hipFree cost 0.1-1 ms each; a finished session parks its arenas here for the next one
struct Parked { void *p; size_t bytes; int device; };
std::vector<Parked> g_parked;
constexpr size_t kParkMax = 4;
std::mutex g_cache_mu;  // guards g_parked / g_streams (sessions may be created from several host threads)

int arena_get(size_t bytes, void **out, size_t *cap) {
    std::lock_guard<std::mutex> lock(g_cache_mu);
    const int dev = cur_device();
    size_t best = g_parked.size();
    for (size_t i = 0; i < g_parked.size(); ++i)
        if (g_parked[i].device == dev && g_parked[i].bytes >= bytes && g_parked[i].bytes <= 4 * bytes + (1u << 20) &&
            (best == g_parked.size() || g_parked[i].bytes < g_parked[best].bytes)) best = i;
    if (best < g_parked.size()) {
        *out = g_parked[best].p; *cap = g_parked[best].bytes;
        g_parked.erase(g_parked.begin() + (long)best);
        return KAO_OK;
    }
    const size_t want = ((bytes + (1u << 16)) + 4095) & ~(size_t)4095;
    HIP_TRY(hipMalloc(out, want));
    *cap = want;
    return KAO_OK;
}
void arena_put(void *p, size_t bytes, int device) {
    if (!p) return;
    std::lock_guard<std::mutex> lock(g_cache_mu);
    (void)hipSetDevice(device);
    if (g_parked.size() >= kParkMax) {
        size_t small = 0;
        for (size_t i = 1; i < g_parked.size(); ++i) if (g_parked[i].bytes < g_parked[small].bytes) small = i;
        if (g_parked[small].bytes >= bytes) { (void)hipFree(p); return; }
        (void)hipSetDevice(g_parked[small].device);
        (void)hipFree(g_parked[small].p);
        (void)hipSetDevice(device);
        g_parked.erase(g_parked.begin() + (long)small);
    }
    g_parked.push_back({p, bytes, device});
}
std::vector<std::pair<hipStream_t, int>> g_streams;  // parked streams with their device (create/destroy cost ~1 ms)
int stream_get(hipStream_t *out) {
    std::lock_guard<std::mutex> lock(g_cache_mu);
    const int dev = cur_device();
    for (size_t i = 0; i < g_streams.size(); ++i)
        if (g_streams[i].second == dev) { *out = g_streams[i].first; g_streams.erase(g_streams.begin() + (long)i); return KAO_OK; }
    HIP_TRY(hipStreamCreateWithFlags(out, hipStreamNonBlocking));
    return KAO_OK;
}
void stream_put(hipStream_t st, int device) {
    if (!st) return;
    std::lock_guard<std::mutex> lock(g_cache_mu);
    if (g_streams.size() < 16) g_streams.push_back({st, device}); else { (void)hipSetDevice(device); (void)hipStreamDestroy(st); }
}
void arena_drop_all() {
    std::lock_guard<std::mutex> lock(g_cache_mu);
    for (auto &a : g_parked) { (void)hipSetDevice(a.device); (void)hipFree(a.p); }
    g_parked.clear();
    for (auto &st : g_streams) { (void)hipSetDevice(st.second); (void)hipStreamDestroy(st.first); }
    g_streams.clear();
    if (g_device >= 0) (void)hipSetDevice(g_device);
}
thread_local double g_timing[8] = {0, 0, 0, 0, 0, 0, 0, 0};
inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

int session_drain_events(kao_session *s) {
    for (int i = 0; i < s->ev_pending; ++i) {
        float a = 0, b = 0;
        HIP_TRY(hipEventSynchronize(s->ev[i * 3 + 2]));
        HIP_TRY(hipEventElapsedTime(&a, s->ev[i * 3 + 0], s->ev[i * 3 + 1]));
        HIP_TRY(hipEventElapsedTime(&b, s->ev[i * 3 + 1], s->ev[i * 3 + 2]));
        s->ms_search += a;
        s->ms_eval += b;
    }
    s->ev_pending = 0;
    return KAO_OK;
}

int require_init() {
    if (!g_init) {
        int rc = kao_init(g_device < 0 ? 0 : g_device);
        if (rc) return rc;
    }
    HIP_TRY(hipSetDevice(cur_device()));
    return KAO_OK;
}

}  // namespace

extern "C" {

int kao_version(void) { return KAO_VERSION; }

const char *kao_strerror(int code) {
    switch (code) {
        case KAO_OK: return "ok";
        case KAO_ERR_INVALID: return "invalid argument";
        case KAO_ERR_UNSUPPORTED: return "instance not supported by the gfx950 kernels";
        case KAO_ERR_NO_DEVICE: return "no usable HIP device (libkao has no CPU fallback)";
        case KAO_ERR_HIP: return "HIP runtime error";
        case KAO_ERR_NOMEM: return "out of memory";
        case KAO_ERR_NOT_INIT: return "kao_init not called";
        default: return "unknown error";
    }
}

const char *kao_last_error(void) { return g_err.c_str(); }

int kao_init(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return fail(KAO_ERR_NO_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
    if (device < 0 || device >= n) return fail(KAO_ERR_INVALID, "device ordinal out of range");
    e = hipSetDevice(device);
    if (e != hipSuccess) return fail(KAO_ERR_NO_DEVICE, std::string("hipSetDevice: ") + hipGetErrorString(e));
    (void)num_cu(device);
    g_device = device;
    g_init = true;
    return KAO_OK;
}

void kao_multi_shutdown_comms(void);

void kao_shutdown(void) {
    kao_multi_shutdown_comms();
    if (g_init) arena_drop_all();
    g_init = false;
}

int kao_device_name(char *buf, int len) {
    int rc = require_init();
    if (rc) return rc;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, cur_device()));
    std::snprintf(buf, (size_t)len, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return KAO_OK;
}

int kao_derive_bounds(const kao_topic *t, int32_t out[8]) {
    int rc = validate(t);
    if (rc) return rc;
    derive_bounds(t, out);
    return KAO_OK;
}

int kao_check_infeasible(const kao_topic *t, char *why, int why_len) {
    int rc = validate(t);
    if (rc) return rc;
    const std::string r = infeasible_reason(t);
    if (why && why_len > 0) std::snprintf(why, (size_t)why_len, "%s", r.c_str());
    return r.empty() ? 0 : 1;
}

int kao_upper_bound(const kao_topic *t, int64_t *ub) {
    int rc = validate(t);
    if (rc) return rc;
    *ub = upper_bound_w(t);
    return KAO_OK;
}

// ------------------------------------------------------------------------------------------------
// K-eval plans
// ------------------------------------------------------------------------------------------------
int kao_eval_plan_create(const kao_topic *t, kao_eval_plan **out) {
    if (!out) return fail(KAO_ERR_INVALID, "null out");
    *out = nullptr;
    int rc = require_init();
    if (rc) return rc;
    kao_eval_plan *p = new kao_eval_plan();
    p->device = cur_device();
    rc = prepare(t, 0, p->pt);
    if (rc) { delete p; return rc; }
    p->cur_in_lds = eval_lds_bytes(p->pt.d.P, p->pt.d.B, true, p->pt.d.nw) <= 160 * 1024;
    if (eval_lds_bytes(p->pt.d.P, p->pt.d.B, p->cur_in_lds, p->pt.d.nw) > 160 * 1024) { delete p; return fail(KAO_ERR_UNSUPPORTED, "broker tables exceed 160 KiB of LDS"); }
    p->pt.d.best_off = 0; p->pt.d.rackof_off = 0; p->pt.d.curd_off = 0; p->pt.d.bwd_off = 0;
    std::vector<TopicDev> td(1, p->pt.d);
    if ((rc = dev_alloc_copy(&p->d_topic, td)) || (rc = dev_alloc_copy(&p->d_rackof, p->pt.rack_of)) ||
        (rc = dev_alloc_copy(&p->d_curd, p->pt.cur_dense)) || (p->pt.d.has_bw && (rc = dev_alloc_copy(&p->d_bwd, p->pt.bw_dense)))) { kao_eval_plan_destroy(p); return rc; }
    hipError_t e = hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking);
    if (e == hipSuccess && (int64_t)p->pt.d.P * p->pt.d.RF > 65535) {
        e = hipMalloc(reinterpret_cast<void **>(&p->d_overflow), 4);
        if (e == hipSuccess) e = hipMemset(p->d_overflow, 0, 4);
    }
    if (e == hipSuccess) e = hipEventCreate(&p->ev0);
    if (e == hipSuccess) e = hipEventCreate(&p->ev1);
    if (e != hipSuccess) { kao_eval_plan_destroy(p); return fail(KAO_ERR_HIP, std::string("kao_eval_plan_create: ") + hipGetErrorString(e)); }
    *out = p;
    return KAO_OK;
}

int kao_eval_plan_run(kao_eval_plan *p, const void *d_candidates, int64_t n, void *d_objective, void *d_violations,
                      void *d_best_key) {
    if (!p || !d_candidates || n < 1) return fail(KAO_ERR_INVALID, "bad plan/candidates");
    if (n > (1 << 20)) return fail(KAO_ERR_INVALID, "at most 2^20 candidates per run (packed key id width)");
    HIP_TRY(hipSetDevice(p->device));
    if (n != p->map_n) {
        // small batches of large candidates (KAO-CX: <= 513 assignments of up to 10^5 slots) spread over the compute units: one
        // candidate per wavefront instead of eight, as soon as 32 per workgroup would leave most of the chip idle
        int cpb = p->cands_per_block;
        const int64_t fill = 4 * (int64_t)std::max(num_cu(p->device), 1);
        if (n < fill * 8) cpb = (int)std::min<int64_t>(cpb, std::max<int64_t>(kWaves, ((n + fill - 1) / fill) * kWaves));
        const int nb = (int)((n + cpb - 1) / cpb);
        std::vector<int4> map((size_t)nb);
        for (int b = 0; b < nb; ++b) {
            const int first = b * cpb;
            map[b] = make_int4(0, first, (int)std::min<int64_t>(cpb, n - first), first);
        }
        p->map_n = -1;  // no valid map until the new one is uploaded
        if (p->d_map) { int4 *old_map = p->d_map; p->d_map = nullptr; HIP_TRY(hipFree(old_map)); }
        int rc = dev_alloc_copy(&p->d_map, map);
        if (rc) return rc;
        p->map_n = n;
        p->map_blocks = nb;
    }
    EvalPools pl{};
    pl.topics = p->d_topic; pl.block_map = p->d_map; pl.rackof_pool = p->d_rackof; pl.curd_pool = p->d_curd;
    pl.cand = static_cast<const uint16_t *>(d_candidates);
    pl.objective = static_cast<int32_t *>(d_objective);
    pl.violations = static_cast<int32_t *>(d_violations);
    pl.best_key = static_cast<unsigned long long *>(d_best_key);
    pl.maxP = p->pt.d.P; pl.maxB = p->pt.d.B; pl.cur_in_lds = p->cur_in_lds ? 1 : 0;
    pl.overflow = p->d_overflow; pl.bwd_pool = p->d_bwd;
    HIP_TRY(hipEventRecord(p->ev0, p->stream));
    launch_eval(pl, p->map_blocks, p->pt.d.nw, p->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(p->ev1, p->stream));
    p->timed = true;
    return KAO_OK;
}

int kao_eval_plan_sync(kao_eval_plan *p, double *ms_last) {
    if (!p) return fail(KAO_ERR_INVALID, "null plan");
    HIP_TRY(hipStreamSynchronize(p->stream));
    if (p->d_overflow) {
        int32_t flag = 0;
        HIP_TRY(hipMemcpy(&flag, p->d_overflow, 4, hipMemcpyDeviceToHost));
        if (flag) {
            HIP_TRY(hipMemset(p->d_overflow, 0, 4));
            return fail(KAO_ERR_UNSUPPORTED, "a candidate puts more than 65,535 replicas on one broker (16-bit per-broker counters)");
        }
    }
    if (ms_last) {
        float ms = 0;
        if (p->timed) HIP_TRY(hipEventElapsedTime(&ms, p->ev0, p->ev1));
        *ms_last = ms;
    }
    return KAO_OK;
}

void kao_eval_plan_destroy(kao_eval_plan *p) {
    if (!p) return;
    (void)hipSetDevice(p->device);
    if (p->stream) (void)hipStreamSynchronize(p->stream);
    (void)hipFree(p->d_topic); (void)hipFree(p->d_rackof); (void)hipFree(p->d_curd); (void)hipFree(p->d_map); (void)hipFree(p->d_overflow); (void)hipFree(p->d_bwd);
    if (p->ev0) (void)hipEventDestroy(p->ev0);
    if (p->ev1) (void)hipEventDestroy(p->ev1);
    if (p->stream) (void)hipStreamDestroy(p->stream);
    delete p;
}

namespace {
// a plan plus growable device buffers, reused across batches (kao_canonicalize issues many small ones)
struct EvalCtx {
    kao_eval_plan *plan = nullptr;
    size_t per = 0, cap = 0;
    uint16_t *d_c = nullptr; int32_t *d_o = nullptr, *d_v = nullptr;
    ~EvalCtx() { (void)hipFree(d_c); (void)hipFree(d_o); (void)hipFree(d_v); kao_eval_plan_destroy(plan); }
    int open(const kao_topic *t) {
        per = (size_t)t->n_partitions * t->rf;
        return kao_eval_plan_create(t, &plan);
    }
    int run(const uint16_t *candidates, int64_t n, int32_t *objective, int32_t *violations) {
        const int64_t chunk_max = 1 << 20;
        for (int64_t done = 0; done < n; done += chunk_max) {
            const int64_t c = std::min(chunk_max, n - done);
            if ((size_t)c > cap) {
                (void)hipFree(d_c); (void)hipFree(d_o); (void)hipFree(d_v);
                d_c = nullptr; d_o = d_v = nullptr;
                cap = std::max<size_t>((size_t)c, std::min<size_t>(2 * cap + 64, (size_t)chunk_max));
                if (hipMalloc(reinterpret_cast<void **>(&d_c), cap * per * 2) != hipSuccess ||
                    hipMalloc(reinterpret_cast<void **>(&d_o), cap * 4) != hipSuccess ||
                    hipMalloc(reinterpret_cast<void **>(&d_v), cap * 32) != hipSuccess) { cap = 0; return fail(KAO_ERR_NOMEM, "hipMalloc"); }
            }
            HIP_TRY(hipMemcpy(d_c, candidates + (size_t)done * per, (size_t)c * per * 2, hipMemcpyHostToDevice));
            int rc = kao_eval_plan_run(plan, d_c, c, d_o, d_v, nullptr);
            if (!rc) rc = kao_eval_plan_sync(plan, nullptr);
            if (rc) return rc;
            HIP_TRY(hipMemcpy(objective + done, d_o, (size_t)c * 4, hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(violations + done * 8, d_v, (size_t)c * 32, hipMemcpyDeviceToHost));
        }
        return KAO_OK;
    }
};
}  // namespace

int kao_evaluate_batch(const kao_topic *t, const uint16_t *candidates, int64_t n, int32_t *objective, int32_t *violations) {
    if (!candidates || !objective || !violations || n < 1) return fail(KAO_ERR_INVALID, "null buffers");
    EvalCtx ctx;
    int rc = ctx.open(t);
    if (rc) return rc;
    return ctx.run(candidates, n, objective, violations);
}

int kao_evaluate(const kao_topic *t, const uint16_t *assignment, int64_t *objective, int32_t violations[8]) {
    int32_t obj = 0;
    int rc = kao_evaluate_batch(t, assignment, 1, &obj, violations);
    if (!rc && objective) *objective = obj;
    return rc;
}

int kao_canonicalize(const kao_topic *t, uint16_t *a) {
    if (!a) return fail(KAO_ERR_INVALID, "null assignment");
    if (t && (t->broker_w || t->broker_wl)) return KAO_OK;   // moving a replica to another broker changes the objective: nothing to canonicalise
    int rc = require_init();
    if (rc) return rc;
    PreparedTopic pt;
    if ((rc = prepare(t, 0, pt))) return rc;
    const TopicDev &d = pt.d;
    const int P = d.P, RF = d.RF, B = d.B;
    if (canon_lds_bytes(d.Bx) > 160 * 1024) return fail(KAO_ERR_UNSUPPORTED, "broker tables exceed 160 KiB of LDS");
    auto word = [&](uint16_t x) { return x == KAO_NONE ? kNoneW : ((uint32_t)x | ((uint32_t)(x / d.m) << 16)); };
    const int nw = d.nw;
    std::vector<uint32_t> cur_words((size_t)P * nw), a_words((size_t)P * nw, kNoneW);
    for (int p = 0; p < P; ++p) {
        const uint16_t *c = &pt.cur_int[(size_t)p * nw];
        for (int k = 0; k < nw; ++k) cur_words[(size_t)p * nw + k] = word(c[k]);
        for (int k = 0; k < RF; ++k) {
            const unsigned b = a[(size_t)p * RF + k];
            if (b >= (unsigned)B) return KAO_OK;  // an empty slot: infeasible, nothing to polish
            a_words[(size_t)p * nw + k] = word(pt.int_of[b]);
        }
    }
    // one device buffer: [TopicDev][status 16 B][cur words][A words][ext][rsz]
    const size_t wbytes = (size_t)P * nw * 4;
    const size_t o_status = align_up(sizeof(TopicDev)), o_cur = o_status + 256, o_a = o_cur + align_up(wbytes);
    const size_t o_ext = o_a + align_up(wbytes), o_rsz = o_ext + align_up(pt.ext_of.size() * 2);
    const size_t total = o_rsz + align_up(pt.rack_size.size() * 4);
    std::vector<unsigned char> stage(total, 0);
    std::memcpy(stage.data(), &d, sizeof(TopicDev));
    std::memcpy(stage.data() + o_cur, cur_words.data(), wbytes);
    std::memcpy(stage.data() + o_a, a_words.data(), wbytes);
    std::memcpy(stage.data() + o_ext, pt.ext_of.data(), pt.ext_of.size() * 2);
    std::memcpy(stage.data() + o_rsz, pt.rack_size.data(), pt.rack_size.size() * 4);
    void *dev = nullptr; size_t cap = 0;
    if ((rc = arena_get(total, &dev, &cap))) return rc;
    unsigned char *db = static_cast<unsigned char *>(dev);
    hipStream_t st = nullptr;
    if ((rc = stream_get(&st))) { arena_put(dev, cap, cur_device()); return rc; }
    int32_t status[2] = {0, 0};
    hipError_t e = hipMemcpyAsync(db, stage.data(), total, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        launch_canon(reinterpret_cast<const TopicDev *>(db), reinterpret_cast<const uint32_t *>(db + o_cur),
                     reinterpret_cast<const uint16_t *>(db + o_ext), reinterpret_cast<const int32_t *>(db + o_rsz),
                     reinterpret_cast<uint32_t *>(db + o_a), d.Bx, nw, reinterpret_cast<int32_t *>(db + o_status), st);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(a_words.data(), db + o_a, wbytes, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(status, db + o_status, sizeof status, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    stream_put(st, cur_device());
    arena_put(dev, cap, cur_device());
    if (e != hipSuccess) return fail(KAO_ERR_HIP, std::string("kao_canonicalize: ") + hipGetErrorString(e));
    if (!status[0]) return KAO_OK;  // only feasible assignments are polished
    for (int p = 0; p < P; ++p) {
        for (int k = 0; k < RF; ++k) a[(size_t)p * RF + k] = pt.ext_of[a_words[(size_t)p * nw + k] & 0xFFFFu];
    }
    for (int p = 0; p < P; ++p) {  // followers: retained ones in their current order, then new ones ascending
        std::vector<uint16_t> fol(a + (size_t)p * RF + 1, a + (size_t)p * RF + RF), kept, fresh;
        for (int k = 0; k < t->rf_cur; ++k) {
            const uint16_t c = t->current[(size_t)p * t->rf_cur + k];
            if (std::find(fol.begin(), fol.end(), c) != fol.end() && std::find(kept.begin(), kept.end(), c) == kept.end()) kept.push_back(c);
        }
        for (uint16_t f : fol) if (std::find(kept.begin(), kept.end(), f) == kept.end()) fresh.push_back(f);
        std::sort(fresh.begin(), fresh.end());
        kept.insert(kept.end(), fresh.begin(), fresh.end());
        std::copy(kept.begin(), kept.end(), a + (size_t)p * RF + 1);
    }
    return KAO_OK;
}

// ------------------------------------------------------------------------------------------------
// sessions
// ------------------------------------------------------------------------------------------------
void kao_session_destroy(kao_session *s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    if (s->stream_bound) (void)hipStreamSynchronize(s->stream_bound);
    if (s->ev_bound0) (void)hipEventDestroy(s->ev_bound0);
    if (s->ev_bound1) (void)hipEventDestroy(s->ev_bound1);
    if (s->ev_search) (void)hipEventDestroy(s->ev_search);
    if (s->stream_bound) (void)hipStreamDestroy(s->stream_bound);
    arena_put(s->arena_ro, s->arena_ro_bytes, s->device);
    arena_put(s->arena_rw, s->arena_rw_bytes, s->device);
    for (hipEvent_t e : s->ev) (void)hipEventDestroy(e);
    stream_put(s->stream, s->device);
    delete s;
}

int kao_session_create(const kao_topic *topics, int32_t n_topics, const kao_opts *opts_in, kao_session **out) {
    if (!out) return fail(KAO_ERR_INVALID, "null out");
    *out = nullptr;
    if (!topics || n_topics < 1) return fail(KAO_ERR_INVALID, "no topics");
    int rc = require_init();
    if (rc) return rc;
    kao_session *s = new kao_session();
    s->device = cur_device();
    s->n_topics = n_topics;
    const int g_num_cu = num_cu(s->device);
    kao_opts o{};
    if (opts_in) o = *opts_in;
    if (o.iters_per_launch <= 0) o.iters_per_launch = 512;
    if (o.obj_scale <= 0) o.obj_scale = 4;
    if (o.lam_min <= 0) o.lam_min = 1;
    if (o.lam_max <= 0) o.lam_max = 40;
    if (o.lam_max < o.lam_min) o.lam_max = o.lam_min;
    if (o.period_log2 < 0) o.period_log2 = 0;    // 0 = per topic, by size (auto_period_log2)
    if (o.period_log2 > 20) o.period_log2 = 20;
    if (o.time_limit_s <= 0) o.time_limit_s = 10.0;
    if (o.elite_period < 0) o.elite_period = 0;  // sessions: 0 = never (kao_solve picks its own default before creating the session)
    const bool auto_restarts = o.restarts <= 0;
    if (auto_restarts) {  // one full round of resident wavefronts (8 per SIMD = 32 per CU) across all topics
        const int want = g_num_cu * 32;
        int r = want / n_topics;
        r = (r / kWaves) * kWaves;
        o.restarts = std::min(std::max(r, 8), 8192);
        // large topics need depth (iterations per second) more than breadth: at most 2^22 replica slots over all the
        // restarts of the largest topic, but never fewer than one restart per compute unit
        int64_t slots = 1;
        for (int t = 0; t < n_topics; ++t) slots = std::max<int64_t>(slots, (int64_t)topics[t].n_partitions * std::max(topics[t].rf, 1));
        const int cap = (int)std::max<int64_t>(g_num_cu, (((int64_t)1 << 22) / slots) / kWaves * kWaves);
        o.restarts = std::min(o.restarts, cap);
    }
    if (o.restarts > (1 << 20) - 2) o.restarts = (1 << 20) - 2;  // id 0xFFFFF is reserved (kExternalRestart)
    {   // huge topics: bound the per-restart state in HBM (16 B of working words + the snapshot per partition):
        // 1 GB when the count was chosen automatically, 8 GB for an explicit request
        uint64_t per_restart = 0;
        for (int t = 0; t < n_topics; ++t) per_restart += (uint64_t)topics[t].n_partitions * (32 + 2 * (uint64_t)std::max(topics[t].rf, 1));
        const uint64_t cap = ((auto_restarts ? 1ull : 8ull) << 30) / std::max<uint64_t>(per_restart, 1);
        if ((uint64_t)o.restarts > cap) o.restarts = (int)std::max<uint64_t>(cap / kWaves * kWaves, kWaves);
    }
    s->opts = o;
    s->pts.resize((size_t)n_topics);
    s->topics.assign(topics, topics + n_topics);
    s->ub.resize((size_t)n_topics);

    std::vector<uint32_t> cur_pool, bw_pool, bwd_pool; std::vector<uint16_t> ext_pool, curd_pool, int_pool; std::vector<int32_t> rsz_pool;
    uint64_t price_i32 = 0;
    std::vector<uint8_t> rackof_pool;
    uint64_t state_bytes = 0, best_u16 = 0, win_u16 = 0, dual_i32 = 0, wide_slices = 0;
    s->topic_global.assign((size_t)n_topics, 0);
    for (int t = 0; t < n_topics; ++t) s->any_bw |= topics[t].broker_w || topics[t].broker_wl;
    int restart_base = 0;
    for (int t = 0; t < n_topics; ++t) {
        PreparedTopic &pt = s->pts[(size_t)t];
        const uint64_t seed = o.seed ^ ((uint64_t)(t + 1) * 0x9E3779B97F4A7C15ull);
        rc = prepare(&topics[t], seed, pt);
        if (rc) { kao_session_destroy(s); return rc; }
        s->ub[(size_t)t] = upper_bound_w(&topics[t]);
        s->topic_infeasible.push_back(infeasible_reason(&topics[t]).empty() ? 0 : 1);
        TopicDev &d = pt.d;
        d.n_restarts = o.restarts;
        d.period_log2 = o.period_log2 > 0 ? o.period_log2 : auto_period_log2(d.P, d.RF);
        d.restart_base = restart_base;
        restart_base += o.restarts;
        auto word = [&](uint16_t x) { return x == KAO_NONE ? kNoneW : ((uint32_t)x | ((uint32_t)(x / d.m) << 16)); };
        while (cur_pool.size() % 4) cur_pool.push_back(kNoneW);   // every topic's words start 16-byte aligned
        d.cur_off = (uint32_t)cur_pool.size();
        for (size_t i = 0; i < (size_t)d.P * d.nw; ++i) cur_pool.push_back(word(pt.cur_int[i]));  // LDS / register form of a replica: internal index | rack << 16
        const bool global_a = search_lds_bytes(d.P, d.Bx, 1, false, true, d.nw, s->any_bw) > 160 * 1024;
        s->topic_global[(size_t)t] = global_a;
        d.ext_off = (uint32_t)ext_pool.size();
        ext_pool.insert(ext_pool.end(), pt.ext_of.begin(), pt.ext_of.end());
        d.rsz_off = (uint32_t)rsz_pool.size();
        rsz_pool.insert(rsz_pool.end(), pt.rack_size.begin(), pt.rack_size.end());
        d.state_off = state_bytes;  // bytes: 8 per partition (packed, LDS path) or 16 (working words, global path)
        state_bytes += align_up((uint64_t)o.restarts * d.P * d.nw * (global_a ? 4 : 2));
        d.best_off = best_u16;
        best_u16 += (uint64_t)o.restarts * d.P * d.RF;
        d.win_off = (uint32_t)win_u16;
        win_u16 += (uint64_t)d.P * d.RF;
        d.rackof_off = (uint32_t)rackof_pool.size();
        rackof_pool.insert(rackof_pool.end(), pt.rack_of.begin(), pt.rack_of.end());
        d.curd_off = (uint32_t)curd_pool.size();
        curd_pool.insert(curd_pool.end(), pt.cur_dense.begin(), pt.cur_dense.end());
        if (d.has_bw) {
            d.bw_off = (uint32_t)bw_pool.size(); bw_pool.insert(bw_pool.end(), pt.bw_int.begin(), pt.bw_int.end());
            d.bwd_off = (uint32_t)bwd_pool.size(); bwd_pool.insert(bwd_pool.end(), pt.bw_dense.begin(), pt.bw_dense.end());
            s->priced = true;   // broker weights live in the tables of the priced K-search instantiation
            s->any_bw = true;
        }
        d.int_off = (uint32_t)int_pool.size();
        int_pool.insert(int_pool.end(), pt.int_of.begin(), pt.int_of.end());
        d.price_off = (uint32_t)price_i32;
        price_i32 += 2 * (uint64_t)d.B + kRackTab;
        dual_i32 = (dual_i32 + 1) & ~(uint64_t)1;  // the level-control words are 64-bit
        d.dual_off = (uint32_t)dual_i32;
        dual_i32 += 6 * (uint64_t)d.B + 3 * kRackTab + 8;
        d.cnt_off = (uint32_t)dual_i32;             // counters of the sliced K-bound live in the same pool (zeroed with it)
        dual_i32 += 2 * (uint64_t)d.B + kRackTab;
        wide_slices += (uint64_t)(d.P + 63) / 64;
        s->dual_ok.push_back(dual_supported(&topics[t]) ? 1 : 0);
        // algorithmic bytes: full evaluation = 2*RF*P + 2*rf_cur*P + B per candidate
        s->eval_bytes_per_launch += (uint64_t)o.restarts * (uint64_t)(2 * d.RF * d.P + 2 * d.rf_cur * d.P + d.B);
    }
    s->total_restarts = restart_base;
    // ---- launch groups: topics sorted by single-wave LDS need, a new group whenever the need doubles (<= 8 groups) ----
    std::vector<int> order((size_t)n_topics);
    for (int t = 0; t < n_topics; ++t) order[(size_t)t] = t;
    auto need1 = [&](int t) {  // topics kept in global memory sort last (their LDS need is tiny but they form their own groups)
        const TopicDev &d = s->pts[(size_t)t].d;
        const bool ga = s->topic_global[(size_t)t] != 0;
        return (ga ? ((size_t)1 << 40) : 0) + (d.nw > kRFP ? ((size_t)1 << 41) : 0) + search_lds_bytes(d.P, d.Bx, 1, ga, true, d.nw, s->any_bw);
    };
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return need1(x) < need1(y); });
    std::vector<std::vector<int>> members;
    size_t group_base = 0;
    for (int t : order) {
        if (members.empty() || (need1(t) > 2 * group_base && members.size() < 8)) { members.emplace_back(); group_base = need1(t); }
        members.back().push_back(t);
    }
    std::vector<int2> smap; std::vector<int4> emap;
    const int cpb = 32;  // candidates per K-eval workgroup
    for (const std::vector<int> &mem : members) {
        kao_session::LaunchGroup g;
        for (int t : mem) {
            const TopicDev &d = s->pts[(size_t)t].d;
            g.maxP = std::max(g.maxP, d.P); g.maxBx = std::max(g.maxBx, d.Bx); g.maxB = std::max(g.maxB, d.B);
        }
        g.global_a = s->topic_global[(size_t)mem[0]] != 0;
        g.nw = s->pts[(size_t)mem[0]].d.nw;
        while (g.waves > 1 && search_lds_bytes(g.maxP, g.maxBx, g.waves, g.global_a, true, g.nw, s->any_bw) > 160 * 1024) g.waves /= 2;
        g.cur_in_lds = eval_lds_bytes(g.maxP, g.maxB, true, g.nw) <= 160 * 1024;
        if (search_lds_bytes(g.maxP, g.maxBx, g.waves, g.global_a, true, g.nw, s->any_bw) > 160 * 1024 || eval_lds_bytes(g.maxP, g.maxB, g.cur_in_lds, g.nw) > 160 * 1024) {
            kao_session_destroy(s);
            return fail(KAO_ERR_UNSUPPORTED, "broker tables exceed 160 KiB of LDS (about 30,000 padded brokers)");
        }
        std::vector<int2> gs; std::vector<int> gs_topic;
        std::vector<int4> ge; std::vector<int> ge_topic;
        for (int t : mem) {
            const TopicDev &d = s->pts[(size_t)t].d;
            for (int r = 0; r < d.n_restarts; r += g.waves) { gs.push_back(make_int2(t, r)); gs_topic.push_back(t); }
            for (int r = 0; r < d.n_restarts; r += cpb) {
                ge.push_back(make_int4(t, r, std::min(cpb, d.n_restarts - r), d.restart_base + r));
                ge_topic.push_back(t);
            }
        }
        gs = xcd_order(gs, gs_topic);
        ge = xcd_order(ge, ge_topic);
        g.smap_off = (int)smap.size(); g.smap_n = (int)gs.size();
        g.emap_off = (int)emap.size(); g.emap_n = (int)ge.size();
        smap.insert(smap.end(), gs.begin(), gs.end());
        emap.insert(emap.end(), ge.begin(), ge.end());
        s->groups.push_back(g);
    }
    s->blocks_search = (int)smap.size();
    s->blocks_eval = (int)emap.size();
    std::vector<TopicDev> tds;
    for (auto &pt : s->pts) tds.push_back(pt.d);

    // ---- read-only arena: stage everything on the host, ONE hipMalloc (or a parked arena), ONE H2D copy ----
    struct Sec { const void *src; size_t bytes; size_t off; };
    Sec secs[11] = {{tds.data(), tds.size() * sizeof(TopicDev), 0}, {smap.data(), smap.size() * sizeof(int2), 0},
                   {emap.data(), emap.size() * sizeof(int4), 0}, {cur_pool.data(), cur_pool.size() * sizeof(uint32_t), 0},
                   {ext_pool.data(), ext_pool.size() * 2, 0}, {rsz_pool.data(), rsz_pool.size() * 4, 0},
                   {rackof_pool.data(), rackof_pool.size(), 0}, {curd_pool.data(), curd_pool.size() * 2, 0},
                   {int_pool.data(), int_pool.size() * 2, 0}, {bw_pool.data(), bw_pool.size() * 4, 0},
                   {bwd_pool.data(), bwd_pool.size() * 4, 0}};
    size_t ro_bytes = 0;
    for (Sec &sec : secs) { sec.off = ro_bytes; ro_bytes += align_up(sec.bytes); }
    std::vector<unsigned char> stage(ro_bytes);
    for (const Sec &sec : secs) if (sec.bytes) std::memcpy(stage.data() + sec.off, sec.src, sec.bytes);
    if ((rc = arena_get(ro_bytes, &s->arena_ro, &s->arena_ro_bytes))) { kao_session_destroy(s); return rc; }
    unsigned char *ro = static_cast<unsigned char *>(s->arena_ro);
    s->d_topics = reinterpret_cast<TopicDev *>(ro + secs[0].off);
    s->d_smap = reinterpret_cast<int2 *>(ro + secs[1].off);
    s->d_emap = reinterpret_cast<int4 *>(ro + secs[2].off);
    s->d_cur = reinterpret_cast<uint32_t *>(ro + secs[3].off);
    s->d_ext = reinterpret_cast<uint16_t *>(ro + secs[4].off);
    s->d_rsz = reinterpret_cast<int32_t *>(ro + secs[5].off);
    s->d_rackof = reinterpret_cast<uint8_t *>(ro + secs[6].off);
    s->d_curd = reinterpret_cast<uint16_t *>(ro + secs[7].off);
    s->d_int = reinterpret_cast<uint16_t *>(ro + secs[8].off);
    s->d_bw = reinterpret_cast<uint32_t *>(ro + secs[9].off);
    s->d_bwd = reinterpret_cast<uint32_t *>(ro + secs[10].off);

    // ---- mutable arena ----
    const size_t state_b = align_up(state_bytes), best_b = align_up(best_u16 * 2);
    const size_t info_b = align_up((size_t)s->total_restarts * 16), obj_b = align_up((size_t)s->total_restarts * 4);
    const size_t viol_b = align_up((size_t)s->total_restarts * 32);
    s->rb_viol_off = align_up((size_t)n_topics * 8 + 16, 16);
    s->rb_assign_off = s->rb_viol_off + (size_t)n_topics * 32;
    s->readback_bytes = s->rb_assign_off + win_u16 * 2;
    // behind the per-topic blocks: control blocks (8 x int64 per topic) and the workgroup map of the sliced K-bound
    dual_i32 = (dual_i32 + 1) & ~(uint64_t)1;
    s->wide_ctl_i32 = dual_i32; dual_i32 += 16 * (uint64_t)n_topics;
    s->wide_map_i32 = dual_i32; dual_i32 += 2 * wide_slices;
    const size_t dual_b = align_up(dual_i32 * 4), dtarget_b = align_up((size_t)n_topics * 8), dids_b = align_up((size_t)n_topics * 4);
    s->dual_rb_bytes = (size_t)n_topics * 24;
    s->price_half_i32 = align_up(price_i32 * 4) / 4;
    const size_t price_b = 2 * s->price_half_i32 * 4;
    const size_t rw_bytes = state_b + best_b + info_b + obj_b + viol_b + align_up(s->readback_bytes) + dual_b + dtarget_b + dids_b +
                            align_up(s->dual_rb_bytes) + price_b + align_up((size_t)n_topics * 8);
    if ((rc = arena_get(rw_bytes, &s->arena_rw, &s->arena_rw_bytes))) { kao_session_destroy(s); return rc; }
    unsigned char *rw = static_cast<unsigned char *>(s->arena_rw);
    s->d_state = rw;
    s->d_best = reinterpret_cast<uint16_t *>(rw + state_b);
    s->d_info = reinterpret_cast<int32_t *>(rw + state_b + best_b);
    s->d_obj = reinterpret_cast<int32_t *>(rw + state_b + best_b + info_b);
    s->d_viol = reinterpret_cast<int32_t *>(rw + state_b + best_b + info_b + obj_b);
    s->d_readback = rw + state_b + best_b + info_b + obj_b + viol_b;
    s->d_keys = reinterpret_cast<unsigned long long *>(s->d_readback);
    s->d_drift = reinterpret_cast<int32_t *>(s->d_readback + (size_t)n_topics * 8);
    s->d_win_viol = reinterpret_cast<int32_t *>(s->d_readback + s->rb_viol_off);
    s->d_win_assign = reinterpret_cast<uint16_t *>(s->d_readback + s->rb_assign_off);
    s->h_readback.assign(s->readback_bytes, 0);
    {
        unsigned char *q = s->d_readback + align_up(s->readback_bytes);
        s->d_dual = reinterpret_cast<int32_t *>(q); q += dual_b;
        s->d_dual_target = reinterpret_cast<long long *>(q); q += dtarget_b;
        s->d_dual_ids = reinterpret_cast<int32_t *>(q); q += dids_b;
        s->d_dual_rb = q; q += align_up(s->dual_rb_bytes);
        s->d_price = reinterpret_cast<int32_t *>(q); q += price_b;
        s->d_keys_glob = reinterpret_cast<unsigned long long *>(q);
        s->dual_bytes = dual_b;
        s->dual_flags.assign((size_t)n_topics, 0);
        s->dual_iters.assign((size_t)n_topics, 0);
        for (int t = 0; t < n_topics; ++t) if (!s->dual_ok[(size_t)t]) s->dual_flags[(size_t)t] = 8;
    }

    if ((rc = stream_get(&s->stream))) { kao_session_destroy(s); return rc; }
    // restart states / info / obj / viol are fully written by launch 0 (init) and the first K-eval; only the
    // snapshots ("no snapshot" = all KAO_NONE), the keys (all ones) and the drift counter need initial values
    hipError_t e1 = hipMemcpyAsync(ro, stage.data(), ro_bytes, hipMemcpyHostToDevice, s->stream);
    hipError_t e2 = hipMemsetAsync(s->d_best, 0xFF, best_u16 * 2 ? best_u16 * 2 : 2, s->stream);
    hipError_t e3 = hipMemsetAsync(s->d_readback, 0xFF, (size_t)n_topics * 8, s->stream);
    hipError_t e4 = hipMemsetAsync(s->d_drift, 0, 16, s->stream);
    if (e4 == hipSuccess) e4 = hipMemsetAsync(s->d_price, 0, price_b ? price_b : 4, s->stream);  // no prices yet
    hipError_t e5 = hipStreamSynchronize(s->stream);  // `stage` is pageable host memory and goes out of scope
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess || e4 != hipSuccess || e5 != hipSuccess) {
        kao_session_destroy(s);
        return fail(KAO_ERR_HIP, "session upload failed");
    }
    if (o.profile) {
        s->ev.resize(kEvRing * 3);
        for (auto &e : s->ev) if (hipEventCreate(&e) != hipSuccess) { kao_session_destroy(s); return fail(KAO_ERR_HIP, "hipEventCreate"); }
    }
    *out = s;
    return KAO_OK;
}

int kao_session_step(kao_session *s) {
    if (!s) return fail(KAO_ERR_INVALID, "null session");
    HIP_TRY(hipSetDevice(s->device));
    const bool prof = s->opts.profile != 0;
    if (prof && s->ev_pending == kEvRing) {
        int rc = session_drain_events(s);
        if (rc) return rc;
    }
    SearchPools sp{};
    sp.topics = s->d_topics; sp.block_map = s->d_smap; sp.cur_pool = s->d_cur; sp.ext_pool = s->d_ext; sp.rsz_pool = s->d_rsz;
    sp.state_pool = s->d_state; sp.best_pool = s->d_best; sp.restart_info = s->d_info; sp.drift = s->d_drift;
    SearchParams prm{};
    prm.obj_scale = s->opts.obj_scale; prm.lam_min = s->opts.lam_min; prm.lam_max = s->opts.lam_max;
    prm.launch = s->launch; prm.iters = (uint32_t)s->opts.iters_per_launch; prm.init = s->launch == 0 ? 1 : 0;
    const int eper = s->opts.elite_period;
    prm.bw = s->any_bw ? 1 : 0;
    prm.elite = (eper > 0 && s->launch > 0 && s->launch % (uint32_t)eper == 0) ? 1 : 0;
    sp.price_pool = s->d_price + (size_t)s->price_read * s->price_half_i32;
    sp.int_pool = s->d_int; sp.elite_assign = s->d_win_assign; sp.elite_key = s->d_keys; sp.bw_pool = s->d_bw;
    EvalPools ep{};
    ep.topics = s->d_topics; ep.rackof_pool = s->d_rackof; ep.curd_pool = s->d_curd;
    ep.cand = s->d_best; ep.objective = s->d_obj; ep.violations = s->d_viol; ep.best_key = s->d_keys; ep.bwd_pool = s->d_bwd;
    hipEvent_t *e = prof ? &s->ev[(size_t)s->ev_pending * 3] : nullptr;
    if (prof) HIP_TRY(hipEventRecord(e[0], s->stream));
    for (const kao_session::LaunchGroup &g : s->groups) {
        sp.block_map = s->d_smap + g.smap_off;
        prm.maxP = g.maxP; prm.maxBx = g.maxBx;
        launch_search(sp, prm, g.smap_n, g.waves, g.global_a, s->priced, g.nw, s->stream);
        HIP_TRY(hipGetLastError());
    }
    if (prof) HIP_TRY(hipEventRecord(e[1], s->stream));
    for (const kao_session::LaunchGroup &g : s->groups) {
        ep.block_map = s->d_emap + g.emap_off;
        ep.maxP = g.maxP; ep.maxB = g.maxB; ep.cur_in_lds = g.cur_in_lds ? 1 : 0;
        launch_eval(ep, g.emap_n, g.nw, s->stream);
        HIP_TRY(hipGetLastError());
    }
    if (prof) { HIP_TRY(hipEventRecord(e[2], s->stream)); s->ev_pending++; }
    if (eper > 0 && (s->launch + 1) % (uint32_t)eper == 0) {  // the next launch is an elite launch: stage every topic's best assignment
        launch_gather(s->d_topics, s->n_topics, s->d_keys, s->d_best, s->d_viol, s->d_win_assign, s->d_win_viol, s->stream);
        HIP_TRY(hipGetLastError());
    }
    for (const PreparedTopic &pt : s->pts) {
        const uint64_t n = neighbours_in_range(prm.launch * prm.iters, prm.iters, pt.d.RF, pt.d.B, pt.d.P) * (uint64_t)pt.d.n_restarts;
        s->delta_total += n;
        s->search_bytes_total += n * (uint64_t)(8 * pt.d.RF + 10);
    }
    s->launch++;
    return KAO_OK;
}

int kao_session_sync(kao_session *s) {
    if (!s) return fail(KAO_ERR_INVALID, "null session");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipStreamSynchronize(s->stream));
    if (s->opts.profile) return session_drain_events(s);
    return KAO_OK;
}

int kao_session_best_keys(kao_session *s, uint64_t *keys) {
    if (!s || !keys) return fail(KAO_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipMemcpyAsync(keys, s->d_keys, (size_t)s->n_topics * 8, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    return KAO_OK;
}

int kao_session_device_keys(kao_session *s, void **d_keys) {
    if (!s || !d_keys) return fail(KAO_ERR_INVALID, "null argument");
    *d_keys = s->d_keys;
    return KAO_OK;
}

int kao_session_best(kao_session *s, kao_result *results) {
    if (!s || !results) return fail(KAO_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(s->device));
    launch_gather(s->d_topics, s->n_topics, s->d_keys, s->d_best, s->d_viol, s->d_win_assign, s->d_win_viol, s->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(s->h_readback.data(), s->d_readback, s->readback_bytes, hipMemcpyDeviceToHost, s->stream));
    int rc = kao_session_sync(s);
    if (rc) return rc;
    const uint64_t *keys = reinterpret_cast<const uint64_t *>(s->h_readback.data());
    const int32_t *wv = reinterpret_cast<const int32_t *>(s->h_readback.data() + s->rb_viol_off);
    const uint16_t *wa = reinterpret_cast<const uint16_t *>(s->h_readback.data() + s->rb_assign_off);
    for (int t = 0; t < s->n_topics; ++t) {
        const TopicDev &d = s->pts[(size_t)t].d;
        kao_result &r = results[t];
        r.upper_bound = s->ub[(size_t)t];
        const uint64_t key = keys[t];
        if (key == ~0ull) {  // no step has run yet
            r.status = s->topic_infeasible[(size_t)t] ? KAO_STATUS_INFEASIBLE_PROVEN : KAO_STATUS_NO_FEASIBLE; r.best_restart = -1; r.objective = -1;
            std::memset(r.violations, 0, sizeof r.violations);
            continue;
        }
        r.best_restart = (key & 0xFFFFF) == kExternalRestart ? -1 : (int)(key & 0xFFFFF);  // -1: adopted from another GPU
        r.objective = (int64_t)kObjCap - (int64_t)((key >> 20) & 0xFFFFFF);
        std::memcpy(r.violations, wv + (size_t)t * 8, 32);
        if (r.assignment) std::memcpy(r.assignment, wa + d.win_off, (size_t)d.P * d.RF * 2);
        if (r.violations[0] != 0) { r.status = s->topic_infeasible[(size_t)t] ? KAO_STATUS_INFEASIBLE_PROVEN : KAO_STATUS_NO_FEASIBLE; r.objective = -1; }
        else r.status = r.objective >= r.upper_bound ? KAO_STATUS_OPTIMAL_PROVEN : KAO_STATUS_FEASIBLE_BOUND_GAP;
    }
    return KAO_OK;
}

int kao_session_stats(kao_session *s, kao_stats *out) {
    if (!s || !out) return fail(KAO_ERR_INVALID, "null argument");
    int rc = kao_session_sync(s);
    if (rc) return rc;
    std::memset(out, 0, sizeof *out);
    out->launches = s->launch;
    out->delta_candidates = s->delta_total;
    out->full_candidates = (uint64_t)s->launch * (uint64_t)s->total_restarts;
    out->ms_search = s->ms_search; out->ms_eval = s->ms_eval;
    out->search_bytes_algo = s->search_bytes_total;
    out->eval_bytes_algo = s->eval_bytes_per_launch * s->launch;
    out->n_restarts_total = s->total_restarts;
    for (const kao_session::LaunchGroup &g : s->groups)
        out->lds_bytes_search = std::max(out->lds_bytes_search, (int32_t)search_lds_bytes(g.maxP, g.maxBx, g.waves, g.global_a, s->priced, g.nw, s->any_bw));
    out->launch_groups = (int32_t)s->groups.size();
    out->blocks_search = s->blocks_search;
    HIP_TRY(hipMemcpy(&out->drift, s->d_drift, 4, hipMemcpyDeviceToHost));
    return KAO_OK;
}

int kao_session_bound_step(kao_session *s, const int64_t *target, int32_t iters) {
    if (!s || !target) return fail(KAO_ERR_INVALID, "null argument");
    if (iters < 1) return fail(KAO_ERR_INVALID, "iters < 1");
    HIP_TRY(hipSetDevice(s->device));
    // the previous launch's H2D copies read the staging vectors below: wait for them before rewriting
    if (s->stream_bound) HIP_TRY(hipStreamSynchronize(s->stream_bound));
    s->h_dual_ids.clear();
    s->h_dual_target.assign((size_t)s->n_topics, -1);
    int maxB = 0, maxP = 0, maxR = 0;
    for (int t = 0; t < s->n_topics; ++t) {
        if (target[t] < 0 || !s->dual_ok[(size_t)t] || s->topic_infeasible[(size_t)t]) continue;
        if (target[t] > (int64_t)1 << 40) return fail(KAO_ERR_INVALID, "target out of range");
        s->h_dual_ids.push_back(t);
        s->h_dual_target[(size_t)t] = target[t];
        maxB = std::max(maxB, s->pts[(size_t)t].d.B);
        maxP = std::max(maxP, s->pts[(size_t)t].d.P);
        maxR = std::max(maxR, s->pts[(size_t)t].d.R);
    }
    if (s->h_dual_ids.empty()) return KAO_OK;
    if (!s->stream_bound) {
        // highest priority: a K-bound launch is a handful of workgroups that should not queue behind a full K-search grid
        int lo = 0, hi = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
        HIP_TRY(hipStreamCreateWithPriority(&s->stream_bound, hipStreamNonBlocking, hi));
        HIP_TRY(hipEventCreate(&s->ev_bound0));
        HIP_TRY(hipEventCreate(&s->ev_bound1));
    }
    // one K-bound launch in flight at a time (it continues from the state the previous one left in HBM); the session
    // upload was synchronised at creation, K-search and K-bound share read-only tables only
    HIP_TRY(hipStreamSynchronize(s->stream_bound));
    if (s->bound_launches == 0) {
        // K-bound state is initialised by the first launch only (most sessions never need K-bound): multipliers and
        // directions 0, best dual value "infinite" (0x7F7F...), info 0
        HIP_TRY(hipMemsetAsync(s->d_dual, 0, s->dual_bytes, s->stream_bound));
        HIP_TRY(hipMemsetAsync(s->d_dual_rb, 0x7F, (size_t)s->n_topics * 8, s->stream_bound));
        HIP_TRY(hipMemsetAsync(s->d_dual_rb + (size_t)s->n_topics * 8, 0, (size_t)s->n_topics * 16, s->stream_bound));
    }
    // pageable staging: hipMemcpyAsync returns once the host buffers have been consumed
    HIP_TRY(hipMemcpyAsync(s->d_dual_target, s->h_dual_target.data(), (size_t)s->n_topics * 8, hipMemcpyHostToDevice, s->stream_bound));
    HIP_TRY(hipMemcpyAsync(s->d_dual_ids, s->h_dual_ids.data(), s->h_dual_ids.size() * 4, hipMemcpyHostToDevice, s->stream_bound));
    BoundPools bp{};
    bp.topics = s->d_topics; bp.ids = s->d_dual_ids; bp.rackof_pool = s->d_rackof; bp.curd_pool = s->d_curd;
    bp.dual_pool = s->d_dual; bp.target = s->d_dual_target;
    bp.best_L = reinterpret_cast<long long *>(s->d_dual_rb);
    bp.info = reinterpret_cast<int32_t *>(s->d_dual_rb + (size_t)s->n_topics * 8);
    bp.ext_pool = s->d_ext; bp.rsz_pool = s->d_rsz;
    bp.iters = iters; bp.maxB = maxB; bp.maxP = maxP; bp.maxR = maxR;
    bp.cur_in_lds = bound_lds_bytes(maxB, maxP, maxR, true) <= 160 * 1024 ? 1 : 0;
    if (s->priced) {  // K-search launches already enqueued may still read the half this launch is about to overwrite
        if (!s->ev_search) HIP_TRY(hipEventCreateWithFlags(&s->ev_search, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(s->ev_search, s->stream));
        HIP_TRY(hipStreamWaitEvent(s->stream_bound, s->ev_search, 0));
    }
    // prices go into the half K-search is NOT reading; kao_session_adopt_prices flips the halves once this launch is done
    const int wh = s->price_read ^ 1;
    if (s->price_write_last >= 0 && s->price_write_last != wh)  // keep the prices of topics this launch does not cover
        HIP_TRY(hipMemcpyAsync(s->d_price + (size_t)wh * s->price_half_i32, s->d_price + (size_t)(wh ^ 1) * s->price_half_i32,
                               s->price_half_i32 * 4, hipMemcpyDeviceToDevice, s->stream_bound));
    bp.price_pool = s->d_price + (size_t)wh * s->price_half_i32;
    bp.export_prices = 1;
    if (const char *e = std::getenv("KAO_X_PRICE_SRC")) bp.export_prices = std::atoi(e);  // experiment knob
    s->price_write_last = wh;
    // lanes own partitions, wavefronts own racks when the pools are rebuilt: enough wavefronts for either, at most 16
    const int waves = std::min(16, std::max({1, (maxP + 63) / 64, std::min(maxR, 8)}));
    // topics beyond a few thousand partitions: one iteration per launch, the partitions sliced over several workgroups
    // (k_bound_step); a launch that holds such a topic runs all its topics that way.  KAO_BOUND_CHUNK = partitions per
    // slice (test hook: small values slice small topics)
    int chunk = maxP > 2048 ? 512 : 0;
    if (const char *e = std::getenv("KAO_BOUND_CHUNK")) chunk = std::max(0, std::atoi(e)) / 64 * 64;
    HIP_TRY(hipEventRecord(s->ev_bound0, s->stream_bound));
    if (chunk > 0) {
        s->h_wide_map.clear();
        for (int t : s->h_dual_ids)
            for (int sl = 0, n = (s->pts[(size_t)t].d.P + chunk - 1) / chunk; sl < n; ++sl) s->h_wide_map.push_back(make_int2(t, sl));
        BoundWide wd{};
        wd.map = reinterpret_cast<const int2 *>(s->d_dual + s->wide_map_i32);
        wd.cnt_pool = s->d_dual;
        wd.ctl = reinterpret_cast<long long *>(s->d_dual + s->wide_ctl_i32);
        wd.chunk = chunk;
        HIP_TRY(hipMemcpyAsync(s->d_dual + s->wide_map_i32, s->h_wide_map.data(), s->h_wide_map.size() * sizeof(int2), hipMemcpyHostToDevice,
                               s->stream_bound));
        launch_bound_wide(bp, wd, (int)s->h_dual_ids.size(), (int)s->h_wide_map.size(), 16, s->stream_bound);
    } else
        launch_bound(bp, (int)s->h_dual_ids.size(), waves, s->stream_bound);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(s->ev_bound1, s->stream_bound));
    s->bound_inflight = true;
    s->bound_iters_last = iters;
    s->bound_launches++;
    return KAO_OK;
}

int kao_session_set_prices(kao_session *s, int32_t topic, const int32_t *a, const int32_t *l, const int32_t *g) {
    if (!s || topic < 0 || topic >= s->n_topics || !a || !l || !g) return fail(KAO_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(s->device));
    const TopicDev &d = s->pts[(size_t)topic].d;
    std::vector<int32_t> buf(2 * (size_t)d.B + kRackTab, 0);
    std::memcpy(buf.data(), a, (size_t)d.B * 4);
    std::memcpy(buf.data() + d.B, l, (size_t)d.B * 4);
    std::memcpy(buf.data() + 2 * (size_t)d.B, g, (size_t)d.R * 4);
    // both halves, so that a later adopt (which flips them) keeps host-set prices of topics K-bound does not cover
    HIP_TRY(hipStreamSynchronize(s->stream));
    if (s->stream_bound) HIP_TRY(hipStreamSynchronize(s->stream_bound));
    for (int h = 0; h < 2; ++h)
        HIP_TRY(hipMemcpy(s->d_price + (size_t)h * s->price_half_i32 + d.price_off, buf.data(), buf.size() * 4, hipMemcpyHostToDevice));
    s->priced = true;
    return KAO_OK;
}

int kao_session_prices(kao_session *s, int32_t topic, int32_t *a, int32_t *l, int32_t *g) {
    if (!s || topic < 0 || topic >= s->n_topics) return fail(KAO_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(s->device));
    const TopicDev &d = s->pts[(size_t)topic].d;
    const int32_t *base = s->d_price + (size_t)s->price_read * s->price_half_i32 + d.price_off;
    HIP_TRY(hipStreamSynchronize(s->stream));
    if (a) HIP_TRY(hipMemcpy(a, base, (size_t)d.B * 4, hipMemcpyDeviceToHost));
    if (l) HIP_TRY(hipMemcpy(l, base + d.B, (size_t)d.B * 4, hipMemcpyDeviceToHost));
    if (g) HIP_TRY(hipMemcpy(g, base + 2 * (size_t)d.B, (size_t)d.R * 4, hipMemcpyDeviceToHost));
    return KAO_OK;
}

int kao_session_adopt_prices(kao_session *s) {
    if (!s) return fail(KAO_ERR_INVALID, "null session");
    if (s->price_write_last < 0) return KAO_OK;  // K-bound has not run: nothing to adopt
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipStreamSynchronize(s->stream_bound));
    s->price_read = s->price_write_last;
    s->priced = true;
    return KAO_OK;
}

int kao_session_bound_busy(kao_session *s) {
    if (!s) return fail(KAO_ERR_INVALID, "null session");
    HIP_TRY(hipSetDevice(s->device));
    if (!s->bound_inflight) return 0;
    const hipError_t e = hipEventQuery(s->ev_bound1);
    if (e == hipErrorNotReady) return 1;
    if (e != hipSuccess) return fail(KAO_ERR_HIP, std::string("hipEventQuery: ") + hipGetErrorString(e));
    return 0;
}

int kao_session_bounds(kao_session *s, int64_t *upper_bound, int32_t *flags, int32_t *iters) {
    if (!s) return fail(KAO_ERR_INVALID, "null session");
    HIP_TRY(hipSetDevice(s->device));
    if (s->bound_launches) {
        std::vector<unsigned char> rb(s->dual_rb_bytes);
        HIP_TRY(hipMemcpyAsync(rb.data(), s->d_dual_rb, s->dual_rb_bytes, hipMemcpyDeviceToHost, s->stream_bound));
        HIP_TRY(hipStreamSynchronize(s->stream_bound));
        if (s->bound_inflight) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, s->ev_bound0, s->ev_bound1) == hipSuccess) s->bound_ms_last = ms;
            s->bound_inflight = false;
        }
        const int64_t *best = reinterpret_cast<const int64_t *>(rb.data());
        const int32_t *info = reinterpret_cast<const int32_t *>(rb.data() + (size_t)s->n_topics * 8);
        for (int t = 0; t < s->n_topics; ++t) {
            if (!s->dual_ok[(size_t)t]) continue;
            s->dual_iters[(size_t)t] = info[t * 4 + 0];
            s->dual_flags[(size_t)t] = info[t * 4 + 1];
            if ((info[t * 4 + 1] & 4) || info[t * 4 + 0] == 0 || best[t] >= (int64_t)0x7F7F7F7F7F7F7F7Fll) continue;
            const int64_t b = best[t] >= 0 ? best[t] / kDualScale : -((-best[t] + kDualScale - 1) / kDualScale);  // floor
            s->ub[(size_t)t] = std::min(s->ub[(size_t)t], b);
        }
    }
    for (int t = 0; t < s->n_topics; ++t) {
        if (upper_bound) upper_bound[t] = s->ub[(size_t)t];
        if (flags) flags[t] = s->dual_flags[(size_t)t];
        if (iters) iters[t] = s->dual_iters[(size_t)t];
    }
    return KAO_OK;
}

int kao_session_dual_state(kao_session *s, int32_t topic, int32_t *a, int32_t *l, int32_t *g, int64_t *best_dual) {
    if (!s || topic < 0 || topic >= s->n_topics) return fail(KAO_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(s->device));
    const TopicDev &d = s->pts[(size_t)topic].d;
    HIP_TRY(hipStreamSynchronize(s->stream));
    if (s->stream_bound) HIP_TRY(hipStreamSynchronize(s->stream_bound));
    const int32_t *base = s->d_dual + d.dual_off;
    if (s->bound_launches == 0) {  // no K-bound launch yet: the initial state
        if (a) std::memset(a, 0, (size_t)d.B * 4);
        if (l) std::memset(l, 0, (size_t)d.B * 4);
        if (g) std::memset(g, 0, (size_t)d.R * 4);
        if (best_dual) *best_dual = (int64_t)0x7F7F7F7F7F7F7F7Fll;
        return KAO_OK;
    }
    if (a) HIP_TRY(hipMemcpy(a, base, (size_t)d.B * 4, hipMemcpyDeviceToHost));
    if (l) HIP_TRY(hipMemcpy(l, base + d.B, (size_t)d.B * 4, hipMemcpyDeviceToHost));
    if (g) HIP_TRY(hipMemcpy(g, base + 4 * (size_t)d.B, (size_t)d.R * 4, hipMemcpyDeviceToHost));
    if (best_dual) HIP_TRY(hipMemcpy(best_dual, s->d_dual_rb + (size_t)topic * 8, 8, hipMemcpyDeviceToHost));
    return KAO_OK;
}

int kao_dual_bound(const kao_topic *t, int64_t target, int32_t iters, int32_t launches, int64_t *bound, int64_t *best_dual,
                   int32_t *iters_done, int32_t *flags, int32_t *multipliers) {
    if (!t) return fail(KAO_ERR_INVALID, "null topic");
    if (target < 0 || iters < 1 || launches < 1) return fail(KAO_ERR_INVALID, "bad target / iters / launches");
    kao_opts o{};
    o.restarts = kWaves;  // no search is run: the smallest session there is
    kao_session *s = nullptr;
    int rc = kao_session_create(t, 1, &o, &s);
    if (rc) return rc;
    if (!s->dual_ok[0]) { kao_session_destroy(s); return fail(KAO_ERR_UNSUPPORTED, "topic outside K-bound's limits"); }
    int32_t fl = 0, itn = 0;
    for (int i = 0; i < launches && !rc; ++i) {
        rc = kao_session_bound_step(s, &target, iters);
        if (!rc) rc = kao_session_bounds(s, nullptr, &fl, &itn);
        if (fl & 7) break;
    }
    int64_t bd = 0;
    if (!rc) rc = kao_session_dual_state(s, 0, multipliers, multipliers ? multipliers + t->n_brokers : nullptr,
                                         multipliers ? multipliers + 2 * (size_t)t->n_brokers : nullptr, &bd);
    if (!rc) {
        if (best_dual) *best_dual = bd;
        if (bound) *bound = (fl & 4) || itn == 0 ? INT64_MAX : (bd >= 0 ? bd / kDualScale : -((-bd + kDualScale - 1) / kDualScale));
        if (iters_done) *iters_done = itn;
        if (flags) *flags = fl;
    }
    kao_session_destroy(s);
    return rc;
}

int kao_session_restart_state(kao_session *s, int32_t topic, int32_t restart, uint16_t *final_state, uint16_t *best_state,
                              int32_t info[4]) {
    if (!s || topic < 0 || topic >= s->n_topics) return fail(KAO_ERR_INVALID, "bad topic");
    const PreparedTopic &pt = s->pts[(size_t)topic];
    const TopicDev &d = pt.d;
    if (restart < 0 || restart >= d.n_restarts) return fail(KAO_ERR_INVALID, "bad restart");
    HIP_TRY(hipSetDevice(s->device));
    int rc = kao_session_sync(s);
    if (rc) return rc;
    if (final_state) {
        const bool ga = s->topic_global[(size_t)topic] != 0;
        const int nw = d.nw;
        std::vector<uint16_t> raw((size_t)d.P * nw * (ga ? 2 : 1));   // global path: nw words per partition; LDS path: nw x u16
        HIP_TRY(hipMemcpy(raw.data(), s->d_state + d.state_off + (uint64_t)restart * d.P * nw * (ga ? 4 : 2), raw.size() * 2, hipMemcpyDeviceToHost));
        for (int p = 0; p < d.P; ++p)
            for (int k = 0; k < d.RF; ++k) {
                // global path: word = x | rack << 16 (little endian: x is the low half; none = all ones); LDS path: the index itself
                const uint16_t x = ga ? raw[((size_t)p * nw + k) * 2] : raw[(size_t)p * nw + k];
                final_state[(size_t)p * d.RF + k] = x < pt.ext_of.size() ? pt.ext_of[x] : (uint16_t)KAO_NONE;
            }
    }
    if (best_state)
        HIP_TRY(hipMemcpy(best_state, s->d_best + d.best_off + (uint64_t)restart * d.P * d.RF, (size_t)d.P * d.RF * 2, hipMemcpyDeviceToHost));
    if (info) HIP_TRY(hipMemcpy(info, s->d_info + (size_t)(d.restart_base + restart) * 4, 16, hipMemcpyDeviceToHost));
    return KAO_OK;
}

}  // extern "C"

namespace {

// One whole job on one device: the kao_solve loop, cut into steps so that kao_solve_multi can drive several of them in
// lockstep from one host thread (launches of all devices are enqueued before any of them is waited for).
// the topic's winning assignment (dense [P*RF]) as of the last finished launch
int session_topic_best(kao_session *s, int i, uint16_t *out) {
    HIP_TRY(hipSetDevice(s->device));
    launch_gather(s->d_topics, s->n_topics, s->d_keys, s->d_best, s->d_viol, s->d_win_assign, s->d_win_viol, s->stream);
    HIP_TRY(hipGetLastError());
    const TopicDev &d = s->pts[(size_t)i].d;
    HIP_TRY(hipMemcpyAsync(out, s->d_win_assign + d.win_off, (size_t)d.P * d.RF * 2, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    return KAO_OK;
}
// a feasible assignment found outside K-search (KAO-CX) becomes the topic's incumbent: winner buffer + packed key with the
// reserved restart id, exactly as an assignment adopted from another GPU (k_gather leaves it alone, elite launches re-seed from it)
int session_adopt_external(kao_session *s, int i, const uint16_t *assign, int64_t objective, uint64_t *key_out) {
    HIP_TRY(hipSetDevice(s->device));
    const TopicDev &d = s->pts[(size_t)i].d;
    const uint64_t key = ((uint64_t)((int64_t)kObjCap - objective) << 20) | (uint64_t)kExternalRestart;
    HIP_TRY(hipStreamSynchronize(s->stream));
    HIP_TRY(hipMemcpyAsync(s->d_win_assign + d.win_off, assign, (size_t)d.P * d.RF * 2, hipMemcpyHostToDevice, s->stream));
    HIP_TRY(hipMemcpyAsync(s->d_keys + i, &key, 8, hipMemcpyHostToDevice, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    *key_out = key;
    return KAO_OK;
}

struct SolveRun {
    kao_session *s = nullptr;
    int n = 0;
    bool has_target = false;
    std::vector<int64_t> target;
    std::vector<uint64_t> keys, prev;
    std::vector<double> t_best;
    std::vector<int64_t> dual_target;
    double t0 = 0, t_last_improve = 0;
    int launches = 0, dual_iters = 0, dual_now = 0;
    bool use_prices = true, all_done = false;
    // KAO-CX (kao_cycle.hip): cyclic-exchange improvement of incumbents K-search has stopped improving
    const kao_topic *topics = nullptr;
    std::vector<kao_topic> xt;                    // the session's topics: every caller topic `islands` times
    std::vector<int> origin;                      // session topic -> caller topic
    int n_user = 0;
    bool cx_on = true;
    bool cx_eager = false;                        // test hook KAO_CX_EAGER=1: KAO-CX after every launch, whatever the clock says
    double deadline = 0;
    std::vector<double> t_improved, t_cx;         // per topic: last improvement, last KAO-CX call (seconds from t0)
    std::vector<uint64_t> cx_seen;                // packed key KAO-CX last ran to a fixpoint on
    std::vector<uint16_t> cx_buf;
    std::vector<CycleCtx *> cx_ctx;
    int cx_calls = 0, cx_gains = 0;
    double cx_slice = 0.1;                        // seconds one KAO-CX call may take

    ~SolveRun() { for (CycleCtx *c : cx_ctx) cycle_close(c); if (s) kao_session_destroy(s); }

    // `so` is the caller's options with kao_solve's defaults applied; the session is created on the calling thread's device
    int begin(const kao_topic *user_topics, int n_topics, const kao_opts &so, const int64_t *tgt, double t_start, bool allow_islands = false) {
        t0 = t_start;
        n_user = n_topics;
        // Islands (kao_opts.islands > 1, off by default): every topic is searched as several independent copies (own seed,
        // own restarts, own elite, own K-bound trajectory and prices, own KAO-CX) sharing one copy's restart budget;
        // certificates are shared, the best copy answers.  Measured on the drifted 300 x 2000 topic (4 islands, 4 seeds,
        // 8 s): 14825 / 14825 / 14824 / 14822 against 14825 / 14826 (proven) / ... without -- no gain, so not the default.
        int k = 1;
        if (allow_islands && user_topics && n_topics >= 1 && so.restarts <= 0 && so.islands > 1) k = std::min(so.islands, 8);
        xt.clear(); origin.clear();
        for (int i = 0; i < n_topics; ++i)
            for (int c = 0; c < k; ++c) { xt.push_back(user_topics[i]); origin.push_back(i); }
        const kao_topic *topics = xt.data();
        kao_opts so_x = so;
        if (k > 1) {   // the islands share what one copy would have got: same work per launch, k basins
            int64_t slots = 1;
            for (int i = 0; i < n_topics; ++i) slots = std::max<int64_t>(slots, (int64_t)user_topics[i].n_partitions * std::max(user_topics[i].rf, 1));
            (void)require_init();
            const int cu = std::max(num_cu(cur_device()), 1);
            int r = std::min(std::max((cu * 32 / n_topics) / kWaves * kWaves, 8), 8192);
            r = std::min<int>(r, (int)std::max<int64_t>(cu, (((int64_t)1 << 22) / slots) / kWaves * kWaves));
            so_x.restarts = std::max((r / k) / kWaves * kWaves, 2 * kWaves);
        }
        n = n_topics = (int)xt.size();
        int rc = kao_session_create(topics, n_topics, &so_x, &s);
        if (rc) return rc;
        has_target = tgt != nullptr;
        if (tgt) { target.resize((size_t)n); for (int i = 0; i < n; ++i) target[(size_t)i] = tgt[origin[(size_t)i]]; }
        keys.assign((size_t)n, 0); prev.assign((size_t)n, ~0ull); t_best.assign((size_t)n, 0.0); dual_target.assign((size_t)n, -1);
        this->topics = topics;
        t_improved.assign((size_t)n, 0.0); t_cx.assign((size_t)n, 0.0); cx_seen.assign((size_t)n, ~0ull);
        cx_on = so.use_cycles >= 0;
        { const char *e = std::getenv("KAO_CX_EAGER"); cx_eager = e && e[0] == '1'; }
        cx_ctx.assign((size_t)n, nullptr);
        cx_slice = std::max(0.1, 0.1 * (so.time_limit_s > 0 ? so.time_limit_s : 10.0));
        deadline = t_start + (so.time_limit_s > 0 ? so.time_limit_s : 10.0);
        const kao_opts &o = s->opts;
        dual_iters = o.dual_iters < 0 ? 0 : (o.dual_iters == 0 ? 128 : o.dual_iters);
        dual_now = dual_iters;
        use_prices = so.use_prices >= 0;
        return KAO_OK;
    }
    int launch() { return kao_session_step(s); }   // asynchronous
    bool feasible(int i) const { return (keys[(size_t)i] >> 44) == 0; }
    int64_t objective(int i) const { return (int64_t)kObjCap - (int64_t)((keys[(size_t)i] >> 20) & 0xFFFFFF); }
    bool topic_done(int i) const {
        if (s->topic_infeasible[(size_t)i]) return true;  // proven infeasible: nothing to wait for
        const int64_t goal = has_target ? target[(size_t)i] : s->ub[(size_t)i];
        return feasible(i) && objective(i) >= goal;
    }
    bool check_done() const {   // every caller topic has one island that is done
        for (int i = 0; i < n;) {
            bool any = false;
            int j = i;
            for (; j < n && origin[(size_t)j] == origin[(size_t)i]; ++j) any = any || topic_done(j);
            if (!any) return false;
            i = j;
        }
        return true;
    }
    void share_bounds() {       // a certificate of any island holds for its caller topic
        for (int i = 0; i < n;) {
            int64_t ub = INT64_MAX;
            int j = i;
            for (; j < n && origin[(size_t)j] == origin[(size_t)i]; ++j) ub = std::min(ub, s->ub[(size_t)j]);
            for (int q = i; q < j; ++q) s->ub[(size_t)q] = ub;
            i = j;
        }
    }
    // waits for the launch, books improvements, merges a finished K-bound launch and starts the next one
    int after_launch() {
        int rc = kao_session_best_keys(s, keys.data());
        if (rc) return rc;
        ++launches;
        const double t = now_s() - t0;
        for (int i = 0; i < n; ++i)
            if (keys[(size_t)i] < prev[(size_t)i]) { prev[(size_t)i] = keys[(size_t)i]; t_best[(size_t)i] = t; t_last_improve = t; t_improved[(size_t)i] = t; }
        all_done = check_done();
        if ((rc = service_bound())) return rc;
        if (cx_on && !all_done && !has_target && (rc = cycles(t))) return rc;
        return KAO_OK;
    }
    // K-bound runs beside the search on its own stream; when a launch has finished its certificates are merged and, while
    // some feasible incumbent is still below its bound, the next launch starts (aimed at the new incumbents).  Launch length
    // adapts so that one launch takes about 10 ms.  Called after every K-search launch and between the rounds of KAO-CX.
    int service_bound() {
        if (has_target || dual_iters <= 0) return KAO_OK;
        int rc;
        const int busy = kao_session_bound_busy(s);
        if (busy < 0) return busy;
        if (busy) return KAO_OK;
        if (s->bound_inflight) {
            if ((rc = kao_session_bounds(s, nullptr, nullptr, nullptr))) return rc;
            share_bounds();
            // the finished launch's multipliers (rounded to quarters) become the prices of the next K-search launches
            if (use_prices && (rc = kao_session_adopt_prices(s))) return rc;
            if (s->bound_ms_last > 0) {
                const double scale = 10.0 / s->bound_ms_last;
                dual_now = (int)std::min(4096.0, std::max(32.0, s->bound_iters_last * std::min(4.0, std::max(0.25, scale))));
            }
            all_done = check_done();
        }
        bool any = false;
        for (int i = 0; i < n && !all_done; ++i) {
            const bool want = feasible(i) && objective(i) < s->ub[(size_t)i] && s->dual_ok[(size_t)i] && !s->topic_infeasible[(size_t)i] &&
                              !(s->dual_flags[(size_t)i] & 6);
            dual_target[(size_t)i] = want ? objective(i) : -1;
            any |= want;
        }
        if (any && (rc = kao_session_bound_step(s, dual_target.data(), dual_now))) return rc;
        return KAO_OK;
    }
    // between the rounds of KAO-CX K-bound is serviced.  (Keeping K-search running as well -- launches enqueued from here
    // whenever its stream had drained -- was measured and dropped: 300 x 2000, 8 seeds, 3 s: mean 14823.5 with, 14824.4 without.)
    static int poll_bound(void *self) { return static_cast<SolveRun *>(self)->service_bound(); }
    // KAO-CX for feasible, unproven topics whose search has stalled (no improvement for 50 ms) or that have not been looked at
    // for 250 ms: the incumbent goes through kao_cycle.hip to a fixpoint of the cyclic-exchange neighbourhood and, when that
    // improved it, comes back as the topic's incumbent (elite launches re-seed the restarts from it)
    int cycles(double t) {
        for (int i = 0; i < n; ++i) {
            if (s->topic_infeasible[(size_t)i] || !feasible(i) || objective(i) >= s->ub[(size_t)i]) continue;
            if (!cycle_supported(&topics[i])) continue;
            if ((keys[(size_t)i] >> 20) == (cx_seen[(size_t)i] >> 20)) continue;   // same incumbent as the last fixpoint
            const bool stalled = t - t_improved[(size_t)i] >= 0.05, due = t - t_cx[(size_t)i] >= 2.0 * cx_slice;
            if (!cx_eager && (t < 0.05 || !(stalled || due) || t - t_cx[(size_t)i] < 0.05)) continue;
            const size_t slots = (size_t)topics[i].n_partitions * topics[i].rf;
            cx_buf.resize(slots);
            int rc = session_topic_best(s, i, cx_buf.data());
            if (rc) return rc;
            int64_t obj = objective(i);
            int32_t st[8];
            if (!cx_ctx[(size_t)i] && !(cx_ctx[(size_t)i] = cycle_open(&topics[i], &rc))) return rc;
            const double slice_end = std::min(deadline, now_s() + cx_slice);
            rc = cycle_run(cx_ctx[(size_t)i], cx_buf.data(), 0, slice_end, &obj, st, &SolveRun::poll_bound, this);
            ++cx_calls;
            if (rc) return rc;
            const bool fixpoint = st[0] > st[1];   // the last round found nothing
            const double t2 = now_s() - t0;
            t_cx[(size_t)i] = t2;
            if (obj > objective(i)) {
                uint64_t key = 0;
                if ((rc = session_adopt_external(s, i, cx_buf.data(), obj, &key))) return rc;
                keys[(size_t)i] = prev[(size_t)i] = key;
                t_best[(size_t)i] = t_improved[(size_t)i] = t_last_improve = t2;
                ++cx_gains;
            }
            if (fixpoint) cx_seen[(size_t)i] = keys[(size_t)i];
            // a context holds ~90 B per broker pair on the device and as much on the host: keep a handful, not one per topic
            int open = 0;
            for (CycleCtx *c : cx_ctx) open += c != nullptr;
            if (open > 8) { cycle_close(cx_ctx[(size_t)i]); cx_ctx[(size_t)i] = nullptr; }
        }
        all_done = check_done();
        return KAO_OK;
    }
    int finish(kao_result *results, bool hit_time) {
        int rc = KAO_OK;
        if (s->bound_inflight && (rc = kao_session_bounds(s, nullptr, nullptr, nullptr))) return rc;  // last K-bound launch
        share_bounds();
        std::vector<kao_result> rs((size_t)n);
        std::vector<std::vector<uint16_t>> bufs((size_t)n);
        for (int i = 0; i < n; ++i) {
            rs[(size_t)i] = kao_result{};
            bufs[(size_t)i].assign((size_t)xt[(size_t)i].n_partitions * std::max(xt[(size_t)i].rf, 1), (uint16_t)KAO_NONE);
            rs[(size_t)i].assignment = bufs[(size_t)i].data();
        }
        if ((rc = kao_session_best(s, rs.data()))) return rc;
        auto better = [](const kao_result &a, const kao_result &b) {   // feasible first, then objective
            const bool fa = a.status != KAO_STATUS_NO_FEASIBLE && a.status != KAO_STATUS_INFEASIBLE_PROVEN;
            const bool fb = b.status != KAO_STATUS_NO_FEASIBLE && b.status != KAO_STATUS_INFEASIBLE_PROVEN;
            return fa != fb ? fa : a.objective > b.objective;
        };
        for (int i = 0; i < n;) {
            int best = i, j = i + 1;
            for (; j < n && origin[(size_t)j] == origin[(size_t)i]; ++j) if (better(rs[(size_t)j], rs[(size_t)best])) best = j;
            kao_result &out = results[origin[(size_t)i]];
            uint16_t *dst = out.assignment;
            out = rs[(size_t)best];
            out.assignment = dst;
            if (dst) std::memcpy(dst, bufs[(size_t)best].data(), bufs[(size_t)best].size() * 2);
            out.seconds_to_best = t_best[(size_t)best];
            if (hit_time && out.status == KAO_STATUS_FEASIBLE_BOUND_GAP) out.status = KAO_STATUS_TIME_LIMIT;
            i = j;
        }
        return rc;
    }
};

// kao_solve's defaults on top of the caller's options
kao_opts solve_defaults(const kao_topic *topics, int32_t n_topics, const kao_opts *opts) {
    kao_opts so{};
    if (opts) so = *opts;
    if (so.iters_per_launch <= 0) {
        // latency first (the host checks the bound after every launch): 128 iterations; large topics pay O(P) per launch
        // for loading, recounting and storing a restart (drifted 500 x 5000 topic: 40 % more iterations per second with
        // 512 per launch), so they get longer launches
        int64_t slots = 0;
        for (int i = 0; topics && i < n_topics; ++i) slots = std::max<int64_t>(slots, (int64_t)topics[i].n_partitions * std::max(topics[i].rf, 1));
        so.iters_per_launch = slots <= 4096 ? 128 : (slots <= 8192 ? 256 : 512);
    }
    if (so.elite_period == 0 && topics && n_topics > 0) {  // about one penalty period of the largest topic between elite launches
        int lg = 8;
        for (int i = 0; i < n_topics; ++i)
            lg = std::max(lg, so.period_log2 > 0 ? so.period_log2 : auto_period_log2(topics[i].n_partitions, std::max(topics[i].rf, 1)));
        so.elite_period = std::max(1, (1 << std::min(lg, 20)) / so.iters_per_launch);
    }
    return so;
}

}  // namespace

namespace kao {
int api_fail(int code, const char *msg) { return fail(code, msg ? msg : ""); }
int api_require_init() { return require_init(); }
double api_now_s() { return now_s(); }
}  // namespace kao

extern "C" {

int kao_solve(const kao_topic *topics, int32_t n_topics, const kao_opts *opts, kao_result *results) {
    const double t0 = now_s();
    if (!results) return fail(KAO_ERR_INVALID, "null results");
    const kao_opts so = solve_defaults(topics, n_topics, opts);
    SolveRun run;
    int rc = run.begin(topics, n_topics, so, opts ? opts->target_objective : nullptr, t0, true);
    if (rc) return rc;
    g_timing[0] = now_s() - t0;
    const kao_opts &o = run.s->opts;
    bool hit_time = false;
    for (;;) {
        if ((rc = run.launch()) || (rc = run.after_launch())) return rc;
        if (o.stop_at_bound && run.all_done) break;
        if (o.max_launches > 0 && run.launches >= o.max_launches) break;
        if (now_s() - t0 >= o.time_limit_s) { hit_time = true; break; }
    }
    g_timing[1] = run.t_last_improve;
    rc = run.finish(results, hit_time);
    g_timing[2] = now_s() - t0;
    g_timing[5] = (double)run.s->delta_total;
    g_timing[6] = (double)run.s->bound_launches;
    kao_session_destroy(run.s);
    run.s = nullptr;
    g_timing[3] = now_s() - t0;
    g_timing[4] = run.launches;
    return rc;
}

// ------------------------------------------------------------------------------------------------
// kao_solve_multi: one process, several GPUs
// ------------------------------------------------------------------------------------------------
namespace {
// librccl.so is half a gigabyte of code objects; linking it would make every process that loads libkao.so (the CLI, a JVM)
// pay for registering them.  It is opened on the first multi-GPU exchange instead.
struct Rccl {
    void *h = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    bool load() {
        if (h) return true;
        // RCCL must sit on the SAME HIP / HSA runtime instance this library runs on (a process may hold two: PyTorch wheels
        // bundle their own next to /opt/rocm's): look for librccl next to the libamdhip64 that serves our HIP calls first
        std::vector<std::string> names;
        Dl_info info;
        if (dladdr(reinterpret_cast<void *>(&hipGetDeviceCount), &info) && info.dli_fname) {
            std::string dir(info.dli_fname);
            const size_t slash = dir.rfind('/');
            if (slash != std::string::npos) { dir.resize(slash); names.push_back(dir + "/librccl.so.1"); names.push_back(dir + "/librccl.so"); }
        }
        names.push_back("librccl.so.1"); names.push_back("librccl.so"); names.push_back("/opt/rocm/lib/librccl.so.1");
        for (const std::string &name : names) {
            h = dlopen(name.c_str(), RTLD_NOW | RTLD_LOCAL);
            if (h) break;
        }
        if (!h) return false;
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(dlsym(h, "ncclCommInitAll"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(h, "ncclCommDestroy"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(h, "ncclGetErrorString"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(dlsym(h, "ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(dlsym(h, "ncclGroupEnd"));
        AllReduce = reinterpret_cast<decltype(AllReduce)>(dlsym(h, "ncclAllReduce"));
        Broadcast = reinterpret_cast<decltype(Broadcast)>(dlsym(h, "ncclBroadcast"));
        return CommInitAll && CommDestroy && GetErrorString && GroupStart && GroupEnd && AllReduce && Broadcast;
    }
} g_rccl;
struct CommSet { std::vector<int> devices; std::vector<ncclComm_t> comms; };
std::vector<CommSet> g_comms;   // RCCL communicators per device list (creation costs ~100 ms; kept until kao_shutdown)
std::mutex g_comm_mu;
int comms_for(const std::vector<int> &devices, std::vector<ncclComm_t> &out) {
    std::lock_guard<std::mutex> lock(g_comm_mu);
    for (const CommSet &c : g_comms) if (c.devices == devices) { out = c.comms; return KAO_OK; }
    if (!g_rccl.load()) return fail(KAO_ERR_HIP, std::string("librccl.so not available: ") + (dlerror() ? dlerror() : "missing symbol"));
    for (int d : devices) {  // RCCL expects every device's primary context to exist already
        if (hipSetDevice(d) != hipSuccess || hipFree(nullptr) != hipSuccess) return fail(KAO_ERR_HIP, "cannot initialise device " + std::to_string(d));
        void *probe = nullptr;
        if (hipMalloc(&probe, 256) == hipSuccess) (void)hipFree(probe);
    }
    if (cur_device() >= 0) (void)hipSetDevice(cur_device());
    CommSet c; c.devices = devices; c.comms.resize(devices.size());
    const ncclResult_t r = g_rccl.CommInitAll(c.comms.data(), (int)devices.size(), devices.data());
    if (r != ncclSuccess) return fail(KAO_ERR_HIP, std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(r));
    g_comms.push_back(c);
    out = c.comms;
    return KAO_OK;
}
}  // namespace

void kao_multi_shutdown_comms(void) {
    std::lock_guard<std::mutex> lock(g_comm_mu);
    for (CommSet &c : g_comms) for (ncclComm_t cm : c.comms) (void)g_rccl.CommDestroy(cm);
    g_comms.clear();
}

// Diagnostic: the collectives kao_solve_multi uses, on small resident buffers of the listed (distinct) devices -- rank r holds
// keys {100 - r, 7 + r, ~0, r}; after ncclAllReduce(ncclUint64, ncclMin) every rank must hold {101 - n, 7, ~0, 0}, and after
// ncclBroadcast from the last rank every rank holds that rank's 64-byte pattern.  0 = ok.
int kao_rccl_selftest(const int32_t *devices, int32_t n_dev) {
    if (!devices || n_dev < 1 || n_dev > kMaxDevices) return fail(KAO_ERR_INVALID, "bad device list");
    std::vector<int> devs(devices, devices + n_dev);
    std::vector<ncclComm_t> comms;
    int rc = comms_for(devs, comms);
    if (rc) return rc;
    std::vector<unsigned long long *> keys((size_t)n_dev, nullptr), out((size_t)n_dev, nullptr);
    std::vector<unsigned char *> pat((size_t)n_dev, nullptr);
    std::vector<hipStream_t> st((size_t)n_dev, nullptr);
    bool ok = true;
    for (int d = 0; d < n_dev && ok; ++d) {
        ok = hipSetDevice(devs[(size_t)d]) == hipSuccess && hipMalloc(reinterpret_cast<void **>(&keys[(size_t)d]), 32) == hipSuccess &&
             hipMalloc(reinterpret_cast<void **>(&out[(size_t)d]), 32) == hipSuccess && hipMalloc(reinterpret_cast<void **>(&pat[(size_t)d]), 64) == hipSuccess &&
             hipStreamCreateWithFlags(&st[(size_t)d], hipStreamNonBlocking) == hipSuccess;
        const unsigned long long h[4] = {100ull - (unsigned)d, 7ull + (unsigned)d, ~0ull, (unsigned long long)d};
        unsigned char p[64];
        for (int i = 0; i < 64; ++i) p[i] = (unsigned char)(d * 64 + i);
        ok = ok && hipMemcpy(keys[(size_t)d], h, 32, hipMemcpyHostToDevice) == hipSuccess && hipMemcpy(pat[(size_t)d], p, 64, hipMemcpyHostToDevice) == hipSuccess;
    }
    ncclResult_t nr = ok ? g_rccl.GroupStart() : ncclSystemError;
    for (int d = 0; d < n_dev && nr == ncclSuccess; ++d) nr = g_rccl.AllReduce(keys[(size_t)d], out[(size_t)d], 4, ncclUint64, ncclMin, comms[(size_t)d], st[(size_t)d]);
    if (nr == ncclSuccess) nr = g_rccl.GroupEnd();
    if (nr == ncclSuccess) nr = g_rccl.GroupStart();
    for (int d = 0; d < n_dev && nr == ncclSuccess; ++d) nr = g_rccl.Broadcast(pat[(size_t)d], pat[(size_t)d], 64, ncclUint8, n_dev - 1, comms[(size_t)d], st[(size_t)d]);
    if (nr == ncclSuccess) nr = g_rccl.GroupEnd();
    ok = ok && nr == ncclSuccess;
    for (int d = 0; d < n_dev && ok; ++d) {
        unsigned long long h[4]; unsigned char p[64];
        ok = hipSetDevice(devs[(size_t)d]) == hipSuccess && hipStreamSynchronize(st[(size_t)d]) == hipSuccess &&
             hipMemcpy(h, out[(size_t)d], 32, hipMemcpyDeviceToHost) == hipSuccess && hipMemcpy(p, pat[(size_t)d], 64, hipMemcpyDeviceToHost) == hipSuccess;
        ok = ok && h[0] == 101ull - (unsigned)n_dev && h[1] == 7ull && h[2] == ~0ull && h[3] == 0ull;
        for (int i = 0; i < 64 && ok; ++i) ok = p[i] == (unsigned char)((n_dev - 1) * 64 + i);
    }
    for (int d = 0; d < n_dev; ++d) {
        (void)hipSetDevice(devs[(size_t)d]);
        (void)hipFree(keys[(size_t)d]); (void)hipFree(out[(size_t)d]); (void)hipFree(pat[(size_t)d]);
        if (st[(size_t)d]) (void)hipStreamDestroy(st[(size_t)d]);
    }
    if (cur_device() >= 0) (void)hipSetDevice(cur_device());
    if (!ok) return fail(KAO_ERR_HIP, nr != ncclSuccess ? std::string("RCCL: ") + g_rccl.GetErrorString(nr) : std::string("RCCL self-test: wrong result"));
    return KAO_OK;
}

int kao_solve_multi(const kao_topic *topics, int32_t n_topics, const int32_t *devices, int32_t n_dev, const kao_opts *opts,
                    kao_result *results) {
    const double t0 = now_s();
    if (!topics || n_topics < 1 || !results) return fail(KAO_ERR_INVALID, "no topics / null results");
    if (!devices || n_dev < 1 || n_dev > kMaxDevices) return fail(KAO_ERR_INVALID, "bad device list");
    int n_hw = 0;
    if (hipGetDeviceCount(&n_hw) != hipSuccess || n_hw <= 0) return fail(KAO_ERR_NO_DEVICE, "no HIP device");
    std::vector<int> devs(devices, devices + n_dev);
    bool distinct = true;
    for (int i = 0; i < n_dev; ++i) {
        if (devs[(size_t)i] < 0 || devs[(size_t)i] >= n_hw) return fail(KAO_ERR_INVALID, "device ordinal out of range");
        for (int j = 0; j < i; ++j) distinct &= devs[(size_t)j] != devs[(size_t)i];
    }
    if (!g_init) { int rc0 = kao_init(devs[0]); if (rc0) return rc0; }
    const kao_opts so = solve_defaults(topics, n_topics, opts);
    const bool replicated = n_topics < n_dev;   // fewer topics than GPUs: every GPU searches every topic, elites are exchanged
    // ---- shards: LPT by brokers x partitions (independent sub-problems, README.md:146-184) ----
    std::vector<std::vector<int>> shard((size_t)n_dev);
    if (replicated) for (auto &sh : shard) for (int i = 0; i < n_topics; ++i) sh.push_back(i);
    else {
        std::vector<int> order((size_t)n_topics);
        for (int i = 0; i < n_topics; ++i) order[(size_t)i] = i;
        auto size_of = [&](int i) { return (int64_t)topics[i].n_brokers * topics[i].n_partitions; };
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return size_of(a) > size_of(b); });
        std::vector<int64_t> load((size_t)n_dev, 0);
        for (int i : order) {
            int best = 0;
            for (int d = 1; d < n_dev; ++d) if (load[(size_t)d] < load[(size_t)best]) best = d;
            shard[(size_t)best].push_back(i);
            load[(size_t)best] += size_of(i);
        }
    }
    std::vector<ncclComm_t> comms;
    const bool use_rccl = replicated && distinct;   // logical shards on one device (tests) merge through the host instead
    if (use_rccl) { int rc0 = comms_for(devs, comms); if (rc0) return rc0; }

    struct Dev { SolveRun run; std::vector<kao_topic> tp; std::vector<kao_result> res; std::vector<int64_t> tgt; };
    std::vector<Dev> D((size_t)n_dev);
    const int saved_t_device = t_device;
    auto cleanup = [&]() { t_device = saved_t_device; if (cur_device() >= 0) (void)hipSetDevice(cur_device()); };
    int rc = KAO_OK;
    for (int d = 0; d < n_dev && !rc; ++d) {
        Dev &x = D[(size_t)d];
        for (int i : shard[(size_t)d]) { x.tp.push_back(topics[i]); if (opts && opts->target_objective) x.tgt.push_back(opts->target_objective[i]); }
        if (x.tp.empty()) continue;
        kao_opts o = so;
        o.seed = so.seed + 0x9E3779B97F4A7C15ull * (uint64_t)d;            // replicated topics: a different seed per GPU
        if (replicated && d > 0) o.dual_iters = -1;                        // one certificate per topic is enough: device 0 runs K-bound
        t_device = devs[(size_t)d];
        if (hipSetDevice(t_device) != hipSuccess) { rc = fail(KAO_ERR_NO_DEVICE, "hipSetDevice"); break; }
        rc = x.run.begin(x.tp.data(), (int)x.tp.size(), o, x.tgt.empty() ? nullptr : x.tgt.data(), t0);
    }
    if (rc) { cleanup(); return rc; }
    g_timing[0] = now_s() - t0;
    const int exch = std::max(1, so.elite_period);
    bool hit_time = false;
    int rounds = 0;
    uint64_t exchanges = 0;
    std::vector<uint64_t> gmin((size_t)n_topics);
    std::vector<int> root((size_t)n_topics);
    for (;;) {
        for (int d = 0; d < n_dev && !rc; ++d) if (D[(size_t)d].run.s) { t_device = devs[(size_t)d]; rc = D[(size_t)d].run.launch(); }
        for (int d = 0; d < n_dev && !rc; ++d) if (D[(size_t)d].run.s) { t_device = devs[(size_t)d]; rc = D[(size_t)d].run.after_launch(); }
        if (rc) break;
        ++rounds;
        if (replicated) {
            // certificates: any GPU's bound is valid for the topic
            for (int i = 0; i < n_topics; ++i) {
                int64_t ub = INT64_MAX;
                for (Dev &x : D) ub = std::min(ub, x.run.s->ub[(size_t)i]);
                for (Dev &x : D) x.run.s->ub[(size_t)i] = ub;
            }
            if (rounds % exch == 0 || so.stop_at_bound) {
                // ---- elite exchange: min-allreduce of the packed best keys on the resident buffers, winners broadcast ----
                bool differ = false;
                for (int i = 0; i < n_topics; ++i) {
                    gmin[(size_t)i] = ~0ull; root[(size_t)i] = 0;
                    for (int d = 0; d < n_dev; ++d) {
                        const uint64_t k = D[(size_t)d].run.keys[(size_t)i] | 0;  // host copy read by after_launch
                        if (k < gmin[(size_t)i]) { gmin[(size_t)i] = k; root[(size_t)i] = d; }
                    }
                    for (int d = 0; d < n_dev; ++d) differ |= (D[(size_t)d].run.keys[(size_t)i] >> 20) != (gmin[(size_t)i] >> 20);
                }
                if (differ && rounds % exch == 0) {
                    ++exchanges;
                    for (int d = 0; d < n_dev && !rc; ++d) {   // stage every GPU's current winners
                        kao_session *s = D[(size_t)d].run.s;
                        if (hipSetDevice(s->device) != hipSuccess) { rc = fail(KAO_ERR_HIP, "hipSetDevice"); break; }
                        launch_gather(s->d_topics, s->n_topics, s->d_keys, s->d_best, s->d_viol, s->d_win_assign, s->d_win_viol, s->stream);
                    }
                    if (!rc && use_rccl) {
                        ncclResult_t nr = g_rccl.GroupStart();
                        for (int d = 0; d < n_dev && nr == ncclSuccess; ++d) {
                            kao_session *s = D[(size_t)d].run.s;
                            nr = g_rccl.AllReduce(s->d_keys, s->d_keys_glob, (size_t)n_topics, ncclUint64, ncclMin, comms[(size_t)d], s->stream);
                        }
                        if (nr == ncclSuccess) nr = g_rccl.GroupEnd();
                        for (int i = 0; i < n_topics && nr == ncclSuccess; ++i) {
                            if ((gmin[(size_t)i] >> 44) != 0) continue;   // no feasible assignment anywhere yet
                            nr = g_rccl.GroupStart();
                            for (int d = 0; d < n_dev && nr == ncclSuccess; ++d) {
                                kao_session *s = D[(size_t)d].run.s;
                                const TopicDev &td = s->pts[(size_t)i].d;
                                uint16_t *buf = s->d_win_assign + td.win_off;
                                nr = g_rccl.Broadcast(buf, buf, (size_t)td.P * td.RF * 2, ncclUint8, root[(size_t)i], comms[(size_t)d], s->stream);
                            }
                            if (nr == ncclSuccess) nr = g_rccl.GroupEnd();
                        }
                        if (nr != ncclSuccess) rc = fail(KAO_ERR_HIP, std::string("RCCL elite exchange: ") + g_rccl.GetErrorString(nr));
                    } else if (!rc) {   // logical shards on one device: same data movement through plain copies
                        for (int d = 0; d < n_dev && !rc; ++d) {
                            kao_session *s = D[(size_t)d].run.s;
                            if (hipSetDevice(s->device) != hipSuccess) { rc = fail(KAO_ERR_HIP, "hipSetDevice"); break; }
                            for (int r2 = 0; r2 < n_dev; ++r2) (void)hipStreamSynchronize(D[(size_t)r2].run.s->stream);
                            if (hipMemcpyAsync(s->d_keys_glob, gmin.data(), (size_t)n_topics * 8, hipMemcpyHostToDevice, s->stream) != hipSuccess) rc = fail(KAO_ERR_HIP, "hipMemcpyAsync");
                            for (int i = 0; i < n_topics && !rc; ++i) {
                                if ((gmin[(size_t)i] >> 44) != 0 || root[(size_t)i] == d) continue;
                                kao_session *sr = D[(size_t)root[(size_t)i]].run.s;
                                const TopicDev &td = s->pts[(size_t)i].d;
                                if (hipMemcpyAsync(s->d_win_assign + td.win_off, sr->d_win_assign + td.win_off, (size_t)td.P * td.RF * 2,
                                                   hipMemcpyDeviceToDevice, s->stream) != hipSuccess) rc = fail(KAO_ERR_HIP, "hipMemcpyAsync");
                            }
                            if (!rc && hipStreamSynchronize(s->stream) != hipSuccess) rc = fail(KAO_ERR_HIP, "hipStreamSynchronize");  // gmin is reused
                        }
                    }
                    for (int d = 0; d < n_dev && !rc; ++d) {
                        kao_session *s = D[(size_t)d].run.s;
                        if (hipSetDevice(s->device) != hipSuccess) { rc = fail(KAO_ERR_HIP, "hipSetDevice"); break; }
                        launch_adopt_global(s->d_keys, s->d_keys_glob, n_topics, s->stream);
                        if (hipGetLastError() != hipSuccess) rc = fail(KAO_ERR_HIP, "k_adopt_global");
                    }
                    if (rc) break;
                }
                for (Dev &x : D)   // every run now judges "done" against the global incumbents
                    for (int i = 0; i < n_topics; ++i) if (gmin[(size_t)i] < x.run.keys[(size_t)i]) x.run.keys[(size_t)i] = gmin[(size_t)i];
                for (Dev &x : D) x.run.all_done = x.run.check_done();
            }
        }
        bool all = true;
        for (Dev &x : D) if (x.run.s) all &= x.run.all_done;
        if (replicated) { all = false; for (Dev &x : D) all |= x.run.all_done; }   // one GPU holding proven optima for every topic suffices
        if (so.stop_at_bound && all) break;
        if (so.max_launches > 0 && rounds >= so.max_launches) break;
        const double tl = so.time_limit_s > 0 ? so.time_limit_s : 10.0;
        if (now_s() - t0 >= tl) { hit_time = true; break; }
    }
    // ---- results ----
    double t_improve = 0, cand = 0, bl = 0;
    if (!rc) {
        if (replicated) {   // per topic: the GPU holding the best key answers
            std::vector<std::vector<kao_result>> rs((size_t)n_dev, std::vector<kao_result>((size_t)n_topics));
            std::vector<std::vector<std::vector<uint16_t>>> bufs((size_t)n_dev);
            for (int d = 0; d < n_dev && !rc; ++d) {
                bufs[(size_t)d].resize((size_t)n_topics);
                for (int i = 0; i < n_topics; ++i) {
                    bufs[(size_t)d][(size_t)i].assign((size_t)topics[i].n_partitions * topics[i].rf, (uint16_t)KAO_NONE);
                    rs[(size_t)d][(size_t)i] = kao_result{};
                    rs[(size_t)d][(size_t)i].assignment = bufs[(size_t)d][(size_t)i].data();
                }
                t_device = devs[(size_t)d];
                rc = D[(size_t)d].run.finish(rs[(size_t)d].data(), hit_time);
            }
            for (int i = 0; i < n_topics && !rc; ++i) {
                int best = 0;
                auto better = [&](const kao_result &a, const kao_result &b) {   // feasible first, then objective
                    const bool fa = a.status != KAO_STATUS_NO_FEASIBLE && a.status != KAO_STATUS_INFEASIBLE_PROVEN;
                    const bool fb = b.status != KAO_STATUS_NO_FEASIBLE && b.status != KAO_STATUS_INFEASIBLE_PROVEN;
                    return fa != fb ? fa : a.objective > b.objective;
                };
                for (int d = 1; d < n_dev; ++d) if (better(rs[(size_t)d][(size_t)i], rs[(size_t)best][(size_t)i])) best = d;
                uint16_t *dst = results[i].assignment;
                results[i] = rs[(size_t)best][(size_t)i];
                results[i].assignment = dst;
                if (dst) std::memcpy(dst, bufs[(size_t)best][(size_t)i].data(), bufs[(size_t)best][(size_t)i].size() * 2);
                int64_t ub = INT64_MAX;
                for (int d = 0; d < n_dev; ++d) ub = std::min(ub, rs[(size_t)d][(size_t)i].upper_bound);
                results[i].upper_bound = ub;
                if (results[i].status == KAO_STATUS_FEASIBLE_BOUND_GAP || results[i].status == KAO_STATUS_TIME_LIMIT || results[i].status == KAO_STATUS_OPTIMAL_PROVEN)
                    results[i].status = results[i].objective >= ub ? KAO_STATUS_OPTIMAL_PROVEN : (hit_time ? KAO_STATUS_TIME_LIMIT : KAO_STATUS_FEASIBLE_BOUND_GAP);
            }
        } else {
            for (int d = 0; d < n_dev && !rc; ++d) {
                Dev &x = D[(size_t)d];
                if (!x.run.s) continue;
                x.res.assign(x.tp.size(), kao_result{});
                for (size_t k = 0; k < x.tp.size(); ++k) x.res[k].assignment = results[shard[(size_t)d][k]].assignment;
                t_device = devs[(size_t)d];
                rc = x.run.finish(x.res.data(), hit_time);
                for (size_t k = 0; k < x.tp.size() && !rc; ++k) results[shard[(size_t)d][k]] = x.res[k];
            }
        }
    }
    g_timing[2] = now_s() - t0;
    for (Dev &x : D) if (x.run.s) { t_improve = std::max(t_improve, x.run.t_last_improve); cand += (double)x.run.s->delta_total; bl += (double)x.run.s->bound_launches; }
    for (int d = 0; d < n_dev; ++d) if (D[(size_t)d].run.s) { t_device = devs[(size_t)d]; kao_session_destroy(D[(size_t)d].run.s); D[(size_t)d].run.s = nullptr; }
    cleanup();
    g_timing[1] = t_improve; g_timing[3] = now_s() - t0; g_timing[4] = rounds; g_timing[5] = cand; g_timing[6] = bl; g_timing[7] = (double)exchanges;
    return rc;
}

// ------------------------------------------------------------------------------------------------
// kao_solve_capped: cluster-wide per-broker load caps, priced (Lagrangian) over independent per-topic solves
// ------------------------------------------------------------------------------------------------
int kao_solve_capped(const kao_topic *topics, int32_t n_topics, const int32_t *replica_cap, const int32_t *devices, int32_t n_dev,
                     const kao_opts *opts, int32_t max_rounds, kao_result *results, int64_t *lagrangian_bound) {
    const double t0 = now_s();
    if (!topics || n_topics < 1 || !replica_cap || !results) return fail(KAO_ERR_INVALID, "null argument");
    const int B = topics[0].n_brokers;
    int wmax = 1;
    for (int i = 0; i < n_topics; ++i) {
        if (topics[i].n_brokers != B) return fail(KAO_ERR_INVALID, "kao_solve_capped: every topic must use the same broker set");
        if (topics[i].broker_w || topics[i].broker_wl) return fail(KAO_ERR_UNSUPPORTED, "kao_solve_capped: topics with their own broker weights");
        for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) wmax = std::max(wmax, topics[i].w[a][b]);
    }
    const int mu_max = std::min(1023, 4 * wmax);   // a price above every objective weight already repels every replica
    kao_opts o{};
    if (opts) o = *opts;
    const double limit = o.time_limit_s > 0 ? o.time_limit_s : 10.0;
    const int rounds = max_rounds > 0 ? max_rounds : 40;
    if (o.max_launches <= 0) o.max_launches = 6;
    o.stop_at_bound = 1;
    o.target_objective = nullptr;
    std::vector<int32_t> mu((size_t)B, 0), bw((size_t)B, 0);
    std::vector<kao_topic> tp(topics, topics + n_topics);
    std::vector<std::vector<uint16_t>> buf((size_t)n_topics), inc((size_t)n_topics);
    std::vector<kao_result> res((size_t)n_topics), inc_res((size_t)n_topics);
    for (int i = 0; i < n_topics; ++i) buf[(size_t)i].assign((size_t)topics[i].n_partitions * topics[i].rf, (uint16_t)KAO_NONE);
    int64_t inc_total = -1, best_L = INT64_MAX, n_slots = 0;
    for (int i = 0; i < n_topics; ++i) n_slots += (int64_t)topics[i].n_partitions * topics[i].rf;
    std::vector<int64_t> load((size_t)B);
    int rc = KAO_OK, r = 0;
    for (; r < rounds; ++r) {
        const double left = limit - (now_s() - t0);
        if (r > 0 && left <= 0) break;
        int M = 0;
        for (int b = 0; b < B; ++b) M = std::max(M, mu[(size_t)b]);
        for (int b = 0; b < B; ++b) bw[(size_t)b] = M - mu[(size_t)b];   // weights must be >= 0: a constant M per replica does not change any argmax
        for (int i = 0; i < n_topics; ++i) {
            tp[(size_t)i].broker_w = M ? bw.data() : nullptr;
            res[(size_t)i] = kao_result{};
            res[(size_t)i].assignment = buf[(size_t)i].data();
        }
        o.time_limit_s = std::max(0.05, left / std::max(1, std::min(rounds - r, 8)));
        o.seed = (opts ? opts->seed : 0) + (uint64_t)r * 0x9E3779B97F4A7C15ull;
        rc = (devices && n_dev > 1) ? kao_solve_multi(tp.data(), n_topics, devices, n_dev, &o, res.data())
                                    : kao_solve(tp.data(), n_topics, &o, res.data());
        if (rc) break;
        // ---- broker loads over ALL topics (the allreduce(SUM) of a sharded deployment), objective without the weights ----
        std::fill(load.begin(), load.end(), 0);
        bool all_feasible = true, all_proven = true;
        int64_t total = 0, total_w = 0;
        for (int i = 0; i < n_topics; ++i) {
            const kao_result &x = res[(size_t)i];
            if (x.status == KAO_STATUS_NO_FEASIBLE || x.status == KAO_STATUS_INFEASIBLE_PROVEN) { all_feasible = false; continue; }
            all_proven &= x.status == KAO_STATUS_OPTIMAL_PROVEN;
            int64_t wsum = 0;
            for (size_t k = 0; k < buf[(size_t)i].size(); ++k) { const uint16_t b = buf[(size_t)i][k]; load[b]++; wsum += M ? bw[b] : 0; }
            total += x.objective - wsum;
            total_w += x.upper_bound;
        }
        int64_t worst = 0, priced = 0;
        for (int b = 0; b < B; ++b) {
            if (replica_cap[b] < 0) continue;
            worst = std::max<int64_t>(worst, load[(size_t)b] - replica_cap[b]);
            priced += (int64_t)mu[(size_t)b] * replica_cap[b];
        }
        if (all_feasible && all_proven) best_L = std::min(best_L, total_w - (int64_t)M * n_slots + priced);   // L(mu) >= capped optimum
        if (all_feasible && worst <= 0 && total > inc_total) {   // respects every cap: a candidate answer
            inc_total = total;
            for (int i = 0; i < n_topics; ++i) {
                inc[(size_t)i] = buf[(size_t)i];
                inc_res[(size_t)i] = res[(size_t)i];
                int64_t wsum = 0;
                for (uint16_t b : buf[(size_t)i]) wsum += M ? bw[b] : 0;
                inc_res[(size_t)i].objective = res[(size_t)i].objective - wsum;
            }
        }
        if (inc_total >= 0 && best_L != INT64_MAX && inc_total >= best_L) break;   // incumbent meets the Lagrangian bound: optimal
        // ---- projected subgradient step on the prices, diminishing: alpha = 1 / (1 + r / 3) ----
        const int den = 1 + r / 3;
        bool moved = false;
        for (int b = 0; b < B; ++b) {
            if (replica_cap[b] < 0) continue;
            const int64_t ex = load[(size_t)b] - replica_cap[b];
            int d = 0;
            if (ex > 0) d = (int)std::max<int64_t>(1, ex / den);
            else if (ex < 0 && mu[(size_t)b] > 0) d = -(int)std::min<int64_t>(mu[(size_t)b], std::max<int64_t>(r >= 6 ? 0 : 1, (-ex) / (2 * den)));
            const int nm = std::min(mu_max, std::max(0, mu[(size_t)b] + d));
            moved |= nm != mu[(size_t)b];
            mu[(size_t)b] = nm;
        }
        if (!moved) break;   // prices are stationary
    }
    if (!rc) {
        for (int i = 0; i < n_topics; ++i) {
            uint16_t *dst = results[i].assignment;
            if (inc_total >= 0) {
                results[i] = inc_res[(size_t)i];
                results[i].status = (best_L != INT64_MAX && inc_total >= best_L) ? KAO_STATUS_OPTIMAL_PROVEN : KAO_STATUS_FEASIBLE_BOUND_GAP;
                results[i].upper_bound = best_L != INT64_MAX ? best_L : INT64_MAX;   // a bound on the SUM over all topics
                if (dst) std::memcpy(dst, inc[(size_t)i].data(), inc[(size_t)i].size() * 2);
            } else {
                results[i] = res[(size_t)i];
                results[i].status = KAO_STATUS_NO_FEASIBLE; results[i].objective = -1;
            }
            results[i].assignment = dst;
        }
        if (lagrangian_bound) *lagrangian_bound = best_L;
    }
    g_timing[3] = now_s() - t0; g_timing[4] = r;
    return rc;
}

int kao_last_solve_timing(double out[8]) {
    if (!out) return fail(KAO_ERR_INVALID, "null out");
    for (int i = 0; i < 8; ++i) out[i] = g_timing[i];
    return KAO_OK;
}

}  // extern "C"
