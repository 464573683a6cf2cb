// kao_pairs.cpp -- host side of libkao.so, part 4 (see kao_host.h): COMPOUND EDGES of leader-balanced pairs, the combinatorial
// core of the next KAO-CX layer (specification and measurements: oracle/kao_cycle_pairs.py).
//
// What KAO-CX's closures still miss on rigid topics is a cycle of replica units that runs THROUGH the leader slots of leader
// transfers whose leader units balance only pairwise (local branching around its fixpoints: five partitions on the drifted
// 300 x 2000 topic).  Here every way a partition can hand its leadership to another broker is enumerated as a HALF-MOVE
//   a  role swap with a follower                      b  a follower is promoted, the leader leaves, y enters as follower
//   c  v (outside the row) replaces the leader        d  v enters as leader, the old leader stays as follower, a follower leaves
//   e  v replaces the leader AND a follower is replaced by y
// (y = a current replica of the partition outside the row, or a GENERIC broker of weight 0), kept when its objective gain is
// >= gmin; half-moves u -> v of partitions led by u are joined with half-moves v -> u of partitions led by v (leader counts
// unchanged); concrete brokers that leave one row and enter the other cancel, a generic entering follower may take a broker the
// partner releases, and a pair whose net replica effect is ONE unit x -> z is a compound edge of the F graph with cost
// -(gain of both rows) (a generic follower left over: an edge to every z that keeps C7, README.md:178-180, in its row).
// Pure host code, no GPU: the closure of the augmented graph, the unrolling and the exact evaluation (k_cx_square, k_cx_patch,
// K-eval) are not wired in yet -- this entry point is the parity hook of the enumeration (tests/test_host.py against the oracle).
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unordered_map>

#include "kao_host.h"

namespace {

constexpr int kAny = -1;

struct Half {
    int p, gain;
    int row[KAO_MAX_RF];
    int outs[2], ins[2], no, ni;
};

struct PairCtx {
    const kao_topic *t;
    const uint16_t *A;
    int B, R, P, RF, rfc, plo, phi;
    // objective weight of broker b as leader / follower of partition p (README.md:145-146; the last matching current slot wins,
    // as in kao_cycle.hip::cx_wt), plus the topic's broker weights
    int wt(int p, int b, int role) const {
        int w = 0;
        const uint16_t *c = t->current + (size_t)p * rfc;
        for (int k = 0; k < rfc; ++k)
            if ((int)c[k] == b) w = t->w[k == 0 ? 0 : 1][role];
        if (t->broker_w) w += t->broker_w[b];
        if (t->broker_wl && role == 0) w += t->broker_wl[b];
        return w;
    }
    bool c7_ok(const int *row) const {   // distinct brokers, every rack inside the per-partition band
        int cnt[256] = {0};
        for (int i = 0; i < RF; ++i) {
            if (row[i] < 0 || row[i] >= B) return false;
            for (int j = 0; j < i; ++j) if (row[j] == row[i]) return false;
            ++cnt[t->rack_of[row[i]]];
        }
        for (int r = 0; r < R; ++r) if (cnt[r] > phi || cnt[r] < plo) return false;
        return true;
    }
};

bool contains(const int *a, int n, int x) { for (int i = 0; i < n; ++i) if (a[i] == x) return true; return false; }
void remove_first(int *a, int &n, int x) {
    for (int i = 0; i < n; ++i)
        if (a[i] == x) { for (int j = i + 1; j < n; ++j) a[j - 1] = a[j]; --n; return; }
}

}  // namespace

namespace kao {
// edges[x * B + z] = the cheapest compound edge x -> z with the two rows behind it (ties: the lower (p, q)); stats as in kao.h
int pair_edges(const kao_topic *t, const uint16_t *assignment, int gmin, std::unordered_map<uint32_t, PairEdge> &edges, int64_t stats[4]) {
    edges.clear();
    const double t_begin = now_s();
    if (t->rf < 2) return fail(KAO_ERR_UNSUPPORTED, "compound edges need a follower (RF >= 2)");
    // the enumeration sweeps brokers x slots for every partition and the join is quadratic per leader pair -- 0.05 s
    // at 300 x 2000, minutes and gigabytes at 1000 x 30000.  Beyond kPairsMaxWork partition-broker pairs the layer reports no
    // edges (the caller sees a fixpoint), and it stops adding half-moves at kPairsMaxHalves.
    constexpr int64_t kPairsMaxWork = 4000000, kPairsMaxHalves = 4000000;
    if ((int64_t)t->n_partitions * t->n_brokers > kPairsMaxWork) {
        if (stats) { stats[0] = stats[1] = stats[2] = 0; stats[3] = 0; }
        return KAO_OK;
    }
    int32_t bd[8];
    derive_bounds(t, bd);
    PairCtx cx{t, assignment, t->n_brokers, t->n_racks, t->n_partitions, t->rf, t->rf_cur, bd[6], bd[7]};
    const int B = cx.B, P = cx.P, RF = cx.RF;
    std::vector<int32_t> best_cost((size_t)B * B, INT32_MAX);   // dense copy of the costs: most offers are dominated and stop here
    auto offer = [&](int x, int z, int cost, int p, const int *rowp, int q, const int *rowq) {
        int32_t &bc = best_cost[(size_t)x * B + z];
        if (cost > bc) return;
        bc = cost;
        PairEdge &e = edges[(uint32_t)x * (uint32_t)B + (uint32_t)z];   // value-initialised: set = 0
        if (e.set && (e.cost < cost || (e.cost == cost && (e.p < p || (e.p == p && e.q <= q))))) return;
        e.set = 1; e.cost = cost; e.p = p; e.q = q;
        for (int k = 0; k < RF; ++k) { e.rowp[k] = (uint16_t)rowp[k]; e.rowq[k] = (uint16_t)rowq[k]; }
    };
    std::vector<Half> halves;
    std::vector<std::vector<std::pair<int, int>>> by_leader((size_t)B);   // leader u -> (new leader v, half-move), sorted by v below
    std::vector<int> ycur;
    for (int p = 0; p < P; ++p) {
        int row[KAO_MAX_RF];
        for (int k = 0; k < RF; ++k) {
            row[k] = assignment[(size_t)p * RF + k];
        }
        const int u = row[0];
        int w0 = cx.wt(p, u, 0), wf_row[KAO_MAX_RF] = {0}, wf_sum = 0;
        for (int k = 1; k < RF; ++k) { wf_row[k] = cx.wt(p, row[k], 1); w0 += wf_row[k]; wf_sum += wf_row[k]; }
        ycur.clear();
        for (int k = 0; k < cx.rfc; ++k) {
            const int b = t->current[(size_t)p * cx.rfc + k];
            if (b < B && !contains(row, RF, b)) ycur.push_back(b);
        }
        auto add = [&](int v, const int *nrow, int o0, int o1, int i0, int i1) {
            int g = cx.wt(p, nrow[0], 0);
            for (int k = 1; k < RF; ++k) if (nrow[k] != kAny) g += cx.wt(p, nrow[k], 1);
            g -= w0;
            if (g < gmin || (int64_t)halves.size() >= kPairsMaxHalves) return;
            Half h{};
            h.p = p; h.gain = g;
            std::memcpy(h.row, nrow, sizeof(int) * (size_t)RF);
            h.no = (o0 != INT_MIN) + (o1 != INT_MIN); h.outs[0] = o0; h.outs[1] = o1;
            h.ni = (i0 != INT_MIN) + (i1 != INT_MIN); h.ins[0] = i0; h.ins[1] = i1;
            by_leader[(size_t)u].emplace_back(v, (int)halves.size());
            halves.push_back(h);
        };
        int nrow[KAO_MAX_RF];
        auto others_into = [&](int k, int from) {   // the followers except slot k, in slot order, from position `from`
            int n = from;
            for (int j = 1; j < RF; ++j) if (j != k) nrow[n++] = row[j];
        };
        for (int k = 1; k < RF; ++k) {
            const int v = row[k];
            nrow[0] = v; nrow[1] = u; others_into(k, 2);
            add(v, nrow, INT_MIN, INT_MIN, INT_MIN, INT_MIN);                       // a
            for (size_t yi = 0; yi <= ycur.size(); ++yi) {
                const int y = yi < ycur.size() ? ycur[yi] : kAny;
                nrow[0] = v; nrow[1] = y; others_into(k, 2);
                add(v, nrow, u, INT_MIN, y, INT_MIN);                              // b
            }
        }
        // the sweep over every broker pays only when some half-move towards a generic v (weight 0 as leader) can reach gmin
        int best_generic = wf_sum, y_best = 0;
        for (int y : ycur) y_best = std::max(y_best, cx.wt(p, y, 1));
        const int wfu = cx.wt(p, u, 1);
        for (int k = 1; k < RF; ++k) best_generic = std::max(best_generic, std::max(wfu, y_best) + wf_sum - wf_row[k]);
        best_generic -= w0;
        std::vector<int> cand_v;
        if (best_generic >= gmin) { for (int v = 0; v < B; ++v) cand_v.push_back(v); }
        else { cand_v = ycur; std::sort(cand_v.begin(), cand_v.end()); cand_v.erase(std::unique(cand_v.begin(), cand_v.end()), cand_v.end()); }
        for (int v : cand_v) {
            if (contains(row, RF, v)) continue;
            nrow[0] = v; for (int j = 1; j < RF; ++j) nrow[j] = row[j];
            add(v, nrow, u, INT_MIN, v, INT_MIN);                                  // c
            for (int k = 1; k < RF; ++k) {
                const int b = row[k];
                nrow[0] = v; nrow[1] = u; others_into(k, 2);
                add(v, nrow, b, INT_MIN, v, INT_MIN);                              // d
                for (size_t yi = 0; yi <= ycur.size(); ++yi) {
                    const int y = yi < ycur.size() ? ycur[yi] : kAny;
                    if (y == v) continue;
                    nrow[0] = v; nrow[1] = y; others_into(k, 2);
                    add(v, nrow, u, b, v, y);                                      // e
                }
            }
        }
    }
    int64_t n_pairs = 0, n_closed = 0;
    const double t_halves = now_s();
    for (auto &lst : by_leader) std::stable_sort(lst.begin(), lst.end(), [](const std::pair<int, int> &a, const std::pair<int, int> &b) { return a.first < b.first; });
    for (int u = 0; u < B; ++u) {
      const auto &lu = by_leader[(size_t)u];
      for (size_t g0 = 0; g0 < lu.size();) {
        const int v = lu[g0].first;
        size_t g1 = g0;
        while (g1 < lu.size() && lu[g1].first == v) ++g1;
        const size_t gb = g0, ge = g1;
        g0 = g1;
        if (u > v) continue;
        const auto &lv = by_leader[(size_t)v];   // the half-moves v -> u
        const auto r0 = std::lower_bound(lv.begin(), lv.end(), std::make_pair(u, INT_MIN)), r1 = std::lower_bound(lv.begin(), lv.end(), std::make_pair(u + 1, INT_MIN));
        if (r0 == r1) continue;
        for (size_t gi = gb; gi < ge; ++gi) {
            const Half &hp = halves[(size_t)lu[gi].second];
            for (auto rj = r0; rj != r1; ++rj) {
                const Half &hq = halves[(size_t)rj->second];
                if (hp.p == hq.p) continue;
                ++n_pairs;
                int outs[4], ins[4], no = 0, ni = 0;
                for (int i = 0; i < hp.no; ++i) outs[no++] = hp.outs[i];
                for (int i = 0; i < hq.no; ++i) outs[no++] = hq.outs[i];
                for (int i = 0; i < hp.ni; ++i) ins[ni++] = hp.ins[i];
                for (int i = 0; i < hq.ni; ++i) ins[ni++] = hq.ins[i];
                {   // concrete brokers that leave one row and enter the other cancel
                    int orig[4]; const int n0 = no;
                    std::memcpy(orig, outs, sizeof orig);
                    for (int i = 0; i < n0; ++i)
                        if (orig[i] != kAny && contains(ins, ni, orig[i])) { remove_first(outs, no, orig[i]); remove_first(ins, ni, orig[i]); }
                }
                int rp[KAO_MAX_RF], rq[KAO_MAX_RF];
                std::memcpy(rp, hp.row, sizeof(int) * (size_t)RF); std::memcpy(rq, hq.row, sizeof(int) * (size_t)RF);
                auto bind = [&](int *nr, const Half &partner) {   // a generic entering follower takes a broker the partner releases
                    if (!contains(nr, RF, kAny) || !contains(ins, ni, kAny)) return;
                    for (int i = 0; i < partner.no; ++i) {
                        const int x = partner.outs[i];
                        if (!contains(outs, no, x) || contains(nr, RF, x)) continue;
                        int cand[KAO_MAX_RF];
                        for (int k = 0; k < RF; ++k) cand[k] = nr[k] == kAny ? x : nr[k];
                        if (!cx.c7_ok(cand)) continue;
                        remove_first(outs, no, x); remove_first(ins, ni, kAny);
                        std::memcpy(nr, cand, sizeof(int) * (size_t)RF);
                        return;
                    }
                };
                bind(rp, hq); bind(rq, hp);
                if (no > 1 || ni > 1) continue;
                const int gain = hp.gain + hq.gain;
                const bool ap = contains(rp, RF, kAny), aq = contains(rq, RF, kAny);
                if (ap || aq) {   // one generic follower left: an edge to every z that keeps its row inside C7
                    if (no != 1 || ni != 1 || ins[0] != kAny || (ap && aq)) continue;
                    const int x = outs[0];
                    const int *rowt = ap ? rp : rq, *other = ap ? rq : rp;
                    if (!cx.c7_ok(other)) continue;
                    // which racks may the generic follower come from?  (the row without it: distinct brokers, no rack above its
                    // band, at most one rack below it -- and then only that rack completes the row)
                    int cnt[256] = {0}, ka = -1;
                    bool fixed_ok = true;
                    for (int k = 0; k < RF && fixed_ok; ++k) {
                        if (rowt[k] == kAny) { ka = k; continue; }
                        for (int j = 0; j < k; ++j) if (rowt[j] == rowt[k]) fixed_ok = false;
                        ++cnt[t->rack_of[rowt[k]]];
                    }
                    int deficient = 0;
                    for (int r = 0; r < cx.R && fixed_ok; ++r) { if (cnt[r] > cx.phi) fixed_ok = false; deficient += cnt[r] < cx.plo; }
                    if (!fixed_ok || deficient > 1) continue;
                    int cand[KAO_MAX_RF];
                    for (int k = 0; k < RF; ++k) cand[k] = rowt[k];
                    for (int z = 0; z < B; ++z) {
                        if (z == x) continue;
                        const int rz = t->rack_of[z];
                        if (cnt[rz] + 1 > cx.phi || (deficient == 1 && !(cnt[rz] < cx.plo && cnt[rz] + 1 >= cx.plo))) continue;
                        if (contains(rowt, RF, z)) continue;
                        cand[ka] = z;
                        if (ap) offer(x, z, -gain, hp.p, cand, hq.p, rq); else offer(x, z, -gain, hp.p, rp, hq.p, cand);
                    }
                    continue;
                }
                if (!cx.c7_ok(rp) || !cx.c7_ok(rq)) continue;
                if (no == 0 && ni == 0) { n_closed += gain > 0; continue; }
                if (no == 1 && ni == 1) offer(outs[0], ins[0], -gain, hp.p, rp, hq.p, rq);
            }
        }
      }
    }
    if (stats) { stats[0] = (int64_t)halves.size(); stats[1] = n_pairs; stats[2] = (int64_t)edges.size(); stats[3] = n_closed; }
    if (std::getenv("KAO_CX_TRACE"))
        std::fprintf(stderr, "[kao-cx] pairs: %zu half-moves in %.1f ms, %lld pairs -> %zu compound edges in %.1f ms\n", halves.size(),
                     (t_halves - t_begin) * 1e3, (long long)n_pairs, edges.size(), (now_s() - t_halves) * 1e3);
    return KAO_OK;
}
}  // namespace kao

extern "C" int kao_cycle_pair_edges(const kao_topic *t, const uint16_t *assignment, int32_t gmin, int32_t *cost, int64_t stats[4]) {
    if (!t || !assignment || !cost) return fail(KAO_ERR_INVALID, "null topic, assignment or cost matrix");
    int rc = validate(t);
    if (rc) return rc;
    const int B = t->n_brokers;
    for (int p = 0; p < t->n_partitions; ++p)
        for (int k = 0; k < t->rf; ++k)
            if (assignment[(size_t)p * t->rf + k] >= B) return fail(KAO_ERR_INVALID, "the assignment has an empty or out-of-range slot");
    std::unordered_map<uint32_t, PairEdge> edges;
    if ((rc = pair_edges(t, assignment, gmin, edges, stats))) return rc;
    for (size_t i = 0; i < (size_t)B * B; ++i) cost[i] = INT32_MAX;
    for (const auto &kv : edges) cost[kv.first] = kv.second.cost;
    return KAO_OK;
}
