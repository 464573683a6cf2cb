// KAO-LP, the primal side (round 5): from the quantised iterate of the PERTURBED LP (k_lp_round) to an assignment.  Host code -- one
// pass over the partitions, O(P * RF * rack size), plus a bounded search over the handful of partitions with fractional variables.
// Specification: oracle/kao_lp.py round_primal (same quantisation, same order, same ties); the model it rounds: README.md:144-185,
// compact form in DESIGN.md section 4b'.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <functional>
#include <tuple>
#include <vector>

#include "kao_host.h"
#include "kao_internal.h"

namespace kao {

namespace {
constexpr int kTolC = 30;          // a variable farther than 0.30 from an integer makes its partition fractional
constexpr int kMaxCand = 16;       // candidate brokers of a fractional partition
constexpr int kMaxRows = 256;      // candidate rows kept per fractional partition (by objective weight)
constexpr long kMaxNodes = 100000; // search nodes over the fractional partitions
constexpr int kMaxSearch = 64;     // more fractional partitions than this: the iterate is far from a vertex, no search
constexpr uint16_t kUnset = 0xFFFFu;

struct Row { int w; int n; int b[KAO_MAX_RF]; };
}  // namespace

int lp_round_assignment(const kao_topic *t, const uint8_t *q, const int32_t *zq, const uint16_t *fallback, uint16_t *out, int32_t rep[4]) {
    const int P = t->n_partitions, B = t->n_brokers, R = t->n_racks, RF = t->rf, NJ = t->rf_cur;
    int32_t bd[8];
    derive_bounds(t, bd);
    const int lo = bd[0], hi = bd[1], llo = bd[2], lhi = bd[3], phi = bd[7];
    std::vector<int> load((size_t)B, 0), lead_load((size_t)B, 0);
    auto in_row = [&](int p, int b) { for (int k = 0; k < RF; ++k) if (out[(size_t)p * RF + k] == (uint16_t)b) return true; return false; };
    do {   // (q == nullptr: `out` holds a complete assignment already and only the band repair at the end runs -- the test hook's mode 2)
    if (!q) break;
    std::vector<std::vector<int>> members((size_t)R);
    for (int b = 0; b < B; ++b) members[t->rack_of[b]].push_back(b);
    std::vector<int> cap[2] = {std::vector<int>(zq, zq + B), std::vector<int>(zq + B, zq + 2 * B)};   // [0] follower inflow, [1] leader inflow
    std::vector<std::vector<std::pair<int, int>>> placed[2];   // per kind, per rack: (partition, slot) of the new replicas handed out
    placed[0].resize((size_t)R); placed[1].resize((size_t)R);
    int over = 0, unplaced = 0, from_fb = 0, swaps = 0;
    auto Q = [&](int k, int p) { return (int)q[(size_t)k * P + p]; };
    auto frac = [&](int c) { const int d = c - 100 * ((c + 50) / 100); return (d < 0 ? -d : d) > kTolC; };
    auto unit = [&](int c) { return (c + 50) / 100; };
    int used[2 * KAO_MAX_RF + 2]; int n_used = 0;
    auto is_used = [&](int b) { for (int i = 0; i < n_used; ++i) if (used[i] == b) return true; return false; };
    struct Swap { int q, sq, b1; };
    std::vector<Swap> swaps_now;
    // a new replica of kind `kind` in rack r for the row under construction (brokers in `used`): (broker, inside the inflows?)
    auto take = [&](int r, int kind, bool &within) {
        std::vector<int> &c = cap[kind];
        within = true;
        int best = -1;
        for (int b : members[(size_t)r])
            if (c[(size_t)b] > 0 && !is_used(b) && (best < 0 || c[(size_t)b] > c[(size_t)best])) best = b;
        if (best >= 0) { c[(size_t)best]--; return best; }
        for (int b1 : members[(size_t)r]) {   // one swap: an earlier partition moves its new replica from b1 to a broker with inflow left
            if (is_used(b1)) continue;
            for (auto &e : placed[kind][(size_t)r]) {
                if (out[(size_t)e.first * RF + e.second] != (uint16_t)b1) continue;
                for (int b2 : members[(size_t)r])
                    if (c[(size_t)b2] > 0 && !in_row(e.first, b2)) {
                        c[(size_t)b2]--; out[(size_t)e.first * RF + e.second] = (uint16_t)b2; ++swaps;
                        swaps_now.push_back({e.first, e.second, b1});
                        return b1;
                    }
            }
        }
        within = false;
        for (int b : members[(size_t)r])
            if (!is_used(b)) { ++over; return b; }
        ++unplaced;
        return -1;
    };
    std::vector<int> pending, cur((size_t)NJ), row;
    std::vector<std::tuple<int, int, int>> undo;          // (kind, broker, units) taken by the row under construction
    std::vector<std::tuple<int, int, int>> new_slots;     // (rack, kind, slot)
    for (int p = 0; p < P; ++p) {
        bool fr = false;
        for (int j = 0; j < NJ; ++j) {
            const unsigned b = t->current[(size_t)p * NJ + j];
            cur[(size_t)j] = (b == KAO_NONE || (int)b >= B) ? -1 : (int)b;
            if (cur[(size_t)j] >= 0) fr |= frac(Q(j, p)) || frac(Q(NJ + j, p));
        }
        for (int r = 0; r < R && !fr; ++r) fr |= frac(Q(2 * NJ + r, p)) || frac(Q(2 * NJ + R + r, p));
        if (fr) { pending.push_back(p); continue; }
        int lead = -1;
        row.clear(); n_used = 0;
        for (int j = 0; j < NJ; ++j) {
            if (cur[(size_t)j] < 0) continue;
            if (lead < 0 && unit(Q(NJ + j, p)) >= 1) lead = cur[(size_t)j];
            else if (unit(Q(j, p)) >= 1) row.push_back(cur[(size_t)j]);
        }
        const bool ok = (int)row.size() + (lead >= 0 ? 1 : 0) <= RF && (int)row.size() <= RF - 1;
        if (ok) { for (int b : row) used[n_used++] = b; if (lead >= 0) used[n_used++] = lead; }
        const int over0 = over, unplaced0 = unplaced, swaps0 = swaps;
        undo.clear(); new_slots.clear(); swaps_now.clear();
        for (int k = 0; k < RF; ++k) out[(size_t)p * RF + k] = kUnset;
        for (int r = 0; r < R && ok; ++r) {
            if (lead < 0 && unit(Q(2 * NJ + R + r, p)) >= 1) {
                bool within; const size_t ns = swaps_now.size(); const int b = take(r, 1, within);
                if (b >= 0) {
                    lead = b; used[n_used++] = b; out[(size_t)p * RF] = (uint16_t)b;
                    if (within) { new_slots.emplace_back(r, 1, 0); if (swaps_now.size() == ns) undo.emplace_back(1, b, 1); }
                }
            }
            for (int n = unit(Q(2 * NJ + r, p)); n > 0; --n) {
                if ((int)row.size() >= RF - 1) break;
                bool within; const size_t ns = swaps_now.size(); const int b = take(r, 0, within);
                if (b >= 0) {
                    row.push_back(b); used[n_used++] = b; out[(size_t)p * RF + row.size()] = (uint16_t)b;
                    if (within) { new_slots.emplace_back(r, 0, (int)row.size()); if (swaps_now.size() == ns) undo.emplace_back(0, b, 1); }
                }
            }
        }
        if (!ok || lead < 0 || (int)row.size() != RF - 1) {   // incomplete: give back what the row took (its swaps included), treat as fractional
            for (auto &u : undo) cap[std::get<0>(u)][(size_t)std::get<1>(u)] += std::get<2>(u);
            for (size_t i = swaps_now.size(); i-- > 0;) {
                const Swap &sw = swaps_now[i];
                // the swap moved (q, sq) from b1 to some b2 and took one unit of b2's inflow: find the kind by the list it sits in
                const int b2 = out[(size_t)sw.q * RF + sw.sq];
                const int kind = sw.sq == 0 ? 1 : 0;
                cap[kind][(size_t)b2]++;
                out[(size_t)sw.q * RF + sw.sq] = (uint16_t)sw.b1;
            }
            over = over0; unplaced = unplaced0; swaps = swaps0;
            pending.push_back(p);
            continue;
        }
        out[(size_t)p * RF] = (uint16_t)lead;
        for (int k = 1; k < RF; ++k) out[(size_t)p * RF + k] = (uint16_t)row[(size_t)k - 1];
        for (auto &ns : new_slots) placed[std::get<1>(ns)][(size_t)std::get<0>(ns)].emplace_back(p, std::get<2>(ns));
    }
    if (rep) { rep[0] = (int32_t)pending.size(); }
    if (fallback) {
        for (int p : pending) { std::memcpy(out + (size_t)p * RF, fallback + (size_t)p * RF, (size_t)RF * 2); ++from_fb; }
        if (rep) { rep[1] = over; rep[2] = unplaced; rep[3] = from_fb; }
        return KAO_OK;
    }
    // ---- the fractional partitions, together: candidate rows from their support, chosen by a bounded depth-first search so that the
    //      band rows (README.md:158-166) come out right given what the other partitions hold ----
    std::vector<char> is_pending((size_t)P, 0);
    for (int p : pending) is_pending[(size_t)p] = 1;
    for (int p = 0; p < P; ++p) {
        if (is_pending[(size_t)p]) continue;
        for (int k = 0; k < RF; ++k) load[out[(size_t)p * RF + k]]++;
        lead_load[out[(size_t)p * RF]]++;
    }
    const size_t np = pending.size();
    std::vector<std::vector<Row>> rows_of(np);
    std::vector<int> cand, wl((size_t)B, 0), wf((size_t)B, 0), order;
    for (size_t i = 0; i < np; ++i) {
        const int p = pending[i];
        for (int j = 0; j < NJ; ++j) { const unsigned b = t->current[(size_t)p * NJ + j]; cur[(size_t)j] = (b == KAO_NONE || (int)b >= B) ? -1 : (int)b; }
        cand.clear();
        auto in_cand = [&](int b) { return std::find(cand.begin(), cand.end(), b) != cand.end(); };
        for (int j = 0; j < NJ; ++j)
            if (cur[(size_t)j] >= 0 && (Q(j, p) > 0 || Q(NJ + j, p) > 0)) cand.push_back(cur[(size_t)j]);
        for (int r = 0; r < R; ++r) {
            if (!(Q(2 * NJ + r, p) > 0 || Q(2 * NJ + R + r, p) > 0)) continue;
            order.clear();
            for (int b : members[(size_t)r]) if (!in_cand(b) && load[(size_t)b] < hi) order.push_back(b);
            std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
                const int ka = lo - load[(size_t)a], kb = lo - load[(size_t)b];
                if (ka != kb) return ka > kb;
                const int la = llo - lead_load[(size_t)a], lb = llo - lead_load[(size_t)b];
                if (la != lb) return la > lb;
                return a < b;
            });
            size_t n_short = 0;
            for (int b : order) n_short += load[(size_t)b] < lo;
            const size_t keep = std::max<size_t>(2, std::min<size_t>(8, n_short));   // every broker of the rack that is below its band (up to 8), two at least
            for (size_t k = 0; k < order.size() && k < keep; ++k) cand.push_back(order[k]);
        }
        if ((int)cand.size() > kMaxCand) cand.resize(kMaxCand);
        if ((int)cand.size() < RF) {   // (mass on too few options: the brokers that need replicas most)
            order.clear();
            for (int b = 0; b < B; ++b) if (!in_cand(b)) order.push_back(b);
            std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
                const int ka = lo - load[(size_t)a], kb = lo - load[(size_t)b];
                return ka != kb ? ka > kb : a < b;
            });
            for (int b : order) { cand.push_back(b); if ((int)cand.size() >= RF + 2) break; }
        }
        for (int b : cand) {
            const int bw = t->broker_w ? t->broker_w[b] : 0, bwl = t->broker_wl ? t->broker_wl[b] : 0;
            wf[(size_t)b] = bw; wl[(size_t)b] = bw + bwl;
        }
        for (int j = 0; j < NJ; ++j) {
            const int b = cur[(size_t)j];
            if (b >= 0 && in_cand(b)) { wl[(size_t)b] += t->w[j == 0 ? 0 : 1][0]; wf[(size_t)b] += t->w[j == 0 ? 0 : 1][1]; }
        }
        // rows: leader candidates in candidate order, follower combinations of the others in lexicographic (position) order
        std::vector<Row> &rows = rows_of[i];
        const int nc = (int)cand.size(), nf = RF - 1;
        std::vector<int> others, idx((size_t)nf), per_rack((size_t)R);
        for (int li = 0; li < nc; ++li) {
            others.clear();
            for (int k = 0; k < nc; ++k) if (k != li) others.push_back(cand[(size_t)k]);
            const int no = (int)others.size();
            if (no < nf) continue;
            for (int k = 0; k < nf; ++k) idx[(size_t)k] = k;
            for (;;) {
                std::fill(per_rack.begin(), per_rack.end(), 0);
                bool okr = true;
                Row rw; rw.n = RF; rw.b[0] = cand[(size_t)li]; rw.w = wl[(size_t)rw.b[0]];
                per_rack[t->rack_of[rw.b[0]]]++;
                for (int k = 0; k < nf; ++k) {
                    const int b = others[(size_t)idx[(size_t)k]];
                    rw.b[k + 1] = b; rw.w += wf[(size_t)b];
                    if (++per_rack[t->rack_of[b]] > phi) okr = false;
                }
                if (per_rack[t->rack_of[rw.b[0]]] > phi) okr = false;
                if (okr) rows.push_back(rw);
                int k = nf - 1;
                while (k >= 0 && idx[(size_t)k] == no - nf + k) --k;
                if (k < 0) break;
                ++idx[(size_t)k];
                for (int m = k + 1; m < nf; ++m) idx[(size_t)m] = idx[(size_t)m - 1] + 1;
            }
        }
        std::stable_sort(rows.begin(), rows.end(), [](const Row &a, const Row &b) { return a.w > b.w; });
        if ((int)rows.size() > kMaxRows) rows.resize(kMaxRows);
    }
    long nodes = 0;
    bool have_best = false;
    long best_viol = 0, best_obj = 0;
    std::vector<int> pick(np, -1), best_pick(np, -1);
    auto apply = [&](const Row &rw, int d) { for (int k = 0; k < rw.n; ++k) load[(size_t)rw.b[k]] += d; lead_load[(size_t)rw.b[0]] += d; };
    auto admissible = [&](const Row &rw) {
        if (lead_load[(size_t)rw.b[0]] >= lhi) return false;
        for (int k = 0; k < rw.n; ++k) if (load[(size_t)rw.b[k]] >= hi) return false;
        return true;
    };
    if (np > (size_t)kMaxSearch) {   // far from a vertex: no search, the first admissible row of every partition in turn
        for (size_t i = 0; i < np; ++i) {
            const std::vector<Row> &rows = rows_of[i];
            int ri = 0;
            for (size_t k = 0; k < rows.size(); ++k) if (admissible(rows[k])) { ri = (int)k; break; }
            best_pick[i] = ri;
            if (!rows.empty()) apply(rows[(size_t)ri], +1);
        }
        have_best = true;
    } else {
        // cover[i][b] / lcover[i][b]: how many of the partitions i.. can still put a replica / their leader on broker b -- a deficit beyond
        // that stays whatever the rest of the search does; wmax[i]: the most the partitions i.. can add to the objective
        std::vector<int> shortb, lshortb;
        for (int b = 0; b < B; ++b) { if (load[(size_t)b] < lo) shortb.push_back(b); if (lead_load[(size_t)b] < llo) lshortb.push_back(b); }
        std::vector<std::vector<int>> cover(np + 1, std::vector<int>((size_t)B, 0)), lcover(np + 1, std::vector<int>((size_t)B, 0));
        std::vector<long> wmax(np + 1, 0);
        {
            std::vector<char> seen((size_t)B), seenl((size_t)B);
            for (size_t i = np; i-- > 0;) {
                cover[i] = cover[i + 1]; lcover[i] = lcover[i + 1];
                std::fill(seen.begin(), seen.end(), 0); std::fill(seenl.begin(), seenl.end(), 0);
                for (const Row &rw : rows_of[i]) {
                    for (int k = 0; k < rw.n; ++k) if (!seen[(size_t)rw.b[k]]) { seen[(size_t)rw.b[k]] = 1; cover[i][(size_t)rw.b[k]]++; }
                    if (!seenl[(size_t)rw.b[0]]) { seenl[(size_t)rw.b[0]] = 1; lcover[i][(size_t)rw.b[0]]++; }
                }
                wmax[i] = wmax[i + 1] + (rows_of[i].empty() ? 0 : rows_of[i][0].w);
            }
        }
        auto lower_bound = [&](size_t i) {
            long v = 0;
            for (int b : shortb) { const int d = lo - load[(size_t)b] - cover[i][(size_t)b]; if (d > 0) v += d; }
            for (int b : lshortb) { const int d = llo - lead_load[(size_t)b] - lcover[i][(size_t)b]; if (d > 0) v += d; }
            return v;
        };
        auto leaf_viol = [&]() {
            long v = 0;
            for (int b = 0; b < B; ++b) { v += std::max(lo - load[(size_t)b], 0); v += std::max(llo - lead_load[(size_t)b], 0); }
            return v;
        };
        bool perfect = false;
        std::function<void(size_t, long)> dfs = [&](size_t i, long obj) {
            if (nodes > kMaxNodes) return;
            ++nodes;
            if (i == np) {
                const long v = leaf_viol();
                if (!have_best || v < best_viol || (v == best_viol && obj > best_obj)) { have_best = true; best_viol = v; best_obj = obj; best_pick = pick; }
                return;
            }
            const long lb = lower_bound(i);
            if (perfect && lb > 0) return;   // (first pass: only completions that leave no broker below its band)
            if (have_best && (lb > best_viol || (lb == best_viol && obj + wmax[i] <= best_obj))) return;   // cannot beat the best so far
            const std::vector<Row> &rows = rows_of[i];
            bool any = false;
            for (size_t k = 0; k < rows.size(); ++k) {
                if (!admissible(rows[k])) continue;
                any = true;
                apply(rows[k], +1);
                pick[i] = (int)k;
                dfs(i + 1, obj + rows[k].w);
                apply(rows[k], -1);
            }
            if (!any && !rows.empty()) {   // every row breaks an upper band end: take the first, K-eval counts the violation
                apply(rows[0], +1);
                pick[i] = 0;
                dfs(i + 1, obj + rows[0].w);
                apply(rows[0], -1);
            } else if (!any) { pick[i] = -1; dfs(i + 1, obj); }
        };
        perfect = true;
        dfs(0, 0);
        if (!have_best || best_viol > 0) {   // no completion without a violation among the candidate rows (or not found in time): the least violated one
            perfect = false; nodes = 0;
            dfs(0, 0);
        }
    }
    for (size_t i = 0; i < np; ++i) {
        const int p = pending[i];
        const std::vector<Row> &rows = rows_of[i];
        const int ri = have_best && best_pick[i] >= 0 ? best_pick[i] : 0;
        if (rows.empty()) { for (int k = 0; k < RF; ++k) out[(size_t)p * RF + k] = (uint16_t)k; continue; }
        for (int k = 0; k < RF; ++k) out[(size_t)p * RF + k] = (uint16_t)rows[(size_t)ri].b[k];
    }
    if (rep) { rep[1] = over; rep[2] = unplaced; rep[3] = from_fb; }
    (void)swaps;
    } while (0);
    // ---- what the completion of a half-integral vertex leaves: a few brokers one replica (or one leadership) over their band, as many
    //      under it.  Moves that cost nothing put that right (specification: oracle/kao_lp.py repair_bands) ----
    {
        std::fill(load.begin(), load.end(), 0); std::fill(lead_load.begin(), lead_load.end(), 0);
        for (int p = 0; p < P; ++p) { for (int k = 0; k < RF; ++k) load[out[(size_t)p * RF + k]]++; lead_load[out[(size_t)p * RF]]++; }
        bool fine = true;
        for (int b = 0; b < B && fine; ++b) fine = load[(size_t)b] >= lo && load[(size_t)b] <= hi && lead_load[(size_t)b] >= llo && lead_load[(size_t)b] <= lhi;
        if (!fine) {
            auto wts = [&](int p, int b, int &wlo, int &wfo) {
                wlo = (t->broker_w ? t->broker_w[b] : 0) + (t->broker_wl ? t->broker_wl[b] : 0); wfo = t->broker_w ? t->broker_w[b] : 0;
                for (int j = 0; j < NJ; ++j)
                    if ((int)t->current[(size_t)p * NJ + j] == b) { wlo += t->w[j == 0 ? 0 : 1][0]; wfo += t->w[j == 0 ? 0 : 1][1]; }
            };
            std::vector<int> overb;
            bool any_under = false;
            for (int b = 0; b < B; ++b) { if (load[(size_t)b] > hi) overb.push_back(b); any_under |= load[(size_t)b] < lo; }
            if (!overb.empty() || any_under) {
                std::vector<char> is_src((size_t)B, 0);
                for (int b : overb) is_src[(size_t)b] = 1;
                std::vector<std::vector<std::pair<int, int>>> holds((size_t)B);
                for (int p = 0; p < P; ++p)
                    for (int k = 1; k < RF; ++k) { const int b = out[(size_t)p * RF + k]; if (is_src[(size_t)b]) holds[(size_t)b].emplace_back(p, k); }
                std::vector<int> targets;
                for (int b1 : overb) {
                    while (load[(size_t)b1] > hi) {
                        bool moved = false;
                        targets.clear();
                        for (int b = 0; b < B; ++b) if (load[(size_t)b] < lo) targets.push_back(b);
                        if (targets.empty()) for (int b = 0; b < B; ++b) if (load[(size_t)b] < hi && b != b1) targets.push_back(b);
                        std::stable_sort(targets.begin(), targets.end(), [&](int a, int b) {
                            const bool oa = t->rack_of[a] != t->rack_of[b1], ob = t->rack_of[b] != t->rack_of[b1];
                            return oa != ob ? !oa : a < b;
                        });
                        for (int b2 : targets) {
                            for (auto &h : holds[(size_t)b1]) {
                                const int p = h.first, k = h.second;
                                if (out[(size_t)p * RF + k] != (uint16_t)b1) continue;
                                if (in_row(p, b2)) continue;
                                int wl1, wf1, wl2, wf2; wts(p, b1, wl1, wf1); wts(p, b2, wl2, wf2);
                                if (wf1 != wf2) continue;
                                if (t->rack_of[b2] != t->rack_of[b1]) {
                                    int cnt = 0;
                                    for (int m = 0; m < RF; ++m) cnt += t->rack_of[out[(size_t)p * RF + m]] == t->rack_of[b2];
                                    if (cnt >= phi) continue;
                                }
                                out[(size_t)p * RF + k] = (uint16_t)b2; load[(size_t)b1]--; load[(size_t)b2]++; moved = true;
                                break;
                            }
                            if (moved) break;
                        }
                        if (!moved) break;
                    }
                }
                // what is left costs weight: the cheapest follower move of every broker still over its band (first minimal loss in (p, k), target order)
                for (int b1 : overb) {
                    while (load[(size_t)b1] > hi) {
                        targets.clear();
                        for (int b = 0; b < B; ++b) if (load[(size_t)b] < lo) targets.push_back(b);
                        if (targets.empty()) for (int b = 0; b < B; ++b) if (load[(size_t)b] < hi && b != b1) targets.push_back(b);
                        std::stable_sort(targets.begin(), targets.end(), [&](int a, int b) {
                            const bool oa = t->rack_of[a] != t->rack_of[b1], ob = t->rack_of[b] != t->rack_of[b1];
                            return oa != ob ? !oa : a < b;
                        });
                        bool have = false; int bl = 0, bp = 0, bk = 0, bb = 0;
                        for (auto &h : holds[(size_t)b1]) {
                            const int p = h.first, k = h.second;
                            if (out[(size_t)p * RF + k] != (uint16_t)b1) continue;
                            int wl1, wf1; wts(p, b1, wl1, wf1);
                            for (int b2 : targets) {
                                if (in_row(p, b2)) continue;
                                if (t->rack_of[b2] != t->rack_of[b1]) {
                                    int cnt = 0;
                                    for (int m = 0; m < RF; ++m) cnt += t->rack_of[out[(size_t)p * RF + m]] == t->rack_of[b2];
                                    if (cnt >= phi) continue;
                                }
                                int wl2, wf2; wts(p, b2, wl2, wf2);
                                const int loss = wf1 - wf2;
                                if (!have || loss < bl) { have = true; bl = loss; bp = p; bk = k; bb = b2; }
                            }
                        }
                        if (!have) break;
                        out[(size_t)bp * RF + bk] = (uint16_t)bb; load[(size_t)b1]--; load[(size_t)bb]++;
                    }
                }
            }
            std::vector<int> overl;
            for (int b = 0; b < B; ++b) if (lead_load[(size_t)b] > lhi) overl.push_back(b);
            for (int b1 : overl) {
                while (lead_load[(size_t)b1] > lhi) {
                    bool moved = false, under = false;
                    for (int b = 0; b < B && !under; ++b) under = lead_load[(size_t)b] < llo;
                    const int cap2 = under ? llo : lhi;   // the taker is below its band when anyone is, else below the upper end
                    for (int p = 0; p < P && !moved; ++p) {
                        if (out[(size_t)p * RF] != (uint16_t)b1) continue;
                        for (int k = 1; k < RF; ++k) {
                            const int b2 = out[(size_t)p * RF + k];
                            if (lead_load[(size_t)b2] >= cap2) continue;
                            int wl1, wf1, wl2, wf2; wts(p, b1, wl1, wf1); wts(p, b2, wl2, wf2);
                            if (wl1 + wf2 != wl2 + wf1) continue;
                            out[(size_t)p * RF] = (uint16_t)b2; out[(size_t)p * RF + k] = (uint16_t)b1; lead_load[(size_t)b1]--; lead_load[(size_t)b2]++; moved = true;
                            break;
                        }
                    }
                    if (!moved) break;
                }
            }
            overl.clear();
            for (int b = 0; b < B; ++b) if (lead_load[(size_t)b] > lhi) overl.push_back(b);
            for (int b1 : overl) {   // and the cheapest role swap of every broker still leading too many
                while (lead_load[(size_t)b1] > lhi) {
                    bool under = false;
                    for (int b = 0; b < B && !under; ++b) under = lead_load[(size_t)b] < llo;
                    const int cap2 = under ? llo : lhi;
                    bool have = false; int bl = 0, bp = 0, bk = 0, bb = 0;
                    for (int p = 0; p < P; ++p) {
                        if (out[(size_t)p * RF] != (uint16_t)b1) continue;
                        int wl1, wf1; wts(p, b1, wl1, wf1);
                        for (int k = 1; k < RF; ++k) {
                            const int b2 = out[(size_t)p * RF + k];
                            if (lead_load[(size_t)b2] >= cap2) continue;
                            int wl2, wf2; wts(p, b2, wl2, wf2);
                            const int loss = (wl1 + wf2) - (wl2 + wf1);
                            if (!have || loss < bl) { have = true; bl = loss; bp = p; bk = k; bb = b2; }
                        }
                    }
                    if (!have) break;
                    out[(size_t)bp * RF] = (uint16_t)bb; out[(size_t)bp * RF + bk] = (uint16_t)b1; lead_load[(size_t)b1]--; lead_load[(size_t)bb]++;
                }
            }
            // a broker that still leads too many and shares no partition with one that may take a leadership (2,000 brokers: the usual
            // case): a CHAIN of role swaps, breadth first over "u leads p, v follows in p" (partitions ascending, slots ascending),
            // weight-neutral swaps only in the first attempt, any swap in the second; every broker between the ends keeps its count
            overl.clear();
            for (int b = 0; b < B; ++b) if (lead_load[(size_t)b] > lhi) overl.push_back(b);
            struct Par { int u, p, k; };
            std::vector<Par> parent((size_t)B);
            std::vector<char> seen((size_t)B);
            std::vector<int> queue;
            std::vector<std::vector<int>> leads_of((size_t)B);
            for (int b1 : overl) {
                while (lead_load[(size_t)b1] > lhi) {
                    bool under = false;
                    for (int b = 0; b < B && !under; ++b) under = lead_load[(size_t)b] < llo;
                    const int cap2 = under ? llo : lhi;
                    int end = -1;
                    for (int attempt = 0; attempt < 2 && end < 0; ++attempt) {
                        const bool neutral_only = attempt == 0;
                        for (auto &v : leads_of) v.clear();
                        for (int p = 0; p < P; ++p) leads_of[out[(size_t)p * RF]].push_back(p);
                        std::fill(seen.begin(), seen.end(), 0);
                        queue.clear(); queue.push_back(b1); seen[(size_t)b1] = 1; parent[(size_t)b1] = {-1, -1, -1};
                        for (size_t qi = 0; qi < queue.size() && end < 0; ++qi) {
                            const int u = queue[qi];
                            for (int p : leads_of[(size_t)u]) {
                                int wl1, wf1; wts(p, u, wl1, wf1);
                                for (int k = 1; k < RF; ++k) {
                                    const int v = out[(size_t)p * RF + k];
                                    if (seen[(size_t)v]) continue;
                                    if (neutral_only) { int wl2, wf2; wts(p, v, wl2, wf2); if (wl1 + wf2 != wl2 + wf1) continue; }
                                    seen[(size_t)v] = 1; parent[(size_t)v] = {u, p, k}; queue.push_back(v);
                                    if (lead_load[(size_t)v] < cap2) { end = v; break; }
                                }
                                if (end >= 0) break;
                            }
                        }
                    }
                    if (end < 0) break;
                    for (int v = end; parent[(size_t)v].u >= 0; v = parent[(size_t)v].u) {
                        const Par &e = parent[(size_t)v];
                        out[(size_t)e.p * RF] = (uint16_t)v; out[(size_t)e.p * RF + e.k] = (uint16_t)e.u;
                    }
                    lead_load[(size_t)b1]--; lead_load[(size_t)end]++;
                }
            }
        }
    }
    return KAO_OK;
}

}  // namespace kao
