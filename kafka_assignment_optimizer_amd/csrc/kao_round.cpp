// KAO-LP, the primal side (round 5): from the quantised iterate of the PERTURBED LP (k_lp_round) to an assignment.  Host code -- one
// pass over the partitions, O(P * RF * rack size).  Specification: oracle/kao_lp.py round_primal (same quantisation, same order, same
// ties); the model it rounds: README.md:144-185, compact form in DESIGN.md section 4b'.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <tuple>
#include <vector>

#include "kao_host.h"
#include "kao_internal.h"

namespace kao {

int lp_round_assignment(const kao_topic *t, const uint8_t *q, const int32_t *zq, const uint16_t *fallback, uint16_t *out, int32_t rep[4]) {
    const int P = t->n_partitions, B = t->n_brokers, R = t->n_racks, RF = t->rf, NJ = t->rf_cur;
    int32_t bd[8];
    derive_bounds(t, bd);
    const int phi = bd[7];
    constexpr int kTolC = 30;   // a variable farther than 0.30 from an integer makes its partition fractional
    std::vector<std::vector<int>> members((size_t)R);
    for (int b = 0; b < B; ++b) members[t->rack_of[b]].push_back(b);
    std::vector<int> capf(zq, zq + B), capl(zq + B, zq + 2 * B);
    int over = 0, unplaced = 0, from_fb = 0;
    auto Q = [&](int k, int p) { return (int)q[(size_t)k * P + p]; };
    auto frac = [&](int c) { const int d = c - 100 * ((c + 50) / 100); return (d < 0 ? -d : d) > kTolC; };
    auto unit = [&](int c) { return (c + 50) / 100; };
    int used[16]; int n_used = 0;
    auto is_used = [&](int b) { for (int i = 0; i < n_used; ++i) if (used[i] == b) return true; return false; };
    auto take = [&](int r, std::vector<int> &cap) {
        int best = -1;
        for (int b : members[(size_t)r])
            if (cap[(size_t)b] > 0 && !is_used(b) && (best < 0 || cap[(size_t)b] > cap[(size_t)best])) best = b;
        if (best >= 0) { cap[(size_t)best]--; return best; }
        for (int b : members[(size_t)r])
            if (!is_used(b)) { ++over; return b; }
        ++unplaced;
        return -1;
    };
    std::vector<int> pending;
    std::vector<int> cur((size_t)NJ), row;
    std::vector<std::pair<int, int>> undo_f, undo_l;   // (broker, 1): inflow taken by the row under construction
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
        for (int b : row) used[n_used++] = b;
        if (lead >= 0) used[n_used++] = lead;
        const int over0 = over, unplaced0 = unplaced;
        undo_f.clear(); undo_l.clear();
        bool ok = n_used <= RF;
        for (int r = 0; r < R && ok; ++r) {
            if (lead < 0 && unit(Q(2 * NJ + R + r, p)) >= 1) {
                const int o = over; const int b = take(r, capl);
                if (b >= 0) { lead = b; used[n_used++] = b; if (over == o) undo_l.push_back({b, 1}); }
            }
            for (int n = unit(Q(2 * NJ + r, p)); n > 0; --n) {
                if ((int)row.size() >= RF - 1) break;
                const int o = over; const int b = take(r, capf);
                if (b >= 0) { row.push_back(b); used[n_used++] = b; if (over == o) undo_f.push_back({b, 1}); }
            }
        }
        if (!ok || lead < 0 || (int)row.size() != RF - 1) {   // incomplete: give back what the row took, treat as fractional
            for (auto &u : undo_f) capf[(size_t)u.first] += u.second;
            for (auto &u : undo_l) capl[(size_t)u.first] += u.second;
            over = over0; unplaced = unplaced0;
            pending.push_back(p);
            continue;
        }
        out[(size_t)p * RF] = (uint16_t)lead;
        for (int k = 1; k < RF; ++k) out[(size_t)p * RF + k] = (uint16_t)row[(size_t)k - 1];
    }
    std::vector<std::tuple<int, int, int>> opts;   // (-mass, kind: 0 current replica / 1 rack, index)
    std::vector<int> per_rack((size_t)R);
    for (int p : pending) {
        if (fallback) {
            std::memcpy(out + (size_t)p * RF, fallback + (size_t)p * RF, (size_t)RF * 2);
            ++from_fb;
            continue;
        }
        for (int j = 0; j < NJ; ++j) { const unsigned b = t->current[(size_t)p * NJ + j]; cur[(size_t)j] = (b == KAO_NONE || (int)b >= B) ? -1 : (int)b; }
        n_used = 0;
        std::fill(per_rack.begin(), per_rack.end(), 0);
        opts.clear();
        for (int j = 0; j < NJ; ++j) if (cur[(size_t)j] >= 0) opts.emplace_back(-Q(NJ + j, p), 0, j);
        for (int r = 0; r < R; ++r) opts.emplace_back(-Q(2 * NJ + R + r, p), 1, r);
        std::sort(opts.begin(), opts.end());
        int lead = -1;
        for (auto &o : opts) {
            const int kind = std::get<1>(o), k = std::get<2>(o);
            const int b = kind == 0 ? cur[(size_t)k] : take(k, capl);
            if (b >= 0) { lead = b; used[n_used++] = b; per_rack[t->rack_of[b]]++; break; }
        }
        opts.clear();
        for (int j = 0; j < NJ; ++j) if (cur[(size_t)j] >= 0) opts.emplace_back(-Q(j, p), 0, j);
        for (int r = 0; r < R; ++r) opts.emplace_back(-Q(2 * NJ + r, p), 1, r);
        std::sort(opts.begin(), opts.end());
        row.clear();
        for (int rnd = 0; rnd < 2; ++rnd)
            for (auto &o : opts) {
                if ((int)row.size() >= RF - 1) break;
                const int kind = std::get<1>(o), k = std::get<2>(o);
                int b;
                if (kind == 0) {
                    b = cur[(size_t)k];
                    if (is_used(b) || per_rack[t->rack_of[b]] >= phi) continue;
                } else {
                    if (per_rack[(size_t)k] >= phi) continue;
                    b = take(k, capf);
                    if (b < 0) continue;
                }
                row.push_back(b); used[n_used++] = b; per_rack[t->rack_of[b]]++;
            }
        while ((int)row.size() < RF - 1) {   // (cannot happen on a feasible model: R * phi >= RF)
            int b = 0;
            while (is_used(b)) ++b;
            row.push_back(b); used[n_used++] = b;
        }
        if (lead < 0) { lead = 0; while (is_used(lead)) ++lead; }
        out[(size_t)p * RF] = (uint16_t)lead;
        for (int k = 1; k < RF; ++k) out[(size_t)p * RF + k] = (uint16_t)row[(size_t)k - 1];
    }
    if (rep) { rep[0] = (int32_t)pending.size(); rep[1] = over; rep[2] = unplaced; rep[3] = from_fb; }
    return KAO_OK;
}

}  // namespace kao
