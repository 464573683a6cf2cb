#include <vector>
#include "kao_host.h"
#include "kao_internal.h"
void derive_bounds(const kao_topic *t, int32_t o[8]) { kao_derive_bounds(t, o); }
extern "C" int round_asan(const kao_topic *t, const uint8_t *q, const int32_t *zq, int32_t use_fallback, uint16_t *assignment, int32_t rep[4]) {
    if (use_fallback == 2) return lp_round_assignment(t, nullptr, zq, nullptr, assignment, rep);
    std::vector<uint16_t> fb;
    if (use_fallback) fb.assign(assignment, assignment + (size_t)t->n_partitions * t->rf);
    return lp_round_assignment(t, q, zq, use_fallback ? fb.data() : nullptr, assignment, rep);
}
