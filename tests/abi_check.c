/* Plain-C consumer of include/kao.h (what a JNI / cgo shim would be): the header must compile as C99 with
 * -pedantic, struct layouts must match what the ctypes binding assumes, and the host-only entry points must work
 * without a GPU.  Built and run by tests/test_host.py::test_header_is_plain_c. */
#include <stdio.h>
#include <string.h>
#include "kao.h"

typedef char assert_topic[(sizeof(kao_topic) == 104) ? 1 : -1];
typedef char assert_opts[(sizeof(kao_opts) == 88) ? 1 : -1];
typedef char assert_result[(sizeof(kao_result) == 72) ? 1 : -1];
typedef char assert_stats[(sizeof(kao_stats) == 80) ? 1 : -1];

int main(void) {
    /* README example, partition 1 only: brokers 0..18, 2 racks, [8,19] with 19 removed */
    uint8_t rack_of[19];
    uint16_t current[2] = {8, KAO_NONE};
    kao_topic t;
    int32_t bd[8];
    int64_t ub = -1;
    int i, rc;
    for (i = 0; i < 19; ++i) rack_of[i] = (uint8_t)(i % 2);
    memset(&t, 0, sizeof t);
    t.n_brokers = 19; t.n_racks = 2; t.n_partitions = 1; t.rf = 2; t.rf_cur = 2;
    t.rack_of = rack_of; t.current = current;
    t.w[0][0] = 4; t.w[0][1] = 1; t.w[1][0] = 2; t.w[1][1] = 2;
    t.rep_lo = t.rep_hi = t.lead_lo = t.lead_hi = t.rack_lo = t.rack_hi = t.prack_lo = t.prack_hi = -1;
    if (kao_version() != KAO_VERSION) return 1;
    rc = kao_derive_bounds(&t, bd);
    if (rc != KAO_OK || bd[0] != 0 || bd[1] != 1 || bd[4] != 1 || bd[5] != 1 || bd[7] != 1) return 2;
    rc = kao_upper_bound(&t, &ub);
    if (rc != KAO_OK || ub != 4) return 3;                 /* the surviving leader keeps its 4 */
    t.rf = 9;
    if (kao_derive_bounds(&t, bd) != KAO_ERR_UNSUPPORTED) return 4;
    if (strlen(kao_strerror(KAO_ERR_UNSUPPORTED)) == 0 || strlen(kao_last_error()) == 0) return 5;
    printf("abi ok\n");
    return 0;
}
