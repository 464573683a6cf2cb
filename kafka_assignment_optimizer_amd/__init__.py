"""kafka_assignment_optimizer_amd -- MI355X (gfx950) solver for the Kafka partition-assignment
0-1 model of killerwhile/kafka-assignment-optimizer (reference: /root/reference/README.md).

The package is a thin host layer over the C ABI of ``libkao.so`` (include/kao.h): reassignment
JSON in/out (README.md:52-63, README.md:67-78), instance preparation, and ctypes bindings.  All
evaluation and search runs in the hand-written HIP kernels (csrc/kao_kernels.hip); there is no CPU
fallback -- importing works anywhere, computing needs the GPU and the built library.
"""
from .model import DEFAULT_WEIGHTS, NONE, Topic, assignment_to_json, topics_from_json  # noqa: F401
from .solver import (EvalPlan, KaoError, Result, Session, canonicalize, check_infeasible, derive_bounds, device_name, dual_bound, evaluate,  # noqa: F401
                     cycle_matrices, cycle_pair_edges, cycle_seeds, dense_spd_test, evaluate_batch, improve_cycles, init, last_solve_lp, last_solve_profile, last_solve_timing, library_path, lp_bound, lp_repair_host, lp_round, lp_round_host, lp_sharded, lp_trace, rccl_loopback_counts, rccl_selftest, solve, solve_capped, solve_multi, upper_bound)

__version__ = "0.1.0"
