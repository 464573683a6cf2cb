"""ctypes view of include/kao.h.  Loading fails loudly if libkao.so has not been built."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libkao.so")


class KaoTopic(C.Structure):
    _fields_ = [("n_brokers", C.c_int32), ("n_racks", C.c_int32), ("n_partitions", C.c_int32),
                ("rf", C.c_int32), ("rf_cur", C.c_int32),
                ("rack_of", C.POINTER(C.c_uint8)), ("current", C.POINTER(C.c_uint16)),
                ("w", (C.c_int32 * 2) * 2),
                ("rep_lo", C.c_int32), ("rep_hi", C.c_int32), ("lead_lo", C.c_int32), ("lead_hi", C.c_int32),
                ("rack_lo", C.c_int32), ("rack_hi", C.c_int32), ("prack_lo", C.c_int32), ("prack_hi", C.c_int32),
                ("broker_w", C.POINTER(C.c_int32)), ("broker_wl", C.POINTER(C.c_int32))]


class KaoOpts(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("time_limit_s", C.c_double), ("restarts", C.c_int32),
                ("iters_per_launch", C.c_int32), ("max_launches", C.c_int32), ("obj_scale", C.c_int32),
                ("lam_min", C.c_int32), ("lam_max", C.c_int32), ("period_log2", C.c_int32),
                ("stop_at_bound", C.c_int32), ("profile", C.c_int32), ("dual_iters", C.c_int32),
                ("elite_period", C.c_int32), ("use_prices", C.c_int32), ("use_cycles", C.c_int32), ("islands", C.c_int32),
                ("schedule", C.c_int32), ("team", C.c_int32),
                ("target_objective", C.POINTER(C.c_int64))]


class KaoResult(C.Structure):
    _fields_ = [("status", C.c_int32), ("best_restart", C.c_int32), ("objective", C.c_int64),
                ("upper_bound", C.c_int64), ("violations", C.c_int32 * 8), ("seconds_to_best", C.c_double),
                ("assignment", C.POINTER(C.c_uint16))]


class KaoStats(C.Structure):
    _fields_ = [("launches", C.c_uint64), ("delta_candidates", C.c_uint64), ("full_candidates", C.c_uint64),
                ("ms_search", C.c_double), ("ms_eval", C.c_double), ("search_bytes_algo", C.c_uint64),
                ("eval_bytes_algo", C.c_uint64), ("n_restarts_total", C.c_int32), ("lds_bytes_search", C.c_int32),
                ("blocks_search", C.c_int32), ("drift", C.c_int32), ("launch_groups", C.c_int32), ("reserved", C.c_int32)]


# every symbol include/kao.h declares: name -> (restype, argtypes)
_P = C.POINTER
SIGNATURES = {
    "kao_init": (C.c_int, [C.c_int]),
    "kao_shutdown": (None, []),
    "kao_version": (C.c_int, []),
    "kao_strerror": (C.c_char_p, [C.c_int]),
    "kao_last_error": (C.c_char_p, []),
    "kao_device_name": (C.c_int, [C.c_char_p, C.c_int]),
    "kao_derive_bounds": (C.c_int, [_P(KaoTopic), _P(C.c_int32)]),
    "kao_upper_bound": (C.c_int, [_P(KaoTopic), _P(C.c_int64)]),
    "kao_canonicalize": (C.c_int, [_P(KaoTopic), _P(C.c_uint16)]),
    "kao_check_infeasible": (C.c_int, [_P(KaoTopic), C.c_char_p, C.c_int]),
    "kao_evaluate": (C.c_int, [_P(KaoTopic), _P(C.c_uint16), _P(C.c_int64), _P(C.c_int32)]),
    "kao_evaluate_batch": (C.c_int, [_P(KaoTopic), _P(C.c_uint16), C.c_int64, _P(C.c_int32), _P(C.c_int32)]),
    "kao_eval_plan_create": (C.c_int, [_P(KaoTopic), _P(C.c_void_p)]),
    "kao_eval_plan_run": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "kao_eval_plan_sync": (C.c_int, [C.c_void_p, _P(C.c_double)]),
    "kao_eval_plan_destroy": (None, [C.c_void_p]),
    "kao_session_create": (C.c_int, [_P(KaoTopic), C.c_int32, _P(KaoOpts), _P(C.c_void_p)]),
    "kao_session_step": (C.c_int, [C.c_void_p]),
    "kao_session_sync": (C.c_int, [C.c_void_p]),
    "kao_session_new_generation": (C.c_int, [C.c_void_p]),
    "kao_session_best": (C.c_int, [C.c_void_p, _P(KaoResult)]),
    "kao_session_best_keys": (C.c_int, [C.c_void_p, _P(C.c_uint64)]),
    "kao_session_device_keys": (C.c_int, [C.c_void_p, _P(C.c_void_p)]),
    "kao_session_stats": (C.c_int, [C.c_void_p, _P(KaoStats)]),
    "kao_session_restart_state": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, _P(C.c_uint16), _P(C.c_uint16),
                                            _P(C.c_int32)]),
    "kao_session_bound_step": (C.c_int, [C.c_void_p, _P(C.c_int64), C.c_int32]),
    "kao_session_set_prices": (C.c_int, [C.c_void_p, C.c_int32, _P(C.c_int32), _P(C.c_int32), _P(C.c_int32)]),
    "kao_session_adopt_prices": (C.c_int, [C.c_void_p]),
    "kao_session_prices": (C.c_int, [C.c_void_p, C.c_int32, _P(C.c_int32), _P(C.c_int32), _P(C.c_int32)]),
    "kao_session_bound_busy": (C.c_int, [C.c_void_p]),
    "kao_session_bounds": (C.c_int, [C.c_void_p, _P(C.c_int64), _P(C.c_int32), _P(C.c_int32)]),
    "kao_session_dual_state": (C.c_int, [C.c_void_p, C.c_int32, _P(C.c_int32), _P(C.c_int32), _P(C.c_int32), _P(C.c_int64)]),
    "kao_session_set_dual_state": (C.c_int, [C.c_void_p, C.c_int32, _P(C.c_int32), _P(C.c_int32), _P(C.c_int32)]),
    "kao_lp_bound": (C.c_int, [_P(KaoTopic), C.c_double, C.c_int32, _P(C.c_int64), _P(C.c_int64), _P(C.c_int32), _P(C.c_double)]),
    "kao_lp_round": (C.c_int, [_P(KaoTopic), C.c_double, C.c_uint32, C.c_double, C.c_int32, C.c_int32, _P(C.c_uint16), _P(C.c_int64), _P(C.c_int32), _P(C.c_double)]),
    "kao_lp_round_host": (C.c_int, [_P(KaoTopic), _P(C.c_uint8), _P(C.c_int32), C.c_int32, _P(C.c_uint16), _P(C.c_int32)]),
    "kao_dense_spd_test": (C.c_int, [_P(C.c_double), C.c_int32, _P(C.c_double), _P(C.c_double), _P(C.c_double), _P(C.c_double), _P(C.c_double)]),
    "kao_lp_sharded_test": (C.c_int, [_P(KaoTopic), _P(C.c_int32), C.c_int32, C.c_double, C.c_uint32, C.c_double, C.c_int32, _P(C.c_int64), _P(C.c_uint16),
                                      _P(C.c_int64), _P(C.c_int32), _P(C.c_double)]),
    "kao_lp_trace": (C.c_int, [_P(KaoTopic), C.c_double, C.c_int32, _P(C.c_double), _P(C.c_double), _P(C.c_int32)]),
    "kao_dual_bound": (C.c_int, [_P(KaoTopic), C.c_int64, C.c_int32, C.c_int32, _P(C.c_int64), _P(C.c_int64), _P(C.c_int32),
                                 _P(C.c_int32), _P(C.c_int32)]),
    "kao_session_destroy": (None, [C.c_void_p]),
    "kao_solve": (C.c_int, [_P(KaoTopic), C.c_int32, _P(KaoOpts), _P(KaoResult)]),
    "kao_solve_multi": (C.c_int, [_P(KaoTopic), C.c_int32, _P(C.c_int32), C.c_int32, _P(KaoOpts), _P(KaoResult)]),
    "kao_solve_capped": (C.c_int, [_P(KaoTopic), C.c_int32, _P(C.c_int32), _P(C.c_int32), C.c_int32, _P(KaoOpts), C.c_int32,
                                  _P(KaoResult), _P(C.c_int64)]),
    "kao_rccl_selftest": (C.c_int, [_P(C.c_int32), C.c_int32]),
    "kao_rccl_loopback_counts": (C.c_int, [_P(C.c_uint64)]),
    "kao_improve_cycles": (C.c_int, [_P(KaoTopic), _P(C.c_uint16), C.c_int32, _P(C.c_int64), _P(C.c_int32)]),
    "kao_cycle_matrices": (C.c_int, [_P(KaoTopic), _P(C.c_uint16), C.c_int32, C.c_int32, _P(C.c_int32), _P(C.c_int32), _P(C.c_uint32)]),
    "kao_cycle_seeds": (C.c_int, [_P(KaoTopic), _P(C.c_uint16), _P(C.c_int32), _P(C.c_int32)]),
    "kao_cycle_pair_edges": (C.c_int, [_P(KaoTopic), _P(C.c_uint16), C.c_int32, _P(C.c_int32), _P(C.c_int64)]),
    "kao_last_solve_timing": (C.c_int, [_P(C.c_double)]),
    "kao_last_solve_profile": (C.c_int, [_P(C.c_double)]),
    "kao_last_solve_lp": (C.c_int, [_P(C.c_double)]),
}

_lib = None


def load():
    """dlopen libkao.so and bind every declared entry point.  No fallback: a missing library is
    an error the caller must see (build it with ``python -c 'import __graft_entry__ as g; g.build()'``
    or ``make -C kafka_assignment_optimizer_amd/csrc``)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OSError(f"{LIB_PATH} not built: the HIP library is required (no CPU fallback); "
                          "run `make -C kafka_assignment_optimizer_amd/csrc`")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib
