"""Python face of the C ABI (include/kao.h).  Everything here forwards to libkao.so; the numbers
come from the gfx950 kernels."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np

from . import _ffi
from .model import BOUND_KEYS, Topic

STATUS_NAMES = {0: "OPTIMAL_PROVEN", 1: "FEASIBLE_BOUND_GAP", 2: "NO_FEASIBLE", 3: "TIME_LIMIT", 4: "INFEASIBLE_PROVEN"}


class KaoError(RuntimeError):
    def __init__(self, code: int, where: str):
        lib = _ffi.load()
        detail = lib.kao_last_error().decode(errors="replace")
        super().__init__(f"{where}: {lib.kao_strerror(code).decode()} ({code}) {detail}")
        self.code = code


def _check(rc: int, where: str):
    if rc != 0:
        raise KaoError(rc, where)


def library_path() -> str:
    return _ffi.LIB_PATH


def init(device: int = 0):
    _check(_ffi.load().kao_init(int(device)), "kao_init")


def device_name() -> str:
    buf = C.create_string_buffer(256)
    _check(_ffi.load().kao_device_name(buf, 256), "kao_device_name")
    return buf.value.decode()


class _CTopics:
    """C array of kao_topic plus the numpy buffers it points into."""

    def __init__(self, topics: Sequence[Topic]):
        self.keep = []
        self.arr = (_ffi.KaoTopic * len(topics))()
        for i, t in enumerate(topics):
            s = self.arr[i]
            rack = np.ascontiguousarray(t.rack_of, dtype=np.uint8)
            cur = np.ascontiguousarray(t.current, dtype=np.uint16)
            self.keep += [rack, cur]
            s.n_brokers, s.n_racks, s.n_partitions = t.n_brokers, int(t.n_racks), int(t.n_partitions)
            s.rf, s.rf_cur = int(t.rf), int(cur.shape[1])
            s.rack_of = rack.ctypes.data_as(C.POINTER(C.c_uint8))
            s.current = cur.ctypes.data_as(C.POINTER(C.c_uint16))
            for a in range(2):
                for b in range(2):
                    s.w[a][b] = int(t.weights[a][b])
            for k in BOUND_KEYS:
                v = t.bounds_override.get(k, -1) if t.bounds_override else -1
                setattr(s, k, -1 if v is None else int(v))
            for name in ("broker_w", "broker_wl"):
                v = getattr(t, name, None)
                if v is not None:
                    arr = np.ascontiguousarray(v, dtype=np.int32)
                    if arr.shape != (t.n_brokers,):
                        raise ValueError(f"{name}: one weight per broker expected")
                    self.keep.append(arr)
                    setattr(s, name, arr.ctypes.data_as(C.POINTER(C.c_int32)))

    def ptr(self, i: int = 0):
        return C.byref(self.arr[i]) if i else self.arr


def derive_bounds(topic: Topic) -> dict:
    ct = _CTopics([topic])
    out = (C.c_int32 * 8)()
    _check(_ffi.load().kao_derive_bounds(ct.arr, out), "kao_derive_bounds")
    return dict(zip(BOUND_KEYS, [int(v) for v in out]))


def upper_bound(topic: Topic) -> int:
    ct = _CTopics([topic])
    ub = C.c_int64()
    _check(_ffi.load().kao_upper_bound(ct.arr, C.byref(ub)), "kao_upper_bound")
    return int(ub.value)


def check_infeasible(topic: Topic) -> str:
    """'' if no counting argument fails, else the reason the topic is provably infeasible (host only)."""
    ct = _CTopics([topic])
    buf = C.create_string_buffer(256)
    rc = _ffi.load().kao_check_infeasible(ct.arr, buf, 256)
    if rc < 0:
        raise KaoError(rc, "kao_check_infeasible")
    return buf.value.decode() if rc == 1 else ""


def evaluate(topic: Topic, assign) -> tuple:
    """(objective, viol[8]) of one compact candidate, computed by K-eval on the GPU."""
    ct = _CTopics([topic])
    a = np.ascontiguousarray(assign, dtype=np.uint16).reshape(-1)
    if a.size != topic.n_partitions * topic.rf:
        raise ValueError("assignment must have P*rf entries")
    obj = C.c_int64()
    viol = (C.c_int32 * 8)()
    _check(_ffi.load().kao_evaluate(ct.arr, a.ctypes.data_as(C.POINTER(C.c_uint16)), C.byref(obj), viol),
           "kao_evaluate")
    return int(obj.value), np.array(list(viol), dtype=np.int64)


def evaluate_batch(topic: Topic, candidates) -> tuple:
    """candidates [n, P*rf] -> (objective[n] int32, violations[n, 8] int32)."""
    ct = _CTopics([topic])
    c = np.ascontiguousarray(candidates, dtype=np.uint16).reshape(-1, topic.n_partitions * topic.rf)
    n = c.shape[0]
    obj = np.zeros(n, dtype=np.int32)
    viol = np.zeros((n, 8), dtype=np.int32)
    _check(_ffi.load().kao_evaluate_batch(ct.arr, c.ctypes.data_as(C.POINTER(C.c_uint16)), n,
                                          obj.ctypes.data_as(C.POINTER(C.c_int32)),
                                          viol.ctypes.data_as(C.POINTER(C.c_int32))), "kao_evaluate_batch")
    return obj, viol


def canonicalize(topic: Topic, assign) -> np.ndarray:
    ct = _CTopics([topic])
    a = np.ascontiguousarray(assign, dtype=np.uint16).reshape(-1).copy()
    _check(_ffi.load().kao_canonicalize(ct.arr, a.ctypes.data_as(C.POINTER(C.c_uint16))), "kao_canonicalize")
    return a.reshape(topic.n_partitions, topic.rf)


def improve_cycles(topic: Topic, assign, max_rounds: int = 0) -> tuple:
    """KAO-CX (kao_improve_cycles): cyclic-exchange improvement of a FEASIBLE assignment on the GPU.
    Returns (assignment [P, rf], objective, stats dict)."""
    ct = _CTopics([topic])
    a = np.ascontiguousarray(assign, dtype=np.uint16).reshape(-1).copy()
    if a.size != topic.n_partitions * topic.rf:
        raise ValueError("assignment must have P*rf entries")
    obj = C.c_int64()
    st = (C.c_int32 * 8)()
    _check(_ffi.load().kao_improve_cycles(ct.arr, a.ctypes.data_as(C.POINTER(C.c_uint16)), int(max_rounds), C.byref(obj), st),
           "kao_improve_cycles")
    names = ("rounds", "improving_rounds", "realisations", "improving", "candidates", "merged", "objective_before", "objective_after")
    return a.reshape(topic.n_partitions, topic.rf), int(obj.value), dict(zip(names, (int(v) for v in st)))


def cycle_matrices(topic: Topic, assign, layer: int, level: int) -> tuple:
    """Parity hook of KAO-CX: (dist, mid, slot), each [n, n] with n = B + R + 1 (brokers, one slack node per rack, the global slack
    node), of one layer (0 follower moves, 1 role swaps, 2 leader replacements) and level."""
    ct = _CTopics([topic])
    a = np.ascontiguousarray(assign, dtype=np.uint16).reshape(-1)
    n = len(topic.broker_ids) + int(topic.n_racks) + 1
    dist = np.zeros((n, n), dtype=np.int32)
    mid = np.zeros((n, n), dtype=np.int32)
    slot = np.zeros((n, n), dtype=np.uint32)
    _check(_ffi.load().kao_cycle_matrices(ct.arr, a.ctypes.data_as(C.POINTER(C.c_uint16)), layer, level,
                                          dist.ctypes.data_as(C.POINTER(C.c_int32)), mid.ctypes.data_as(C.POINTER(C.c_int32)),
                                          slot.ctypes.data_as(C.POINTER(C.c_uint32))), "kao_cycle_matrices")
    return dist, mid, slot


def cycle_seeds(topic: Topic, assign) -> np.ndarray:
    """Parity hook of KAO-CX: the seed table [P, n_cfg, 2] = (total, completing broker)."""
    ct = _CTopics([topic])
    a = np.ascontiguousarray(assign, dtype=np.uint16).reshape(-1)
    ncfg = C.c_int32()
    _check(_ffi.load().kao_cycle_seeds(ct.arr, a.ctypes.data_as(C.POINTER(C.c_uint16)), None, C.byref(ncfg)), "kao_cycle_seeds")
    tab = np.zeros((topic.n_partitions, ncfg.value, 2), dtype=np.int32)
    _check(_ffi.load().kao_cycle_seeds(ct.arr, a.ctypes.data_as(C.POINTER(C.c_uint16)), tab.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(ncfg)),
           "kao_cycle_seeds")
    return tab


def cycle_pair_edges(topic: Topic, assign, gmin: int = -2):
    """Prototype hook (host only): compound edges of leader-balanced pairs of leader transfers (kao_cycle_pair_edges).
    Returns (cost [B, B] int32 with INT32_MAX = none, dict(half_moves, pairs, edges, closed_improving))."""
    ct = _CTopics([topic])
    a = np.ascontiguousarray(assign, dtype=np.uint16).reshape(-1)
    B = topic.n_brokers
    cost = np.zeros((B, B), dtype=np.int32)
    st = (C.c_int64 * 4)()
    _check(_ffi.load().kao_cycle_pair_edges(ct.arr, a.ctypes.data_as(C.POINTER(C.c_uint16)), int(gmin), cost.ctypes.data_as(C.POINTER(C.c_int32)), st),
           "kao_cycle_pair_edges")
    return cost, dict(half_moves=int(st[0]), pairs=int(st[1]), edges=int(st[2]), closed_improving=int(st[3]))


class EvalPlan:
    """Device-resident K-eval: tables uploaded once; candidates / outputs are device pointers
    (e.g. ``torch.Tensor.data_ptr()``)."""

    def __init__(self, topic: Topic):
        self.topic = topic
        self._ct = _CTopics([topic])
        self._h = C.c_void_p()
        _check(_ffi.load().kao_eval_plan_create(self._ct.arr, C.byref(self._h)), "kao_eval_plan_create")

    def run(self, d_candidates: int, n: int, d_objective: int, d_violations: int = 0, d_best_key: int = 0):
        _check(_ffi.load().kao_eval_plan_run(self._h, C.c_void_p(d_candidates), int(n), C.c_void_p(d_objective),
                                             C.c_void_p(d_violations or None), C.c_void_p(d_best_key or None)),
               "kao_eval_plan_run")

    def sync(self) -> float:
        ms = C.c_double()
        _check(_ffi.load().kao_eval_plan_sync(self._h, C.byref(ms)), "kao_eval_plan_sync")
        return float(ms.value)

    def close(self):
        if self._h:
            _ffi.load().kao_eval_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


@dataclass
class Result:
    status: str
    objective: int
    upper_bound: int
    violations: np.ndarray
    assignment: np.ndarray
    best_restart: int
    seconds_to_best: float


def _make_opts(**kw) -> _ffi.KaoOpts:
    o = _ffi.KaoOpts()
    for k, v in kw.items():
        if not hasattr(o, k):
            raise TypeError(f"unknown option {k}")
        setattr(o, k, v)
    return o


def _results_buffers(topics: Sequence[Topic]):
    res = (_ffi.KaoResult * len(topics))()
    bufs = []
    for i, t in enumerate(topics):
        b = np.full(t.n_partitions * t.rf, 0xFFFF, dtype=np.uint16)
        bufs.append(b)
        res[i].assignment = b.ctypes.data_as(C.POINTER(C.c_uint16))
    return res, bufs


def _unpack(topics, res, bufs) -> List[Result]:
    out = []
    for i, t in enumerate(topics):
        r = res[i]
        out.append(Result(STATUS_NAMES.get(r.status, str(r.status)), int(r.objective), int(r.upper_bound),
                          np.array(list(r.violations), dtype=np.int64), bufs[i].reshape(t.n_partitions, t.rf).copy(),
                          int(r.best_restart), float(r.seconds_to_best)))
    return out


class Session:
    """Resident parallel-restart search over a batch of topics (kao_session_*)."""

    def __init__(self, topics: Sequence[Topic], **opts):
        self.topics = list(topics)
        self._ct = _CTopics(self.topics)
        self._opts = _make_opts(**opts)
        self._h = C.c_void_p()
        _check(_ffi.load().kao_session_create(self._ct.arr, len(self.topics), C.byref(self._opts), C.byref(self._h)),
               "kao_session_create")

    def step(self, n: int = 1):
        lib = _ffi.load()
        for _ in range(n):
            _check(lib.kao_session_step(self._h), "kao_session_step")

    def sync(self):
        _check(_ffi.load().kao_session_sync(self._h), "kao_session_sync")

    def new_generation(self):
        """The next step re-initialises every restart (generation-salted best insertion); old snapshots and best keys are dropped."""
        _check(_ffi.load().kao_session_new_generation(self._h), "kao_session_new_generation")

    def best(self) -> List[Result]:
        res, bufs = _results_buffers(self.topics)
        _check(_ffi.load().kao_session_best(self._h, res), "kao_session_best")
        return _unpack(self.topics, res, bufs)

    def device_keys_ptr(self) -> int:
        """DEVICE address of the packed best keys (uint64[n_topics]); sync() before reading them from another stream."""
        p = C.c_void_p()
        _check(_ffi.load().kao_session_device_keys(self._h, C.byref(p)), "kao_session_device_keys")
        return int(p.value)

    def best_keys(self) -> np.ndarray:
        keys = np.zeros(len(self.topics), dtype=np.uint64)
        _check(_ffi.load().kao_session_best_keys(self._h, keys.ctypes.data_as(C.POINTER(C.c_uint64))),
               "kao_session_best_keys")
        return keys

    def stats(self) -> dict:
        st = _ffi.KaoStats()
        _check(_ffi.load().kao_session_stats(self._h, C.byref(st)), "kao_session_stats")
        return {k: getattr(st, k) for k, _ in _ffi.KaoStats._fields_}

    def restart_state(self, topic: int, restart: int) -> dict:
        t = self.topics[topic]
        n = t.n_partitions * t.rf
        fin = np.zeros(n, dtype=np.uint16)
        best = np.zeros(n, dtype=np.uint16)
        info = (C.c_int32 * 4)()
        _check(_ffi.load().kao_session_restart_state(self._h, topic, restart, fin.ctypes.data_as(C.POINTER(C.c_uint16)),
                                                     best.ctypes.data_as(C.POINTER(C.c_uint16)), info),
               "kao_session_restart_state")
        return dict(final=fin.reshape(t.n_partitions, t.rf), best=best.reshape(t.n_partitions, t.rf),
                    best_obj=int(info[0]), V=int(info[1]), obj=int(info[2]), n_accept=int(info[3]))

    def bound_step(self, targets: Sequence[int], iters: int = 512):
        """One K-bound launch (Lagrangian dual bound) for every topic whose target (incumbent objective) is >= 0."""
        tg = np.ascontiguousarray(targets, dtype=np.int64)
        if tg.shape != (len(self.topics),):
            raise ValueError("one target per topic")
        _check(_ffi.load().kao_session_bound_step(self._h, tg.ctypes.data_as(C.POINTER(C.c_int64)), int(iters)),
               "kao_session_bound_step")

    def set_prices(self, topic: int, a, l, g):
        """Search prices of one topic from the host (fixed point, 65536 = 1): a[n_brokers], l[n_brokers], g[n_racks]."""
        t = self.topics[topic]
        a = np.ascontiguousarray(a, dtype=np.int32); l = np.ascontiguousarray(l, dtype=np.int32); g = np.ascontiguousarray(g, dtype=np.int32)
        if a.shape != (t.n_brokers,) or l.shape != (t.n_brokers,) or g.shape[0] < t.n_racks:
            raise ValueError("a[n_brokers], l[n_brokers], g[n_racks] expected")
        p32 = C.POINTER(C.c_int32)
        _check(_ffi.load().kao_session_set_prices(self._h, topic, a.ctypes.data_as(p32), l.ctypes.data_as(p32), g.ctypes.data_as(p32)),
               "kao_session_set_prices")

    def prices(self, topic: int) -> tuple:
        """(a, l, g): the search prices K-search currently carries for one topic."""
        t = self.topics[topic]
        a = np.zeros(t.n_brokers, dtype=np.int32); l = np.zeros(t.n_brokers, dtype=np.int32); g = np.zeros(t.n_racks, dtype=np.int32)
        p32 = C.POINTER(C.c_int32)
        _check(_ffi.load().kao_session_prices(self._h, topic, a.ctypes.data_as(p32), l.ctypes.data_as(p32), g.ctypes.data_as(p32)),
               "kao_session_prices")
        return a, l, g

    def adopt_prices(self):
        """From the next step on, K-search carries the prices the last finished K-bound launch exported."""
        _check(_ffi.load().kao_session_adopt_prices(self._h), "kao_session_adopt_prices")

    def bounds(self) -> dict:
        """Certificates: upper_bound = min(closed-form bound, floor(best dual value)); flags / iters per topic."""
        n = len(self.topics)
        ub = np.zeros(n, dtype=np.int64)
        fl = np.zeros(n, dtype=np.int32)
        it = np.zeros(n, dtype=np.int32)
        _check(_ffi.load().kao_session_bounds(self._h, ub.ctypes.data_as(C.POINTER(C.c_int64)),
                                              fl.ctypes.data_as(C.POINTER(C.c_int32)), it.ctypes.data_as(C.POINTER(C.c_int32))),
               "kao_session_bounds")
        return dict(upper_bound=ub, flags=fl, iters=it)

    def set_dual_state(self, topic: int, a, l, g):
        """Test hook / KAO-LP: overwrite K-bound's multipliers of one topic (kao_session_set_dual_state)."""
        a, l, g = (np.ascontiguousarray(v, dtype=np.int32) for v in (a, l, g))
        p32 = C.POINTER(C.c_int32)
        _check(_ffi.load().kao_session_set_dual_state(self._h, topic, a.ctypes.data_as(p32), l.ctypes.data_as(p32), g.ctypes.data_as(p32)),
               "kao_session_set_dual_state")

    def dual_state(self, topic: int) -> dict:
        t = self.topics[topic]
        a = np.zeros(t.n_brokers, dtype=np.int32)
        l = np.zeros(t.n_brokers, dtype=np.int32)
        g = np.zeros(t.n_racks, dtype=np.int32)
        best = C.c_int64()
        p32 = C.POINTER(C.c_int32)
        _check(_ffi.load().kao_session_dual_state(self._h, topic, a.ctypes.data_as(p32), l.ctypes.data_as(p32),
                                                  g.ctypes.data_as(p32), C.byref(best)), "kao_session_dual_state")
        return dict(a=a, l=l, g=g, best_dual=int(best.value))

    def close(self):
        if self._h:
            _ffi.load().kao_session_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def dual_bound(topic: Topic, target: int, iters: int = 512, launches: int = 1) -> dict:
    """One-shot K-bound (kao_dual_bound): Lagrangian dual certificate of one topic towards the incumbent `target`."""
    ct = _CTopics([topic])
    bound, best = C.c_int64(), C.c_int64()
    itn, fl = C.c_int32(), C.c_int32()
    mult = np.zeros(2 * topic.n_brokers + topic.n_racks, dtype=np.int32)
    _check(_ffi.load().kao_dual_bound(ct.ptr(0), int(target), int(iters), int(launches), C.byref(bound), C.byref(best),
                                      C.byref(itn), C.byref(fl), mult.ctypes.data_as(C.POINTER(C.c_int32))), "kao_dual_bound")
    B = topic.n_brokers
    return dict(bound=int(bound.value), best_dual=int(best.value), iters=int(itn.value), flags=int(fl.value),
                a=mult[:B].copy(), l=mult[B:2 * B].copy(), g=mult[2 * B:].copy())


def lp_bound(topic: Topic, tol: float = 0.0, max_iters: int = 0) -> dict:
    """KAO-LP (kao_lp_bound): interior-point solve of the compact LP relaxation on the device, its row duals as multipliers,
    K-bound's exact dual value there.  bound = floor(best_dual / 65536) is a valid upper bound on the optimum."""
    ct = _CTopics([topic])
    bound, best = C.c_int64(), C.c_int64()
    B = topic.n_brokers
    mult = np.zeros(2 * B + topic.n_racks, dtype=np.int32)
    st = np.zeros(8)
    _check(_ffi.load().kao_lp_bound(ct.ptr(0), float(tol), int(max_iters), C.byref(bound), C.byref(best),
                                    mult.ctypes.data_as(C.POINTER(C.c_int32)), st.ctypes.data_as(C.POINTER(C.c_double))), "kao_lp_bound")
    return dict(bound=int(bound.value), best_dual=int(best.value), a=mult[:B].copy(), l=mult[B:2 * B].copy(), g=mult[2 * B:].copy(),
                iterations=int(st[0]), primal=float(st[1]), dual=float(st[2]), status=int(st[3]), mu=float(st[4]), pinf=float(st[5]),
                dinf=float(st[6]), ms=float(st[7]))


def lp_round(topic: Topic, pert: float = 0.0, salt: int = 0, tol: float = 0.0, max_iters: int = 0, fallback=None) -> dict:
    """KAO-LP's primal side (kao_lp_round): the interior-point iterate of the LP with perturbed costs, rounded into an assignment and
    evaluated exactly.  `fallback` ([P, rf] dense indices): the rows partitions with fractional variables keep."""
    ct = _CTopics([topic])
    n = topic.n_partitions * topic.rf
    a = np.zeros(n, dtype=np.uint16) if fallback is None else np.ascontiguousarray(np.asarray(fallback).reshape(-1), dtype=np.uint16).copy()
    if a.shape != (n,):
        raise ValueError("fallback: [P, rf] expected")
    obj = C.c_int64()
    viol = (C.c_int32 * 8)()
    st = np.zeros(8)
    _check(_ffi.load().kao_lp_round(ct.ptr(0), float(pert), int(salt) & 0xFFFFFFFF, float(tol), int(max_iters), 0 if fallback is None else 1,
                                    a.ctypes.data_as(C.POINTER(C.c_uint16)), C.byref(obj), viol, st.ctypes.data_as(C.POINTER(C.c_double))), "kao_lp_round")
    return dict(assignment=a.reshape(topic.n_partitions, topic.rf), objective=int(obj.value), violations=[int(v) for v in viol],
                iterations=int(st[0]), status=int(st[1]), fractional=int(st[2]), over_inflow=int(st[3]), from_fallback=int(st[4]),
                ms_lp=float(st[5]), ms_round=float(st[6]), pert=float(st[7]))


def lp_round_host(topic: Topic, q, zq, fallback=None) -> dict:
    """Test hook (kao_lp_round_host): the host half of lp_round from a quantised iterate q [(2 rf_cur + 2 R), P] uint8, zq [2 B] int32."""
    ct = _CTopics([topic])
    n = topic.n_partitions * topic.rf
    q = np.ascontiguousarray(q, dtype=np.uint8); zq = np.ascontiguousarray(zq, dtype=np.int32)
    a = np.zeros(n, dtype=np.uint16) if fallback is None else np.ascontiguousarray(np.asarray(fallback).reshape(-1), dtype=np.uint16).copy()
    rep = (C.c_int32 * 4)()
    _check(_ffi.load().kao_lp_round_host(ct.ptr(0), q.ctypes.data_as(C.POINTER(C.c_uint8)), zq.ctypes.data_as(C.POINTER(C.c_int32)),
                                         0 if fallback is None else 1, a.ctypes.data_as(C.POINTER(C.c_uint16)), rep), "kao_lp_round_host")
    return dict(assignment=a.reshape(topic.n_partitions, topic.rf), fractional=int(rep[0]), over_inflow=int(rep[1]), unplaced=int(rep[2]), from_fallback=int(rep[3]))


def lp_repair_host(topic: Topic, assignment) -> np.ndarray:
    """Test hook (kao_lp_round_host, mode 2): the band repair of the rounding alone, on a complete assignment [P, rf]."""
    ct = _CTopics([topic])
    a = np.ascontiguousarray(np.asarray(assignment).reshape(-1), dtype=np.uint16).copy()
    rep = (C.c_int32 * 4)()
    _check(_ffi.load().kao_lp_round_host(ct.ptr(0), None, None, 2, a.ctypes.data_as(C.POINTER(C.c_uint16)), rep), "kao_lp_round_host")
    return a.reshape(topic.n_partitions, topic.rf)


def lp_trace(topic: Topic, tol: float = 0.0, max_iters: int = 80) -> dict:
    """Test hook (kao_lp_trace): the interior-point solve alone and its per-iterate trace (mu, primal, dual, pinf, dinf)."""
    ct = _CTopics([topic])
    B = topic.n_brokers
    mult = np.zeros(2 * B + topic.n_racks, dtype=np.int32)
    st = np.zeros(8)
    max_iters = int(max_iters) if int(max_iters) > 0 else 80      # the C side reads <= 0 as 80: size the trace for what it will write
    tr = np.zeros(5 * (max_iters + 2))
    pd = C.POINTER(C.c_double)
    _check(_ffi.load().kao_lp_trace(ct.ptr(0), float(tol), int(max_iters), tr.ctypes.data_as(pd), st.ctypes.data_as(pd),
                                    mult.ctypes.data_as(C.POINTER(C.c_int32))), "kao_lp_trace")
    it = int(st[0])
    return dict(a=mult[:B].copy(), l=mult[B:2 * B].copy(), g=mult[2 * B:].copy(), iterations=it, primal=float(st[1]), dual=float(st[2]),
                status=int(st[3]), ms=float(st[7]), trace=tr[:5 * (it + 1)].reshape(-1, 5))


def lp_sharded(topic: Topic, devices: Sequence[int], pert: float = 0.0, salt: int = 0, tol: float = 0.0, max_iters: int = 0) -> dict:
    """Test hook (kao_lp_sharded_test): one topic's LP solved by len(devices) shards of its partitions (repeated devices = logical shards,
    KAO_RCCL_LOOPBACK=1); certificate + rounded iterate as lp_bound / lp_round."""
    ct = _CTopics([topic])
    dv = np.asarray(list(devices), dtype=np.int32)
    A = np.full((topic.n_partitions, topic.rf), 0xFFFF, dtype=np.uint16)
    bound = C.c_int64(0); obj = C.c_int64(0)
    viol = np.zeros(8, dtype=np.int32); st = np.zeros(8)
    _check(_ffi.load().kao_lp_sharded_test(ct.ptr(0), dv.ctypes.data_as(C.POINTER(C.c_int32)), len(dv), float(pert), int(salt), float(tol), int(max_iters),
                                           C.byref(bound), A.ctypes.data_as(C.POINTER(C.c_uint16)), C.byref(obj), viol.ctypes.data_as(C.POINTER(C.c_int32)),
                                           st.ctypes.data_as(C.POINTER(C.c_double))), "kao_lp_sharded_test")
    return dict(bound=int(bound.value), assignment=A, objective=int(obj.value), violations=viol.tolist(), iterations=int(st[0]), status=int(st[1]),
                fractional=int(st[2]), collectives=int(st[3]), dual=float(st[4]), ms_lp=float(st[5]), ms_rest=float(st[6]), pert=float(st[7]))


def dense_spd_test(A: np.ndarray, rhs: Optional[np.ndarray] = None) -> dict:
    """Test hook (kao_dense_spd_test): KAO-LP's dense kernels (kao_chol.hip) alone on a symmetric positive definite matrix whose order
    is a multiple of 64: the factor (L below, L^T tile-wise above), the inverses of L's diagonal tiles, the solution of A x = rhs."""
    A = np.ascontiguousarray(A, dtype=np.float64)
    n = A.shape[0]
    assert A.shape == (n, n)
    f = np.zeros((n, n)); li = np.zeros((n // 64, 64, 64)); x = np.zeros(n); ms = np.zeros(2)
    pd = C.POINTER(C.c_double)
    r = None if rhs is None else np.ascontiguousarray(rhs, dtype=np.float64)
    _check(_ffi.load().kao_dense_spd_test(A.ctypes.data_as(pd), n, None if r is None else r.ctypes.data_as(pd), f.ctypes.data_as(pd),
                                          li.ctypes.data_as(pd), x.ctypes.data_as(pd), ms.ctypes.data_as(pd)), "kao_dense_spd_test")
    return dict(factor=f, linv=li, x=x, ms_factor=float(ms[0]), ms_solve=float(ms[1]))


def decode_key(key: int) -> tuple:
    """packed best key -> (violation, objective, restart)"""
    key = int(key)
    return key >> 44, 0xFFFFFF - ((key >> 20) & 0xFFFFFF), key & 0xFFFFF


def solve(topics: Sequence[Topic], target_objective: Optional[Sequence[int]] = None, **opts) -> List[Result]:
    """Whole job (kao_solve): search every topic until it is proven optimal (or `target_objective[i]` is reached) or the
    time limit / `max_launches` runs out.  `stop_at_bound` defaults to 1 here (return as soon as every topic is proven);
    pass stop_at_bound=0 to keep searching until the limit.

    Result.status: "OPTIMAL_PROVEN" (objective == upper_bound: the answer lp_solve would give, README.md:135-136),
    "FEASIBLE_BOUND_GAP" / "TIME_LIMIT" (feasible plan, possibly suboptimal: upper_bound - objective is the certified
    gap), "NO_FEASIBLE" (no plan found; not a proof), "INFEASIBLE_PROVEN" (counting argument: lp_solve's "This problem
    is infeasible")."""
    topics = list(topics)
    ct = _CTopics(topics)
    opts.setdefault("stop_at_bound", 1)
    o = _make_opts(**opts)
    if target_objective is not None:
        tgt = (C.c_int64 * len(topics))(*[int(v) for v in target_objective])
        o.target_objective = tgt
    res, bufs = _results_buffers(topics)
    _check(_ffi.load().kao_solve(ct.arr, len(topics), C.byref(o), res), "kao_solve")
    return _unpack(topics, res, bufs)


def solve_multi(topics: Sequence[Topic], devices: Sequence[int], target_objective: Optional[Sequence[int]] = None, **opts) -> List[Result]:
    """Whole job on several GPUs of this process (kao_solve_multi): topics are dealt to the devices; with fewer topics
    than devices every device searches every topic and the elites are exchanged over RCCL (min-allreduce of the packed
    best keys + broadcast of the winner)."""
    topics = list(topics)
    ct = _CTopics(topics)
    opts.setdefault("stop_at_bound", 1)
    o = _make_opts(**opts)
    if target_objective is not None:
        tgt = (C.c_int64 * len(topics))(*[int(v) for v in target_objective])
        o.target_objective = tgt
    dev = (C.c_int32 * len(devices))(*[int(d) for d in devices])
    res, bufs = _results_buffers(topics)
    _check(_ffi.load().kao_solve_multi(ct.arr, len(topics), dev, len(devices), C.byref(o), res), "kao_solve_multi")
    return _unpack(topics, res, bufs)


def solve_capped(topics: Sequence[Topic], replica_cap: Sequence[int], devices: Optional[Sequence[int]] = None, max_rounds: int = 0, **opts):
    """Cluster-wide per-broker load caps (kao_solve_capped): sum over all topics of the replicas on broker b <= replica_cap[b]
    (-1 = none).  Returns (results, lagrangian_bound or None)."""
    topics = list(topics)
    ct = _CTopics(topics)
    o = _make_opts(**opts)
    cap = np.ascontiguousarray(replica_cap, dtype=np.int32)
    if cap.shape != (topics[0].n_brokers,):
        raise ValueError("one cap per broker expected")
    dev = (C.c_int32 * len(devices))(*[int(d) for d in devices]) if devices else None
    res, bufs = _results_buffers(topics)
    lb = C.c_int64()
    _check(_ffi.load().kao_solve_capped(ct.arr, len(topics), cap.ctypes.data_as(C.POINTER(C.c_int32)), dev, len(devices) if devices else 0,
                                        C.byref(o), int(max_rounds), res, C.byref(lb)), "kao_solve_capped")
    out = _unpack(topics, res, bufs)
    return out, (None if lb.value == (1 << 63) - 1 else int(lb.value))


def rccl_selftest(devices: Sequence[int]) -> None:
    """Runs the collectives of kao_solve_multi (min-allreduce of uint64 keys, broadcast) on the listed devices; raises on failure."""
    dev = (C.c_int32 * len(devices))(*[int(d) for d in devices])
    _check(_ffi.load().kao_rccl_selftest(dev, len(devices)), "kao_rccl_selftest")


def rccl_loopback_counts() -> tuple:
    """(all-reduces, broadcasts) completed by the in-process loop-back collectives (test hook KAO_RCCL_LOOPBACK=1)."""
    out = (C.c_uint64 * 2)()
    _check(_ffi.load().kao_rccl_loopback_counts(out), "kao_rccl_loopback_counts")
    return int(out[0]), int(out[1])


def last_solve_timing() -> dict:
    """C-side wall-clock breakdown of the last kao_solve (seconds from its entry)."""
    out = (C.c_double * 16)()
    _check(_ffi.load().kao_last_solve_timing(out), "kao_last_solve_timing")
    return dict(session_ready=out[0], time_to_best=out[1], results_read_back=out[2], returned=out[3], launches=int(out[4]),
                delta_candidates=int(out[5]), bound_launches=int(out[6]), elite_exchanges=int(out[7]), bound_iters=int(out[8]),
                cx_calls=int(out[9]), cx_gains=int(out[10]), search_iters=int(out[11]), generations=int(out[12]), cx_further_starts=int(out[13]),
                lp_solves=int(out[14]), lp_iters=int(out[15]))


def last_solve_lp() -> dict:
    """KAO-LP in this thread's last solve (kao_last_solve_lp)."""
    out = (C.c_double * 8)()
    _check(_ffi.load().kao_last_solve_lp(out), "kao_last_solve_lp")
    return dict(solves=int(out[0]), iterations=int(out[1]), rounded=int(out[2]), adopted=int(out[3]), fractional_partitions=int(out[4]))


def last_solve_profile() -> dict:
    """K-search as the last kao_solve ran it (kao_last_solve_profile; needs profile=1 in that solve's options)."""
    out = (C.c_double * 8)()
    _check(_ffi.load().kao_last_solve_profile(out), "kao_last_solve_profile")
    return dict(ms_search=out[0], ms_eval=out[1], search_launches=int(out[2]), restarts=int(out[3]), search_bytes_algo=int(out[4]),
                delta_candidates=int(out[5]), lds_bytes_search=int(out[6]), blocks_search=int(out[7]))
