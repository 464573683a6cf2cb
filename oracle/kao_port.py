"""ctypes wrapper of oracle/libkao_port.so (CPU restatement in C) -- TEST INFRASTRUCTURE ONLY.
See oracle/kao_port.c.  Never imported by the product package."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class PortTopic(C.Structure):
    _fields_ = [("n_brokers", C.c_int32), ("n_racks", C.c_int32), ("n_partitions", C.c_int32),
                ("rf", C.c_int32), ("rf_cur", C.c_int32),
                ("rack_of", C.POINTER(C.c_uint8)), ("current", C.POINTER(C.c_uint16)),
                ("w", (C.c_int32 * 2) * 2),
                ("rep_lo", C.c_int32), ("rep_hi", C.c_int32), ("lead_lo", C.c_int32), ("lead_hi", C.c_int32),
                ("rack_lo", C.c_int32), ("rack_hi", C.c_int32), ("prack_lo", C.c_int32), ("prack_hi", C.c_int32),
                ("broker_w", C.POINTER(C.c_int32)), ("broker_wl", C.POINTER(C.c_int32))]


class PortParams(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("obj_scale", C.c_int32), ("lam_min", C.c_int32),
                ("lam_max", C.c_int32), ("period_log2", C.c_int32), ("team", C.c_int32)]


class PortExtra(C.Structure):
    _fields_ = [("pa", C.POINTER(C.c_int32)), ("pl", C.POINTER(C.c_int32)), ("pg", C.POINTER(C.c_int32)),
                ("elite", C.POINTER(C.c_uint16)), ("elite_obj", C.c_int32), ("elite_rho", C.c_int32), ("gen", C.c_int32)]


def build() -> str:
    so = os.path.join(_HERE, "libkao_port.so")
    src = os.path.join(_HERE, "kao_port.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libkao_port.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.kao_port_eval.argtypes = [C.POINTER(PortTopic), C.POINTER(C.c_uint16), C.POINTER(C.c_int64),
                                       C.POINTER(C.c_int32)]
        _LIB.kao_port_eval.restype = C.c_int
        _LIB.kao_port_ls_create.argtypes = [C.POINTER(PortTopic)]
        _LIB.kao_port_ls_create.restype = C.c_void_p
        _LIB.kao_port_ls_destroy.argtypes = [C.c_void_p]
        _LIB.kao_port_search.argtypes = [C.c_void_p, C.POINTER(PortParams), C.c_uint32, C.c_uint32, C.c_uint32,
                                         C.POINTER(C.c_uint16), C.POINTER(C.c_uint16), C.POINTER(C.c_int64)]
        _LIB.kao_port_search.restype = C.c_int
        _LIB.kao_port_run_create.argtypes = [C.c_void_p, C.POINTER(PortParams), C.c_uint32]
        _LIB.kao_port_run_create.restype = C.c_void_p
        _LIB.kao_port_run_destroy.argtypes = [C.c_void_p]
        _LIB.kao_port_run_launch.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(PortExtra), C.c_int32]
        _LIB.kao_port_run_launch.restype = C.c_int
        _LIB.kao_port_run_read.argtypes = [C.c_void_p, C.POINTER(C.c_uint16), C.POINTER(C.c_uint16), C.POINTER(C.c_int64)]
        _LIB.kao_port_run_read.restype = C.c_int
        _LIB.kao_port_valid_fraction.argtypes = [C.c_void_p, C.POINTER(PortParams), C.c_uint32, C.c_uint32, C.c_uint32]
        _LIB.kao_port_valid_fraction.restype = C.c_double
        _LIB.kao_port_search_many.argtypes = [C.c_void_p, C.POINTER(PortParams)] + [C.c_uint32] * 5
        _LIB.kao_port_search_many.restype = C.c_uint64
        _LIB.kao_port_dual_partition.argtypes = [C.POINTER(PortTopic), C.c_int] + [C.POINTER(C.c_int32)] * 6
        _LIB.kao_port_dual_partition.restype = C.c_int
        _LIB.kao_port_dual_bound.argtypes = [C.POINTER(PortTopic), C.c_int64, C.c_int32] + [C.POINTER(C.c_int32)] * 6 + [
            C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int32)]
        _LIB.kao_port_dual_bound.restype = C.c_int
        _LIB.kao_port_dual_bound_rec.argtypes = _LIB.kao_port_dual_bound.argtypes + [C.POINTER(C.c_int32)] * 3
        _LIB.kao_port_dual_bound_rec.restype = C.c_int
    return _LIB


class CTopic:
    """Keeps the numpy buffers alive next to the C struct."""

    def __init__(self, topic):
        bd = topic.bounds()
        self.rack = np.ascontiguousarray(topic.rack_of, dtype=np.uint8)
        self.cur = np.ascontiguousarray(topic.current, dtype=np.uint16)
        s = PortTopic()
        s.n_brokers, s.n_racks, s.n_partitions = topic.n_brokers, topic.n_racks, topic.n_partitions
        s.rf, s.rf_cur = topic.rf, topic.rf_cur
        s.rack_of = self.rack.ctypes.data_as(C.POINTER(C.c_uint8))
        s.current = self.cur.ctypes.data_as(C.POINTER(C.c_uint16))
        for i in range(2):
            for j in range(2):
                s.w[i][j] = int(topic.weights[i][j])
        for k in ("rep_lo", "rep_hi", "lead_lo", "lead_hi", "rack_lo", "rack_hi", "prack_lo", "prack_hi"):
            setattr(s, k, bd[k])
        self.bw = None if getattr(topic, "broker_w", None) is None else np.ascontiguousarray(topic.broker_w, dtype=np.int32)
        self.bwl = None if getattr(topic, "broker_wl", None) is None else np.ascontiguousarray(topic.broker_wl, dtype=np.int32)
        if self.bw is not None:
            s.broker_w = self.bw.ctypes.data_as(C.POINTER(C.c_int32))
        if self.bwl is not None:
            s.broker_wl = self.bwl.ctypes.data_as(C.POINTER(C.c_int32))
        self.s = s
        self.topic = topic


def port_eval(topic, assign) -> Tuple[int, np.ndarray]:
    ct = CTopic(topic)
    a = np.ascontiguousarray(assign, dtype=np.uint16).reshape(-1)
    obj = C.c_int64()
    viol = (C.c_int32 * 8)()
    rc = lib().kao_port_eval(C.byref(ct.s), a.ctypes.data_as(C.POINTER(C.c_uint16)), C.byref(obj), viol)
    assert rc == 0
    return int(obj.value), np.array(list(viol), dtype=np.int64)


DEFAULT_PARAMS = dict(obj_scale=4, lam_min=1, lam_max=40, period_log2=None, team=1)  # period None = by topic size, see below; team = wavefronts per restart (k_team)


def auto_period_log2(topic) -> int:
    """Sawtooth period chosen by topic size: floor(log2(2 * P * RF)) clamped to 8..16."""
    return min(16, max(8, (2 * topic.n_partitions * topic.rf).bit_length() - 1))


def port_search(topic, seed: int, rho: int, launches: int, iters: int, **params):
    """Replay restart `rho`.  Returns dict(final, best, best_obj, V, obj, n_eval, n_accept)."""
    ct = CTopic(topic)
    h = lib().kao_port_ls_create(C.byref(ct.s))
    if not h:
        raise ValueError("unsupported instance (RF > 8 or racks > 255)")
    try:
        pr = dict(DEFAULT_PARAMS)
        pr.update(params)
        if pr["period_log2"] is None:
            pr["period_log2"] = auto_period_log2(topic)
        pp = PortParams(seed=seed & 0xFFFFFFFFFFFFFFFF, obj_scale=pr["obj_scale"], lam_min=pr["lam_min"],
                        lam_max=pr["lam_max"], period_log2=pr["period_log2"], team=pr["team"])
        n = topic.n_partitions * topic.rf
        fin = np.zeros(n, dtype=np.uint16)
        best = np.zeros(n, dtype=np.uint16)
        st = (C.c_int64 * 6)()
        lib().kao_port_search(h, C.byref(pp), rho, launches, iters, fin.ctypes.data_as(C.POINTER(C.c_uint16)),
                              best.ctypes.data_as(C.POINTER(C.c_uint16)), st)
        return dict(final=fin.reshape(topic.n_partitions, topic.rf), best=best.reshape(topic.n_partitions, topic.rf),
                    best_obj=int(st[0]), V=int(st[1]), obj=int(st[2]), n_eval=int(st[3]) | (int(st[4]) << 32),
                    n_accept=int(st[5]))
    finally:
        lib().kao_port_ls_destroy(h)


class PortRun:
    """Launch-by-launch replay of one restart (sessions with search prices and elite launches):
    run = PortRun(topic, seed, rho); run.launch(0, iters, prices=(a, l, g)); run.launch(1, iters, elite=(assign, obj, rho)); run.read()"""

    def __init__(self, topic, seed: int, rho: int, **params):
        self.topic = topic
        self._ct = CTopic(topic)
        self._h = lib().kao_port_ls_create(C.byref(self._ct.s))
        if not self._h:
            raise ValueError("unsupported instance (RF > 8 or racks > 255)")
        pr = dict(DEFAULT_PARAMS)
        pr.update(params)
        if pr["period_log2"] is None:
            pr["period_log2"] = auto_period_log2(topic)
        self._pp = PortParams(seed=seed & 0xFFFFFFFFFFFFFFFF, obj_scale=pr["obj_scale"], lam_min=pr["lam_min"],
                              lam_max=pr["lam_max"], period_log2=pr["period_log2"], team=pr["team"])
        self._run = lib().kao_port_run_create(self._h, C.byref(self._pp), rho)

    def launch(self, launch: int, iters: int, prices=None, elite=None, gen: int = 0):
        ex = PortExtra()
        ex.gen = int(gen)   # > 0: this launch starts generation `gen` (the restart is re-initialised)
        keep = []
        if prices is not None:
            a, l, g = (np.ascontiguousarray(v, dtype=np.int32) for v in prices)
            g = np.concatenate([g, np.zeros(max(0, 256 - len(g)), dtype=np.int32)])
            keep += [a, l, g]
            p32 = C.POINTER(C.c_int32)
            ex.pa, ex.pl, ex.pg = a.ctypes.data_as(p32), l.ctypes.data_as(p32), g.ctypes.data_as(p32)
        if elite is not None:
            assign, obj, rho = elite
            e = np.ascontiguousarray(assign, dtype=np.uint16).reshape(-1)
            keep.append(e)
            ex.elite = e.ctypes.data_as(C.POINTER(C.c_uint16))
            ex.elite_obj, ex.elite_rho = int(obj), int(rho)
        lib().kao_port_run_launch(self._run, launch, iters, C.byref(ex), self.topic.n_brokers)

    def read(self) -> dict:
        t = self.topic
        n = t.n_partitions * t.rf
        fin = np.zeros(n, dtype=np.uint16)
        best = np.zeros(n, dtype=np.uint16)
        st = (C.c_int64 * 6)()
        lib().kao_port_run_read(self._run, fin.ctypes.data_as(C.POINTER(C.c_uint16)), best.ctypes.data_as(C.POINTER(C.c_uint16)), st)
        return dict(final=fin.reshape(t.n_partitions, t.rf), best=best.reshape(t.n_partitions, t.rf), best_obj=int(st[0]),
                    V=int(st[1]), obj=int(st[2]), n_eval=int(st[3]) | (int(st[4]) << 32), n_accept=int(st[5]))

    def close(self):
        if self._run:
            lib().kao_port_run_destroy(self._run)
            self._run = None
        if self._h:
            lib().kao_port_ls_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def quarter_round(v):
    """Multipliers on the quarter grid (what K-bound exports as search prices): nearest multiple of DB_SCALE / 4, half up."""
    v = np.asarray(v, dtype=np.int64)
    return (((v + 8192) >> 14) << 14).astype(np.int32)


def port_valid_fraction(topic, seed: int, rho: int, launches: int, iters: int, **params) -> float:
    """Share of one restart's delta-evaluated neighbours that were real (non-null) proposals."""
    ct = CTopic(topic)
    h = lib().kao_port_ls_create(C.byref(ct.s))
    if not h:
        raise ValueError("unsupported instance (RF > 8 or racks > 255)")
    try:
        pr = dict(DEFAULT_PARAMS)
        pr.update(params)
        if pr["period_log2"] is None:
            pr["period_log2"] = auto_period_log2(topic)
        pp = PortParams(seed=seed & 0xFFFFFFFFFFFFFFFF, obj_scale=pr["obj_scale"], lam_min=pr["lam_min"],
                        lam_max=pr["lam_max"], period_log2=pr["period_log2"], team=pr["team"])
        return float(lib().kao_port_valid_fraction(h, C.byref(pp), rho, launches, iters))
    finally:
        lib().kao_port_ls_destroy(h)


def port_search_throughput(topic, seed: int, n_restarts: int, launches: int, iters: int, threads: int, **params) -> int:
    """Replays restarts 0..n_restarts-1 of `topic` on `threads` native threads; returns the neighbours evaluated
    (bench.py's cpu_baseline leg)."""
    ct = CTopic(topic)
    h = lib().kao_port_ls_create(C.byref(ct.s))
    if not h:
        raise ValueError("unsupported instance (RF > 8 or racks > 255)")
    try:
        pr = dict(DEFAULT_PARAMS)
        pr.update(params)
        if pr["period_log2"] is None:
            pr["period_log2"] = auto_period_log2(topic)
        pp = PortParams(seed=seed & 0xFFFFFFFFFFFFFFFF, obj_scale=pr["obj_scale"], lam_min=pr["lam_min"],
                        lam_max=pr["lam_max"], period_log2=pr["period_log2"], team=pr["team"])
        return int(lib().kao_port_search_many(h, C.byref(pp), 0, n_restarts, launches, iters, threads))
    finally:
        lib().kao_port_ls_destroy(h)


DB_SCALE = 65536
INT64_MAX = (1 << 63) - 1


class DualState:
    """Multipliers + best dual value of one topic (KAO-DB, oracle/kao_port.c::kao_port_dual_bound)."""

    def __init__(self, topic):
        self.a = np.zeros(topic.n_brokers, dtype=np.int32)
        self.l = np.zeros(topic.n_brokers, dtype=np.int32)
        self.g = np.zeros(max(1, topic.n_racks), dtype=np.int32)
        self.da = np.zeros_like(self.a)   # previous direction (deflected subgradient)
        self.dl = np.zeros_like(self.l)
        self.dg = np.zeros_like(self.g)
        self.lv = np.zeros(4, dtype=np.int64)   # level control: delta, record at stage start, iterations in stage | steps << 8, best iterate value
        self.ra = np.zeros_like(self.a)         # multipliers at the record dual value (exported, rounded, as search prices)
        self.rl = np.zeros_like(self.l)
        self.rg = np.zeros_like(self.g)
        self.best_L = INT64_MAX
        self.iters = 0
        self.flags = 0

    @property
    def bound(self):
        """floor(best_L / DB_SCALE): an upper bound on the optimum (None before the first iteration)."""
        return None if self.best_L == INT64_MAX else self.best_L // DB_SCALE


def port_dual_bound(topic, target: int, iters: int, state: DualState = None) -> DualState:
    st = state or DualState(topic)
    ct = CTopic(topic)
    bl = C.c_int64(st.best_L)
    fl = C.c_int32(0)
    p32 = C.POINTER(C.c_int32)
    n = lib().kao_port_dual_bound_rec(C.byref(ct.s), int(target), int(iters), st.a.ctypes.data_as(p32), st.l.ctypes.data_as(p32),
                                      st.g.ctypes.data_as(p32), st.da.ctypes.data_as(p32), st.dl.ctypes.data_as(p32),
                                      st.dg.ctypes.data_as(p32), st.lv.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(bl), C.byref(fl),
                                      st.ra.ctypes.data_as(p32), st.rl.ctypes.data_as(p32), st.rg.ctypes.data_as(p32))
    st.best_L = int(bl.value)
    st.iters += int(n)
    st.flags = int(fl.value)
    return st


def port_dual_partition(topic, p: int, a, l, g):
    """Brute-force priced subproblem of partition p: (brokers leader first, greedy follower set, value) or None."""
    ct = CTopic(topic)
    a = np.ascontiguousarray(a, dtype=np.int32); l = np.ascontiguousarray(l, dtype=np.int32); g = np.ascontiguousarray(g, dtype=np.int32)
    S = np.zeros(4, dtype=np.int32)
    G = np.zeros(4, dtype=np.int32)
    val = C.c_int32()
    p32 = C.POINTER(C.c_int32)
    rc = lib().kao_port_dual_partition(C.byref(ct.s), int(p), a.ctypes.data_as(p32), l.ctypes.data_as(p32), g.ctypes.data_as(p32),
                                       S.ctypes.data_as(p32), G.ctypes.data_as(p32), C.byref(val))
    return None if rc else (S[:topic.rf].tolist(), G[:topic.rf].tolist(), int(val.value))
