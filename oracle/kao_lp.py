"""KAO-LP oracle -- TEST INFRASTRUCTURE ONLY (never imported by the product package).

The reference proves its optimum by lp_solve's simplex + branch and bound on the generated model (README.md:135-136,
README.md:144-185).  The device certifies an incumbent by a Lagrangian dual of the same model (K-bound, oracle/kao_port.c);
round 5 adds KAO-LP: the multipliers K-bound is evaluated at come from an interior-point solve of the model's LP relaxation
in COMPACT form.  This module restates that compact LP row by row, solves it exactly with HiGHS (scipy) and restates the
interior-point iteration with generic sparse algebra (the device runs the same iteration on the block structure).

Compact form.  A variable t?b<b>p<p>[_l] (README.md:146, README.md:182-184) of a broker that does NOT hold partition p today
has objective coefficient 0 (README.md:145-146: only existing placements carry weight), so within one rack all such brokers
are interchangeable for p up to the rows of the broker itself.  Their mass is pooled:
  per partition p:  f[p,j], l[p,j]   follower / leader variable of the j-th CURRENT replica of p (weights w[cur_role][new_role])
                    yf[p,r], yl[p,r] new follower / leader mass of p in rack r (weight 0)
  per broker b:     zf[b], zl[b]     new follower / leader mass broker b receives (broker weights, if any, ride here)
                    n[b], m[b]       replicas / leaders on b above the lower band end (0 .. hi - lo)
  per rack r:       k[r]             replicas in r above the lower band end
Rows (README lines of the family they restate):
  C1[p]   sum_j (f+l) + sum_r (yf+yl) = RF                      README.md:148-151
  C2[p]   sum_j l + sum_r yl = 1                                README.md:153-156
  C5[p,j] f + l <= 1                                            README.md:168-171 (only kept when a rack may take two replicas)
  C7[p,r] plo <= sum_{j in r} (f+l) + yf + yl <= phi            README.md:178-180
  C3[b]   sum_{(p,j) on b} (f+l) + zf[b] + zl[b] - n[b] = rep_lo        README.md:158-161
  C4[b]   sum_{(p,j) on b} l + zl[b] - m[b] = lead_lo                   README.md:163-166
  C6[r]   sum_{b in r} n[b] - k[r] = rack_lo - |r| rep_lo               README.md:173-176
  NF[r]   sum_p yf[p,r] - sum_{b in r} zf[b] = 0      (pooling)
  NL[r]   sum_p yl[p,r] - sum_{b in r} zl[b] = 0      (pooling)
It is a relaxation of the README model's LP (a pooled unit may land on a broker the partition already uses), and where at
most one replica of a partition fits a rack (phi = 1, the usual case) it loses nothing.  What the certificate rests on is not
this LP's value but the EXACT Lagrangian dual value at its row duals (oracle/kao_port.c::kao_port_dual_bound) -- valid for any
multipliers.  Measured (round 5): drifted 450 x 3500, 500 x 5000, 270 x 2200: value = the full LP's value (26330, 37558,
16459; HiGHS needed 2,876 / 10,008 / 567 s for the full LPs, 14 / 172 / 28 s for the compact ones), exact dual value at the
compact LP's duals = the same.
"""
from __future__ import annotations

import time
from dataclasses import dataclass
from typing import Dict, Tuple

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spl
from scipy.optimize import linprog

import kao_oracle as ko

DB_SCALE = 65536


@dataclass
class CompactLP:
    """min c x,  A x = b,  0 <= x <= u (u = inf where unbounded); the README objective is -c x."""
    A: sp.csr_matrix
    b: np.ndarray
    c: np.ndarray
    u: np.ndarray
    rows: Dict[str, np.ndarray]   # row indices by family: C3[B], C4[B], C6[R] (-1 = row absent), NF[R], NL[R]
    n_local_rows: int


def lp_bands(t: ko.Topic) -> dict:
    """The bands the LP is built on: t.bounds() (README.md:158-176) with the implied ends -- when the brokers' lower ends already add up to all
    the replicas (B rep_lo = P RF) no broker can be above its lower end, so the band is the point rep_lo and its slack column is left out (and
    likewise from above; likewise leaders and racks).  Same feasible set; without it such a band (config 5's "cap + 1" on a cluster whose
    average is whole) has a slack that every feasible point pins at zero, the LP has no interior and the iteration crawls.  (round 6;
    kao_lp.hip lp_open and kao_lp_port.c lp_setup do the same.)"""
    bd = dict(t.bounds())
    B, R, P, RF = t.n_brokers, t.n_racks, t.n_partitions, t.rf
    for lo, hi, n, tot in (("rep_lo", "rep_hi", B, P * RF), ("lead_lo", "lead_hi", B, P), ("rack_lo", "rack_hi", R, P * RF)):
        if n * bd[lo] == tot: bd[hi] = bd[lo]
        elif n * bd[hi] == tot: bd[lo] = bd[hi]
    return bd


def build(t: ko.Topic) -> CompactLP:
    B, R, P, RF = t.n_brokers, t.n_racks, t.n_partitions, t.rf
    bd = lp_bands(t)
    w = t.weights
    rack = np.asarray(t.rack_of)
    rsz = np.bincount(rack, minlength=R)
    bw = np.zeros(B, dtype=np.int64) if t.broker_w is None else np.asarray(t.broker_w, dtype=np.int64)
    bwl = np.zeros(B, dtype=np.int64) if t.broker_wl is None else np.asarray(t.broker_wl, dtype=np.int64)
    lo, hi, llo, lhi = bd["rep_lo"], bd["rep_hi"], bd["lead_lo"], bd["lead_hi"]
    rlo, rhi, plo, phi = bd["rack_lo"], bd["rack_hi"], bd["prack_lo"], bd["prack_hi"]
    has_c5 = phi >= 2
    ri, ci, vi = [], [], []
    c, u, b = [], [], []

    def var(cost, ub=np.inf):
        c.append(-float(cost)); u.append(ub); return len(c) - 1

    def row(rhs):
        b.append(float(rhs)); return len(b) - 1

    def put(r, v, x=1.0):
        ri.append(r); ci.append(v); vi.append(x)

    has_n = hi > lo
    NF = [row(0) for _ in range(R)]
    NL = [row(0) for _ in range(R)]
    C6 = [row(rlo - int(rsz[r]) * lo) if has_n else -1 for r in range(R)]
    C3 = [row(lo) for _ in range(B)]
    C4 = [row(llo) for _ in range(B)]
    for b_ in range(B):
        r = int(rack[b_])
        zf = var(bw[b_]); zl = var(bw[b_] + bwl[b_])
        put(C3[b_], zf); put(C3[b_], zl); put(C4[b_], zl); put(NF[r], zf, -1.0); put(NL[r], zl, -1.0)
        if has_n:
            n = var(0, hi - lo); put(C3[b_], n, -1.0); put(C6[r], n)
        if lhi > llo:
            m = var(0, lhi - llo); put(C4[b_], m, -1.0)
    if has_n and rhi > rlo:
        for r in range(R):
            k = var(0, rhi - rlo); put(C6[r], k, -1.0)
    n_glob_rows = len(b)
    for p in range(P):
        c1 = row(RF); c2 = row(1)
        c7 = [row(phi) for _ in range(R)]
        for j in range(t.rf_cur):
            b_ = int(t.current[p, j])
            if b_ == ko.NONE or b_ >= B:
                continue
            cr = 0 if j == 0 else 1
            f = var(w[cr][1] + bw[b_]); l = var(w[cr][0] + bw[b_] + bwl[b_])
            for v in (f, l):
                put(c1, v); put(C3[b_], v); put(c7[int(rack[b_])], v)
            put(c2, l); put(C4[b_], l)
            if has_c5:
                c5 = row(1); q = var(0); put(c5, f); put(c5, l); put(c5, q)
        for r in range(R):
            yf = var(0); yl = var(0)
            for v in (yf, yl):
                put(c1, v); put(c7[r], v)
            put(c2, yl); put(NF[r], yf); put(NL[r], yl)
            if phi > plo:
                tt = var(0, phi - plo if plo > 0 else np.inf); put(c7[r], tt)
    A = sp.csr_matrix((vi, (ri, ci)), shape=(len(b), len(c)))
    return CompactLP(A=A, b=np.array(b), c=np.array(c), u=np.array(u),
                     rows=dict(C3=np.array(C3), C4=np.array(C4), C6=np.array(C6), NF=np.array(NF), NL=np.array(NL)),
                     n_local_rows=len(b) - n_glob_rows)


def to_fixed(v: np.ndarray) -> np.ndarray:
    return np.clip(np.round(np.asarray(v) * DB_SCALE), -(1 << 26), 1 << 26).astype(np.int32)


def duals_to_alg(t: ko.Topic, lp: CompactLP, y: np.ndarray):
    rack = np.asarray(t.rack_of)
    C6 = lp.rows["C6"]
    g = np.where(C6 >= 0, -y[np.maximum(C6, 0)], 0.0)
    a = -y[lp.rows["C3"]] - g[rack]
    l = -y[lp.rows["C4"]]
    return to_fixed(a), to_fixed(l), to_fixed(g)


def solve_highs(lp: CompactLP, method: str = "highs-ipm"):
    """Exact reference: HiGHS on the compact LP.  Returns (README-objective value, row duals, x, seconds)."""
    bounds = [(0.0, None if not np.isfinite(ub) else ub) for ub in lp.u]
    t0 = time.time()
    res = linprog(lp.c, A_eq=lp.A, b_eq=lp.b, bounds=bounds, method=method)
    if res.status != 0:
        raise RuntimeError(f"HiGHS status {res.status}: {res.message}")
    return -float(res.fun), np.asarray(res.eqlin.marginals), np.asarray(res.x), time.time() - t0


def exact_dual_value(t: ko.Topic, a, l, g) -> float:
    """The exact Lagrangian dual value (oracle/kao_port.c, one evaluation after the common shifts) at given multipliers."""
    import kao_port as kp
    st = kp.DualState(t)
    st.a[:] = a; st.l[:] = l; st.g[:len(g)] = g
    kp.port_dual_bound(t, 0, 1, st)
    return st.best_L / DB_SCALE


MCC_DELTA, MCC_BMIN, MCC_BMAX = 0.3, 0.1, 10.0      # Gondzio's centrality correctors (oracle/kao_lp_port.c, kao_lp.hip)
START_X_FLOOR = 0.1                                 # the starting point's x = max(x~, this), capped at half the upper bound (1.0 until late in round 6: notes section 25)
SIGMA_EXP = 10                                      # sigma = (mu_aff / mu)^SIGMA_EXP (Mehrotra's 3 until late in round 6: docs/notes_r06.md section 24)
SIGMA_EXP_HUGE, SIGMA_HUGE_SLOTS = 24, 131072       # ... and ^SIGMA_EXP_HUGE on topics of more than SIGMA_HUGE_SLOTS replica slots (section 29; ipm's `sigma_exp`)
STEP_FRACTION, STEP_FRACTION_MAX = 0.9, 0.9995      # a blocked step of length a goes min(MAX, max(FRACTION, a)) of the way to the boundary (kao_lp.hip k_lp_sc_final; 0.9995 throughout until late in round 6)


def ipm(lp: CompactLP, tol: float = 1e-7, maxit: int = 80, reg: float = 1e-10, trace=None, mcc: int = 2, sigma_exp: int = SIGMA_EXP):
    """Mehrotra predictor-corrector on min c x, A x = b, 0 <= x <= u -- the iteration kao_lp.hip runs on the block structure
    (same starting point, same step rule, same stopping rule), here with generic sparse algebra.  Returns (x, y, iterations,
    primal objective, dual objective).  `trace`, if a list, receives (mu, pobj, dobj, pinf, dinf) per iteration.  Up to `mcc`
    centrality correctors per iteration: the step lengths of the predictor-corrector direction are enlarged by MCC_DELTA, the
    complementarity products of that trial point are projected onto [MCC_BMIN, MCC_BMAX] x sigma mu, the direction that moves them
    there is added when it lengthens a step by at least a hundredth of MCC_DELTA and shortens neither by more than a tenth."""
    A, b, c, u = lp.A, lp.b, lp.c, lp.u
    m, n = A.shape
    U = np.isfinite(u)
    uu = np.where(U, u, 0.0)
    nU = int(U.sum())
    AT = A.T.tocsr()

    def factor(theta):
        return spl.splu((A @ sp.diags(theta) @ AT).tocsc() + reg * sp.identity(m, format="csc"))

    lu = factor(np.ones(n))
    x = AT @ lu.solve(b)
    y = lu.solve(A @ c)
    s = c - AT @ y
    x = np.maximum(x, START_X_FLOOR)
    x = np.where(U, np.minimum(x, np.maximum(uu * 0.5, 1e-2)), x)
    w = np.where(U, uu - x, 1.0)
    s = np.maximum(s, 1.0)
    v = np.where(U, 1.0, 0.0)
    nb, nc = 1.0 + np.linalg.norm(b), 1.0 + np.linalg.norm(c)
    it = 0
    pobj = dobj = 0.0
    for it in range(maxit + 1):
        rp = b - A @ x
        rd = c - AT @ y - s + v
        mu = (x @ s + (w * v)[U].sum()) / (n + nU)
        pobj = float(c @ x); dobj = float(b @ y - (uu * v)[U].sum())
        pinf, dinf = np.linalg.norm(rp) / nb, np.linalg.norm(rd) / nc
        if trace is not None:
            trace.append((mu, pobj, dobj, pinf, dinf))
        if (abs(pobj - dobj) / (1.0 + abs(pobj)) < tol and pinf < 100 * tol and dinf < tol) or it == maxit:
            break
        wU = np.where(U, w, 1.0)
        theta = 1.0 / (s / x + np.where(U, v / wU, 0.0))
        lu = factor(theta)

        def direction(rxs, rwv):
            h = rd - rxs / x + np.where(U, rwv / wU, 0.0)
            dy = lu.solve(rp + A @ (theta * h))
            dx = theta * (AT @ dy - h)
            ds = (rxs - s * dx) / x
            dv = np.where(U, (rwv + v * dx) / wU, 0.0)
            return dx, dy, ds, dv

        def maxstep(z, dz, mask=None):
            neg = dz < 0
            if mask is not None:
                neg = neg & mask
            return min(1.0, float((-z[neg] / dz[neg]).min())) if neg.any() else 1.0

        dx, dy, ds, dv = direction(-x * s, np.where(U, -w * v, 0.0))
        ap = min(maxstep(x, dx), maxstep(w, -dx, U)); ad = min(maxstep(s, ds), maxstep(v, dv, U))
        mu_aff = ((x + ap * dx) @ (s + ad * ds) + ((w - ap * dx) * (v + ad * dv))[U].sum()) / (n + nU)
        sigma = mu_aff / mu
        for _ in range(sigma_exp - 1):      # the powers multiplied up one by one, as the restatement and the device do
            sigma *= mu_aff / mu
        dx, dy, ds, dv = direction(sigma * mu - x * s - dx * ds, np.where(U, sigma * mu - w * v + dx * dv, 0.0))
        ap = min(maxstep(x, dx), maxstep(w, -dx, U)); ad = min(maxstep(s, ds), maxstep(v, dv, U))
        mut = sigma * mu
        for _ in range(mcc):
            if not (ap < 1.0 or ad < 1.0):
                break
            apt, adt = min(1.0, ap + MCC_DELTA), min(1.0, ad + MCC_DELTA)
            pr = (x + apt * dx) * (s + adt * ds)
            rxs = np.maximum(np.clip(pr, MCC_BMIN * mut, MCC_BMAX * mut) - pr, -MCC_BMAX * mut)
            pr2 = (w - apt * dx) * (v + adt * dv)
            rwv = np.where(U, np.maximum(np.clip(pr2, MCC_BMIN * mut, MCC_BMAX * mut) - pr2, -MCC_BMAX * mut), 0.0)
            hc = -rxs / x + np.where(U, rwv / wU, 0.0)
            dyc = lu.solve(A @ (theta * hc))
            dxc = theta * (AT @ dyc - hc)
            dsc = (rxs - s * dxc) / x
            dvc = np.where(U, (rwv + v * dxc) / wU, 0.0)
            ap2 = min(maxstep(x, dx + dxc), maxstep(w, -(dx + dxc), U)); ad2 = min(maxstep(s, ds + dsc), maxstep(v, dv + dvc, U))
            if not (ap2 >= ap + 0.01 * MCC_DELTA or ad2 >= ad + 0.01 * MCC_DELTA) or ap2 < 0.9 * ap or ad2 < 0.9 * ad:
                break
            dx, dy, ds, dv, ap, ad = dx + dxc, dy + dyc, ds + dsc, dv + dvc, ap2, ad2
        ap = min(STEP_FRACTION_MAX, max(STEP_FRACTION, ap)) * ap if ap < 1.0 else 1.0
        ad = min(STEP_FRACTION_MAX, max(STEP_FRACTION, ad)) * ad if ad < 1.0 else 1.0
        x = x + ap * dx; w = np.where(U, uu - x, 1.0)
        y = y + ad * dy; s = s + ad * ds; v = v + ad * dv
    return x, y, it, -pobj, -dobj


# ---- the block-structured restatement in C (oracle/kao_lp_port.c) ------------------------------------------------
_PORT = None


def _port():
    global _PORT
    if _PORT is None:
        import ctypes as C
        import os
        import subprocess
        here = os.path.dirname(os.path.abspath(__file__))
        so, src = os.path.join(here, "libkao_lp_port.so"), os.path.join(here, "kao_lp_port.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", here, "libkao_lp_port.so"], stdout=subprocess.DEVNULL)
        import kao_port as kp
        _PORT = C.CDLL(so)
        _PORT.kao_lp_port_solve.argtypes = [C.POINTER(kp.PortTopic), C.c_double, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                            C.POINTER(C.c_double)]
        _PORT.kao_lp_port_solve.restype = C.c_int
        _PORT.kao_lp_port_solve_x.argtypes = _PORT.kao_lp_port_solve.argtypes + [C.POINTER(C.c_double), C.POINTER(C.c_double)]
        _PORT.kao_lp_port_solve_x.restype = C.c_int
        _PORT.kao_lp_port_solve_p.argtypes = [C.POINTER(kp.PortTopic), C.c_double, C.c_int, C.c_double, C.c_uint32] + [C.POINTER(C.c_double)] * 5
        _PORT.kao_lp_port_solve_p.restype = C.c_int
    return _PORT


def port_coupling_order(t: ko.Topic, lp: CompactLP) -> np.ndarray:
    """Row indices of `lp` in the order oracle/kao_lp_port.c (and the device) number the coupling rows:
    NF[R], NL[R], C6[R], then C3[b], C4[b] interleaved (-1 = row absent)."""
    B = t.n_brokers
    inter = np.empty(2 * B, dtype=np.int64)
    inter[0::2] = lp.rows["C3"]; inter[1::2] = lp.rows["C4"]
    return np.concatenate([lp.rows["NF"], lp.rows["NL"], lp.rows["C6"], inter])


def port_solve(t: ko.Topic, tol: float = 1e-7, maxit: int = 80, primal: bool = False, pert: float = 0.0, salt: int = 0):
    """The structured iteration on the CPU.  Returns dict(status, iterations, primal, dual (README objective), y (coupling-row
    duals in the port's order), a, l, g (fixed-point multipliers), trace [(mu, pobj, dobj, pinf, dinf)])."""
    import ctypes as C
    import kao_port as kp
    ct = kp.CTopic(t)
    B, R = t.n_brokers, t.n_racks
    mc = 3 * R + 2 * B
    y = np.zeros(mc)
    trace = np.zeros(5 * (maxit + 2))
    stats = np.zeros(4)
    pd = C.POINTER(C.c_double)
    x = np.zeros((3 * t.rf_cur + 3 * R) * t.n_partitions) if primal else None
    xg = np.zeros(4 * B + R) if primal else None
    rc = _port().kao_lp_port_solve_p(C.byref(ct.s), tol, maxit, float(pert), int(salt) & 0xFFFFFFFF, y.ctypes.data_as(pd), trace.ctypes.data_as(pd),
                                     stats.ctypes.data_as(pd), x.ctypes.data_as(pd) if primal else None, xg.ctypes.data_as(pd) if primal else None)
    it = int(stats[0])
    rack = np.asarray(t.rack_of)
    g = -y[2 * R:3 * R]
    a = -y[3 * R::2] - g[rack]
    l = -y[3 * R + 1::2]
    out = dict(status=rc, iterations=it, primal=float(stats[1]), dual=float(stats[2]), y=y, a=to_fixed(a), l=to_fixed(l), g=to_fixed(g),
               trace=trace[:5 * (it + 1)].reshape(-1, 5))
    if primal:   # x[v, p]: v = 3 j + {0 f, 1 l, 2 q} for current replica j, then 3 NJ + 3 r + {0 yf, 1 yl, 2 t}; xg: zf[B] zl[B] n[B] m[B] k[R]
        out["x"] = x.reshape(3 * t.rf_cur + 3 * R, t.n_partitions)
        out["xg"] = xg
    return out


# ---- the primal side (round 5): from the iterate of the PERTURBED LP to an assignment ------------------------------------------
# Specification of kao_lp_round (include/kao.h) / lp_round_assignment (kao_round.cpp): same quantisation, same order, same ties.

MAX_CAND = 16          # candidate brokers of a fractional partition
MAX_ROWS = 256         # candidate rows kept per fractional partition (by objective weight)
MAX_NODES = 100000     # search nodes over the fractional partitions
MAX_SEARCH = 64        # more fractional partitions than this: no search over their rows (the iterate is far from a vertex)
ROUND_TOL_C = 30       # a variable farther than 0.30 from an integer makes its partition "fractional"
PERT_SLOTS = 100.0     # default perturbation: eps = min(1e-2, PERT_SLOTS / (partitions * rf))


def default_pert(t: ko.Topic) -> float:
    return min(1e-2, PERT_SLOTS / (t.n_partitions * t.rf))


def quantise(x) -> np.ndarray:
    """centi-units as the device packs them: min(250, rint(100 x)), negative values 0"""
    return np.minimum(250, np.maximum(0, np.rint(100.0 * np.asarray(x, dtype=np.float64)))).astype(np.int64)


def primal_blocks(t: ko.Topic, x: np.ndarray, xg: np.ndarray):
    """(F, L [P][rf_cur], YF, YL [P][R], ZF, ZL [B]) in centi-units from port_solve(primal=True)'s x / xg."""
    NJ, R, B = t.rf_cur, t.n_racks, t.n_brokers
    F = quantise(np.stack([x[3 * j] for j in range(NJ)], axis=1)); L = quantise(np.stack([x[3 * j + 1] for j in range(NJ)], axis=1))
    YF = quantise(np.stack([x[3 * NJ + 3 * r] for r in range(R)], axis=1)); YL = quantise(np.stack([x[3 * NJ + 3 * r + 1] for r in range(R)], axis=1))
    return F, L, YF, YL, np.rint(xg[:B]).astype(np.int64), np.rint(xg[B:2 * B]).astype(np.int64)


def compact_index(t: ko.Topic) -> dict:
    """Variable indices of build(t), by replaying its loops: zf[B], zl[B], f[P][rf_cur], l[P][rf_cur] (-1 = absent), yf[P][R], yl[P][R], n."""
    B, R, P = t.n_brokers, t.n_racks, t.n_partitions
    bd = lp_bands(t)
    lo, hi, llo, lhi = bd["rep_lo"], bd["rep_hi"], bd["lead_lo"], bd["lead_hi"]
    rlo, rhi, plo, phi = bd["rack_lo"], bd["rack_hi"], bd["prack_lo"], bd["prack_hi"]
    has_c5, has_n = phi >= 2, hi > lo
    n = 0
    zf = np.zeros(B, dtype=np.int64); zl = np.zeros(B, dtype=np.int64)
    for b in range(B):
        zf[b] = n; zl[b] = n + 1; n += 2
        if has_n: n += 1
        if lhi > llo: n += 1
    if has_n and rhi > rlo: n += R
    f = -np.ones((P, t.rf_cur), dtype=np.int64); l = -np.ones((P, t.rf_cur), dtype=np.int64)
    yf = np.zeros((P, R), dtype=np.int64); yl = np.zeros((P, R), dtype=np.int64)
    for p in range(P):
        for j in range(t.rf_cur):
            b = int(t.current[p, j])
            if b == ko.NONE or b >= B: continue
            f[p, j] = n; l[p, j] = n + 1; n += 2
            if has_c5: n += 1
        for r in range(R):
            yf[p, r] = n; yl[p, r] = n + 1; n += 2
            if phi > plo: n += 1
    return dict(zf=zf, zl=zl, f=f, l=l, yf=yf, yl=yl, n=n)


def blocks_from_compact(t: ko.Topic, x: np.ndarray):
    """(F, L, YF, YL, ZF, ZL) in centi-units / rounded inflows from a solution vector of build(t) (solve_highs's x)."""
    ix = compact_index(t)
    g = lambda idx: np.where(idx >= 0, x[np.maximum(idx, 0)], 0.0)
    return quantise(g(ix["f"])), quantise(g(ix["l"])), quantise(g(ix["yf"])), quantise(g(ix["yl"])), np.rint(g(ix["zf"])).astype(np.int64), np.rint(g(ix["zl"])).astype(np.int64)


def round_primal(t: ko.Topic, F, L, YF, YL, ZF, ZL, fallback=None, tolc: int = ROUND_TOL_C):
    """The compact LP pools the NEW replicas of a partition per rack (yf, yl) and counts what every broker receives (zf, zl): an
    integral solution still has to hand the new replicas of a rack to that rack's brokers.  Any way that respects the inflows and
    puts no broker twice into a partition (row C5, README.md:168-171) is as good as any other -- the objective (README.md:145-146)
    only sees current replicas -- so:
      * partitions in ascending order; kept current replicas first (leader: the lowest j with l_j = 1); then racks in ascending
        order, the new leader of a rack before its new followers; a new replica goes to the broker of the rack with the LARGEST
        remaining inflow that is not in the row yet (ties: the lowest index);
      * when every broker with inflow left is in the row already, ONE swap is tried: an earlier partition of the rack that holds a
        new replica of the same kind on a broker b' outside this row moves it to a broker with inflow left that is outside ITS row,
        and b' goes to this partition (the first such b' in index order, the earliest such partition); failing that the replica goes
        to the lowest-index broker of the rack outside the row (`over_inflow`);
      * a partition with a variable farther than tolc / 100 from an integer, or whose row comes out incomplete, is FRACTIONAL: it
        keeps its row of `fallback` (the incumbent) when there is one; else the fractional partitions are completed TOGETHER after
        all others.  In the plain case (one replica per rack and partition, RF <= 4, no broker weights) rows that had to leave the
        inflows, and the last rows of a broker the pass left over a band, are given up and join them while the set stays within
        PAT_MAX_PARTS partitions; then `complete_by_patterns` (below: which current replicas every partition keeps, the new
        replicas matched to the brokers still below their band) -- taken only if every band row holds afterwards;
      * otherwise, up to MAX_SEARCH partitions: candidate rows from each partition's support plus the brokers of its racks that are
        below their band, chosen by a bounded two-pass branch and bound (first only completions that leave no broker below its
        band, then the least violated one); more than MAX_SEARCH (the iterate is far from a vertex): the first admissible row of
        every partition in turn;
      * what that leaves outside the bands is put right by `repair_bands`.
    Returns (A [P][RF] dense broker indices, leader first; report dict)."""
    import itertools
    B, R, P, RF, NJ = t.n_brokers, t.n_racks, t.n_partitions, t.rf, t.rf_cur
    bd = lp_bands(t)      # the implied ends: same feasible set (kao_round.cpp does the same)
    phi = bd["prack_hi"]
    rack = [int(r) for r in np.asarray(t.rack_of)]
    members = [[b for b in range(B) if rack[b] == r] for r in range(R)]
    capf = [int(v) for v in ZF]; capl = [int(v) for v in ZL]
    A = np.zeros((P, RF), dtype=np.int64)
    rep = dict(fractional=0, over_inflow=0, unplaced=0, from_fallback=0, swaps=0)
    placed = [[[], []] for _ in range(R)]      # per rack, per kind (0 follower, 1 leader): (partition, slot) of new replicas handed out

    def frac(c):
        return abs(int(c) - 100 * ((int(c) + 50) // 100)) > tolc

    def take(p, slot, r, kind, used):
        cap = capl if kind else capf
        best = -1
        for b in members[r]:
            if cap[b] > 0 and b not in used and (best < 0 or cap[b] > cap[best]):
                best = b
        if best >= 0:
            cap[best] -= 1
            return best, True
        for b1 in members[r]:          # one swap
            if b1 in used: continue
            for (q, sq) in placed[r][kind]:
                if int(A[q, sq]) != b1: continue
                rowq = [int(x) for x in A[q]]
                for b2 in members[r]:
                    if cap[b2] > 0 and b2 not in rowq:
                        cap[b2] -= 1; A[q, sq] = b2; rep["swaps"] += 1; swaps_now.append((q, sq, b1))
                        return b1, True
        for b in members[r]:
            if b not in used:
                rep["over_inflow"] += 1
                return b, False
        rep["unplaced"] += 1
        return -1, False

    pending = []
    overp = []
    swaps_now = []
    for p in range(P):
        cur = [int(b) if (int(b) != ko.NONE and int(b) < B) else -1 for b in t.current[p]]
        vals = [F[p, j] for j in range(NJ) if cur[j] >= 0] + [L[p, j] for j in range(NJ) if cur[j] >= 0] + list(YF[p]) + list(YL[p])
        if any(frac(c) for c in vals):
            pending.append(p); continue
        lead = -1; row = []
        for j in range(NJ):
            if cur[j] < 0: continue
            if lead < 0 and (int(L[p, j]) + 50) // 100 >= 1: lead = cur[j]
            elif (int(F[p, j]) + 50) // 100 >= 1: row.append(cur[j])
        n_kept = len(row)
        used = set(row) | ({lead} if lead >= 0 else set())
        # the row is built in A[p] itself (a swap reads the rows of earlier partitions there); an incomplete row gives back what it took
        sf, sl, so, su, ss = list(capf), list(capl), rep["over_inflow"], rep["unplaced"], rep["swaps"]
        del swaps_now[:]
        new_slots = []
        ok = len(used) <= RF and len(row) <= RF - 1
        A[p] = -1
        for r in range(R):
            if not ok: break
            if lead < 0 and (int(YL[p, r]) + 50) // 100 >= 1:
                b, within = take(p, 0, r, 1, used)
                if b >= 0:
                    lead = b; used.add(b); A[p, 0] = b
                    if within: new_slots.append((r, 1, 0))
            for _ in range((int(YF[p, r]) + 50) // 100):
                if len(row) >= RF - 1: break
                b, within = take(p, 1 + len(row), r, 0, used)
                if b >= 0:
                    row.append(b); used.add(b); A[p, len(row)] = b
                    if within: new_slots.append((r, 0, len(row)))
        if not ok or lead < 0 or len(row) != RF - 1:
            capf[:], capl[:] = sf, sl
            for (q, sq, b1) in reversed(swaps_now): A[q, sq] = b1      # (the swaps made for this row are undone with it)
            rep["over_inflow"], rep["unplaced"], rep["swaps"] = so, su, ss
            pending.append(p); continue
        A[p, 0] = lead; A[p, 1:] = row
        for (r, kind, slot) in new_slots: placed[r][kind].append((p, slot))
        if rep["over_inflow"] > so: overp.append(p)
    # rows that took a replica outside the inflows (its broker ends over its band) are given up again when the pattern completion
    # can take them along: it sees the bands, not the inflows
    plain = phi == 1 and RF <= 4 and getattr(t, "broker_w", None) is None and getattr(t, "broker_wl", None) is None
    if overp and plain and len(pending) + len(overp) <= PAT_MAX_PARTS:
        pending = sorted(pending + overp)
    rep["fractional"] = len(pending)
    done = np.ones(P, dtype=bool); done[pending] = False
    load = np.zeros(B, dtype=np.int64); lead_load = np.zeros(B, dtype=np.int64)
    for p in range(P):
        if done[p]:
            for k in range(RF): load[int(A[p, k])] += 1
            lead_load[int(A[p, 0])] += 1
    lo, hi, llo, lhi = bd["rep_lo"], bd["rep_hi"], bd["lead_lo"], bd["lead_hi"]
    # likewise a broker the rows now set put over a band (inflows of an iterate that has not converged need not add up to the bands):
    # the last rows that hold it are given up, one per unit of excess (as a new replica first: weightless), while the set stays small
    if plain and 0 < len(pending) < PAT_MAX_PARTS and (bool((load > hi).any()) or bool((lead_load > lhi).any())):
        extra = []
        for want in (0, 1, 2):       # 0: the broker sits there as a new replica; 1: as a kept follower; 2: as the leader it was
            for p in range(P - 1, -1, -1):
                if len(pending) + len(extra) >= PAT_MAX_PARTS or not (bool((load > hi).any()) or bool((lead_load > lhi).any())): break
                if not done[p]: continue
                row = [int(x) for x in A[p]]
                curp = [int(x) for x in t.current[p]]
                if want == 0: hit = any(load[b] > hi and b not in curp for b in row) or (lead_load[row[0]] > lhi and row[0] not in curp)
                elif want == 1: hit = any(load[b] > hi for b in row[1:])
                else: hit = load[row[0]] > hi or lead_load[row[0]] > lhi
                if hit:
                    extra.append(p); done[p] = False
                    for b in row: load[b] -= 1
                    lead_load[row[0]] -= 1
        pending = sorted(pending + extra)
        rep["fractional"] = len(pending)
    if fallback is not None:
        for p in pending:
            A[p] = np.asarray(fallback[p]); rep["from_fallback"] += 1
        return A, rep
    # ---- first attempt: PATTERNS.  The weight of a completion comes from the current replicas a partition keeps (README.md:145-146);
    #      the new replicas are weightless and interchangeable.  So: per partition the patterns (leader: a current replica or a new one;
    #      followers: a subset of the other current replicas; kept brokers in distinct racks), heaviest first; depth first over the
    #      partitions with the sum of the best remaining patterns as the bound; at a leaf the new slots are matched to the brokers still
    #      below their band (leader slots first).  Only the plain case (one replica per rack, no broker weights, few partitions); the
    #      result is checked against the band rows before it is taken -- otherwise the search over candidate rows below runs as before.
    #      The weight the iterate itself gives these partitions bounds what a completion can reach beside the rows already set, so the
    #      search stops as soon as it is met.
    target_c = 0
    for p in pending:
        for j in range(NJ):
            b = int(t.current[p, j])
            if b != ko.NONE and b < B:
                target_c += t.weights[0 if j == 0 else 1][1] * int(F[p, j]) + t.weights[0 if j == 0 else 1][0] * int(L[p, j])
    got = complete_by_patterns(t, A, pending, load, lead_load, (target_c + 25) // 100, F, L)
    rep["patterns"] = int(got)
    if got:
        rep["repaired"] = repair_bands(t, A)
        return A, rep
    # ---- the fractional partitions, together: candidate rows from their support, chosen by a bounded depth-first search so that the
    #      band rows (README.md:158-166) come out right given what the other partitions hold ----
    w = t.weights
    bwv = np.zeros(B, dtype=np.int64) if getattr(t, "broker_w", None) is None else np.asarray(t.broker_w, dtype=np.int64)
    bwlv = np.zeros(B, dtype=np.int64) if getattr(t, "broker_wl", None) is None else np.asarray(t.broker_wl, dtype=np.int64)
    rows_of = []
    for p in pending:
        cur = [int(b) if (int(b) != ko.NONE and int(b) < B) else -1 for b in t.current[p]]
        cand = []
        for j in range(NJ):
            if cur[j] >= 0 and (int(F[p, j]) > 0 or int(L[p, j]) > 0): cand.append(cur[j])
        for r in range(R):
            if int(YF[p, r]) > 0 or int(YL[p, r]) > 0:
                best = sorted((b for b in members[r] if b not in cand and load[b] < hi), key=lambda b: (-(lo - int(load[b])), -(llo - int(lead_load[b])), b))
                n_short = sum(1 for b in best if load[b] < lo)
                cand += best[:max(2, min(8, n_short))]      # every broker of the rack that is below its band (up to 8), two at least
        cand = cand[:MAX_CAND]
        if len(cand) < RF:      # (mass on too few options: the brokers that need replicas most)
            for b in sorted((b for b in range(B) if b not in cand), key=lambda b: (-(lo - int(load[b])), b)):
                cand.append(b)
                if len(cand) >= RF + 2: break
        wl = {}; wf = {}
        for b in cand:
            wf[b] = int(bwv[b]); wl[b] = int(bwv[b]) + int(bwlv[b])
        for j, b in enumerate(cur):
            if b >= 0 and b in wl: wl[b] += w[0 if j == 0 else 1][0]; wf[b] += w[0 if j == 0 else 1][1]
        rows = []
        for ld in cand:
            others = [b for b in cand if b != ld]
            for fol in itertools.combinations(others, RF - 1):
                rowb = (ld,) + fol
                cnt = {}
                for b in rowb: cnt[rack[b]] = cnt.get(rack[b], 0) + 1
                if max(cnt.values()) > phi: continue
                rows.append((wl.get(ld, 0) + sum(wf.get(b, 0) for b in fol), rowb))
        rows.sort(key=lambda x: -x[0])       # (stable: equal weights keep the enumeration order)
        rows_of.append(rows[:MAX_ROWS])
    best = dict(viol=None, obj=None, pick=None); nodes = [0]; perfect = [False]
    pick = [None] * len(pending)
    # cover[i][b] / lcover[i][b]: how many of the partitions i.. can still put a replica / their leader on broker b -- a deficit beyond that
    # stays whatever the rest of the search does (the bound that keeps the search small on rigid bands)
    npend = len(pending)
    short = [b for b in range(B) if load[b] < lo]; lshort = [b for b in range(B) if lead_load[b] < llo]
    cover = [dict.fromkeys(short, 0) for _ in range(npend + 1)]; lcover = [dict.fromkeys(lshort, 0) for _ in range(npend + 1)]
    if npend <= MAX_SEARCH:
        for i in range(npend - 1, -1, -1):
            cover[i] = dict(cover[i + 1]); lcover[i] = dict(lcover[i + 1])
            inrow = set(); leads = set()
            for wv, rowb in rows_of[i]:
                inrow.update(rowb); leads.add(rowb[0])
            for b in inrow:
                if b in cover[i]: cover[i][b] += 1
            for b in leads:
                if b in lcover[i]: lcover[i][b] += 1

    wmax = [0] * (npend + 1)      # the most the partitions i.. can add to the objective
    for i in range(npend - 1, -1, -1):
        wmax[i] = wmax[i + 1] + (rows_of[i][0][0] if rows_of[i] else 0)

    def lower_bound(i):
        v = 0
        for b in short:
            d = lo - int(load[b]) - cover[i][b]
            if d > 0: v += d
        for b in lshort:
            d = llo - int(lead_load[b]) - lcover[i][b]
            if d > 0: v += d
        return v

    def leaf_viol():
        return int(np.maximum(lo - load, 0).sum() + np.maximum(llo - lead_load, 0).sum())

    def dfs(i, obj):
        if nodes[0] > MAX_NODES: return
        nodes[0] += 1
        if i == len(pending):
            v = leaf_viol()
            if best["viol"] is None or (v, -obj) < (best["viol"], -best["obj"]):
                best.update(viol=v, obj=obj, pick=list(pick))
            return
        lb = lower_bound(i)
        if perfect[0] and lb > 0: return                  # (first pass: only completions that leave no broker below its band)
        if best["viol"] is not None and (lb > best["viol"] or (lb == best["viol"] and obj + wmax[i] <= best["obj"])): return      # cannot beat the best so far
        any_row = False
        for wv, rowb in rows_of[i]:
            if any(load[b] >= hi for b in rowb) or lead_load[rowb[0]] >= lhi: continue
            any_row = True
            for b in rowb: load[b] += 1
            lead_load[rowb[0]] += 1
            pick[i] = rowb
            dfs(i + 1, obj + wv)
            for b in rowb: load[b] -= 1
            lead_load[rowb[0]] -= 1
        if not any_row and not rows_of[i]:
            pick[i] = None
            dfs(i + 1, obj)
        if not any_row and rows_of[i]:      # every row breaks an upper band end: take the first, the violation is counted by K-eval
            wv, rowb = rows_of[i][0]
            for b in rowb: load[b] += 1
            lead_load[rowb[0]] += 1
            pick[i] = rowb
            dfs(i + 1, obj + wv)
            for b in rowb: load[b] -= 1
            lead_load[rowb[0]] -= 1

    if len(pending) > MAX_SEARCH:     # far from a vertex: no search, the first admissible row of every partition in turn
        best["pick"] = []
        for i in range(len(pending)):
            ri = 0
            for k, (wv, rowb) in enumerate(rows_of[i]):
                if not (any(load[b] >= hi for b in rowb) or lead_load[rowb[0]] >= lhi):
                    ri = k; break
            if rows_of[i]:
                rowb = rows_of[i][ri][1]
                for b in rowb: load[b] += 1
                lead_load[rowb[0]] += 1
                best["pick"].append(rowb)
            else:
                best["pick"].append(None)
    else:
        perfect[0] = True
        dfs(0, 0)
        if best["viol"] is None or best["viol"] > 0:      # no completion without a violation among the candidate rows (or not found in time): the least violated one
            perfect[0] = False; nodes[0] = 0
            dfs(0, 0)
    rep["dfs_nodes"] = nodes[0]
    for i, p in enumerate(pending):
        rowb = best["pick"][i] if best["pick"] is not None and best["pick"][i] is not None else (rows_of[i][0][1] if rows_of[i] else tuple(range(RF)))
        A[p] = rowb
    rep["repaired"] = repair_bands(t, A)
    return A, rep


PAT_MAX_PARTS = 64       # pattern completion: fractional partitions at most
PAT_MAX_NODES = 60000    # ... nodes of the pattern search and of all its leaf matchings together


def complete_by_patterns(t: ko.Topic, A, pending, load0, lead0, target: int, F, L) -> bool:
    """Specification of the pattern completion (product: kao_round.cpp, the block marked PATTERNS).  The weight of a completion comes
    from the current replicas a partition keeps (README.md:145-146: w[cur_role][new_role]); new replicas are weightless and
    interchangeable.  `load0` / `lead0`: replica and leader counts of the rows already set; `target`: the weight the iterate gives
    the pending partitions (floor of the centi-sum + 0.25), F / L its centi-masses.
      * Not tried when a broker is already over a band, or when more is missing below the bands than the partitions have slots.
      * Patterns of a partition: leader = one of its current replicas or a new one, followers = a subset of the others, kept brokers
        in distinct racks and with room in their bands even now; order: weight descending, then the iterate's mass on the kept
        replicas descending, then enumeration order (leader: new, then current j ascending; subsets by size, lexicographic).
      * Depth first over the partitions in ascending order.  Bounds: weight so far + the best patterns of the remaining partitions;
        and, once a completion is known, + per broker the heaviest kept replicas of the remaining partitions its band still has room
        for.  A pattern is skipped when a kept broker has no room, or when a follower it keeps leaves its broker fewer free places
        than that broker is still short of leaders (llo - lead > hi - load: loads only grow down the tree).
      * Leaf: the new slots (a new leader where the pattern keeps none; the new followers) are matched to brokers depth first, leader
        slots first, within a group the slot with the fewest admissible brokers below their band first; candidates of a slot: brokers
        with room when the matching began, not in the row, rack not in the row, leader slot: below lead_hi, follower slot: still
        leaving room for the leaders the broker is short of; ordered by leaders short (leader slots), replicas short, index; the
        first 12.  A matching succeeds when every broker that was short ends inside its bands.
      * The first completion whose weight reaches `target` ends the search; otherwise PAT_MAX_NODES nodes (pattern nodes and matching
        nodes together) and the best completion found.  The result is checked against rows C1-C8 as they apply here (complete rows,
        distinct brokers and racks, every band) before A is touched.
    On success the rows of `pending` in A are set and every broker is inside its bands; else A is untouched."""
    import itertools
    B, R, P, RF, NJ = t.n_brokers, t.n_racks, t.n_partitions, t.rf, t.rf_cur
    bd = lp_bands(t)      # the implied ends: same feasible set (kao_round.cpp does the same)
    lo, hi, llo, lhi, phi = bd["rep_lo"], bd["rep_hi"], bd["lead_lo"], bd["lead_hi"], bd["prack_hi"]
    if phi != 1 or RF > 4 or not pending or len(pending) > PAT_MAX_PARTS: return False
    if getattr(t, "broker_w", None) is not None or getattr(t, "broker_wl", None) is not None: return False
    rack = [int(r) for r in np.asarray(t.rack_of)]
    w = t.weights
    load = [int(v) for v in load0]; lead = [int(v) for v in lead0]
    npd = len(pending)
    # no completion can be perfect when the rows already set put a broker over a band or leave more to fill than these partitions have
    if any(load[b] > hi or lead[b] > lhi for b in range(B)): return False
    if sum(max(0, lo - load[b]) for b in range(B)) > RF * npd or sum(max(0, llo - lead[b]) for b in range(B)) > npd: return False
    pats = []
    items = []       # per partition: (current replica, weight kept as leader, weight kept as follower)
    for p in pending:
        cur = []
        for j in range(NJ):
            b = int(t.current[p, j])
            if b != ko.NONE and b < B and b not in [c[0] for c in cur]:
                cur.append((b, w[0 if j == 0 else 1][0], w[0 if j == 0 else 1][1], int(L[p, j]), int(F[p, j])))
        items.append([c[:3] for c in cur])
        lst = []
        for li in [-1] + list(range(len(cur))):
            others = [k for k in range(len(cur)) if k != li]
            for sz in range(0, min(RF - 1, len(others)) + 1):
                for fs in itertools.combinations(others, sz):
                    kept = ([cur[li][0]] if li >= 0 else []) + [cur[k][0] for k in fs]
                    if len({rack[b] for b in kept}) != len(kept): continue
                    if any(load[b] >= hi for b in kept) or (li >= 0 and lead[cur[li][0]] >= lhi): continue   # no room even now
                    obj = (cur[li][1] if li >= 0 else 0) + sum(cur[k][2] for k in fs)
                    mass = (cur[li][3] if li >= 0 else 0) + sum(cur[k][4] for k in fs)     # what the iterate itself keeps of this pattern
                    lst.append((obj, cur[li][0] if li >= 0 else -1, tuple(cur[k][0] for k in fs), mass))
        lst.sort(key=lambda x: (-x[0], -x[3]))      # heaviest first; among equals the one the iterate leans to
        pats.append([x[:3] for x in lst])
    wmax = [0] * (npd + 1)
    for i in range(npd - 1, -1, -1): wmax[i] = wmax[i + 1] + (pats[i][0][0] if pats[i] else 0)
    nodes = [0]; cap = [PAT_MAX_NODES]
    best = dict(obj=-1, rows=None)
    choice = [None] * npd

    def fill():
        # the new slots of every partition: a new leader where the pattern keeps none, then the new followers
        rows = [[c[1]] + list(c[2]) if c[1] >= 0 else [-1] + list(c[2]) for c in choice]
        slots = []
        for i in range(npd):
            if rows[i][0] < 0: slots.append((i, 0))
        for i in range(npd):
            for k in range(len(rows[i]), RF): slots.append((i, k)); rows[i].append(-1)

        def place(si):
            if nodes[0] > cap[0]: return False
            nodes[0] += 1
            if si == len(slots):
                return all(load[b] >= lo for b in short_r) and all(lead[b] >= llo for b in short_l)   # (counts only grow: nobody else can be short)
            # what is left must still be able to lift every broker to its band
            left_l = sum(1 for (i2, k2) in slots[si:] if k2 == 0)
            # (the slots are sorted: leader slots first)
            if sum(max(0, llo - lead[b]) for b in short_l) > left_l: return False
            if sum(max(0, lo - load[b]) for b in short_r) > len(slots) - si: return False
            i, k = slots[si]
            used = [b for b in rows[i] if b >= 0]
            racks = {rack[b] for b in used}
            # (a follower more must leave the broker room for the leaders it is still short of, as in dfs below)
            cands = [b for b in open_b if load[b] < hi and (lead[b] < lhi if k == 0 else llo - lead[b] <= hi - load[b] - 1)
                     and b not in used and rack[b] not in racks]
            cands.sort(key=lambda b: (-(max(0, llo - lead[b]) if k == 0 else 0), -max(0, lo - load[b]), b))
            for b in cands[:12]:
                rows[i][k] = b; load[b] += 1
                if k == 0: lead[b] += 1
                ok = place(si + 1)
                if ok: return True
                rows[i][k] = -1; load[b] -= 1
                if k == 0: lead[b] -= 1
            return False

        short_r = [b for b in range(B) if load[b] < lo]; short_l = [b for b in range(B) if lead[b] < llo]
        open_b = [b for b in range(B) if load[b] < hi]      # the brokers with room when the matching begins (ascending)
        # most constrained first: leader slots, then follower slots, each group by the number of brokers below their band the slot may take
        def n_opts(sl):
            i, k = sl
            used = [b for b in rows[i] if b >= 0]; racks = {rack[b] for b in used}
            pool = short_l if k == 0 else short_r
            return sum(1 for b in pool if b not in used and rack[b] not in racks)
        slots.sort(key=lambda sl: (sl[1] != 0, n_opts(sl)))
        ok = place(0)
        if ok:
            out = [list(r) for r in rows]
            # undo the placements (the caller keeps searching for a heavier choice of patterns)
            for (i, k) in slots:
                b = rows[i][k]; load[b] -= 1
                if k == 0: lead[b] -= 1
            return out
        return None

    def room_bound(i):
        # second bound on what partitions i.. can still add: a broker keeps at most as many of their current replicas as its band has
        # room for, the heaviest ones (the one-leader-per-partition row dropped)
        per = sorted((b, -max(wl if lead[b] < lhi else 0, wf)) for q in range(i, npd) for (b, wl, wf) in items[q])
        tot = 0; last = -1; left = 0
        for (b, nw) in per:
            if b != last: last = b; left = hi - load[b]
            if left > 0: tot -= nw; left -= 1
        return tot

    def dfs(i, obj):
        if nodes[0] > cap[0]: return
        nodes[0] += 1
        if obj + wmax[i] <= best["obj"]: return
        if best["obj"] >= 0 and obj + room_bound(i) <= best["obj"]: return
        if i == npd:
            rows = fill()
            if rows is not None:
                best.update(obj=obj, rows=rows)
                if obj >= target: cap[0] = -1          # nothing heavier to find: every call still open returns at once
            return
        for (o, ld, fs) in pats[i]:
            kept = ([ld] if ld >= 0 else []) + list(fs)
            if any(load[b] >= hi for b in kept) or (ld >= 0 and lead[ld] >= lhi): continue
            for b in kept: load[b] += 1
            if ld >= 0: lead[ld] += 1
            # every leader a broker is still short of takes a replica of its band too: once its followers leave no room for them no
            # completion exists (loads only grow from here)
            if all(llo - lead[b] <= hi - load[b] for b in fs):
                choice[i] = (o, ld, fs)
                dfs(i + 1, obj + o)
            for b in kept: load[b] -= 1
            if ld >= 0: lead[ld] -= 1

    dfs(0, 0)
    if best["rows"] is None: return False
    # the result against the rows of the model it must satisfy: complete rows, distinct brokers and racks, every broker inside its bands
    for i, p in enumerate(pending):
        r = best["rows"][i]
        if len(r) != RF or min(r) < 0 or len(set(r)) != RF or len({rack[b] for b in r}) != RF: return False
    l2 = [int(v) for v in load0]; d2 = [int(v) for v in lead0]
    for r in best["rows"]:
        for b in r: l2[b] += 1
        d2[r[0]] += 1
    if any(l2[b] < lo or l2[b] > hi or d2[b] < llo or d2[b] > lhi for b in range(B)): return False
    for i, p in enumerate(pending): A[p] = best["rows"][i]
    return True


def repair_bands(t: ko.Topic, A) -> int:
    """The repairs after the completion, in place: the broker bands (repair_broker_bands), then the rack bands (repair_racks).  Returns the
    number of moves."""
    return repair_broker_bands(t, A) + repair_racks(t, A)


def repair_broker_bands(t: ko.Topic, A) -> int:
    """What the completion of a half-integral vertex leaves: a few brokers one replica (or one leadership) over their band, as many
    under it (README.md:158-166).  Moves that cost nothing put that right: a follower replica that carries no weight (its broker is
    not a current replica of the partition, README.md:145-146) goes from the lowest over-loaded broker to the lowest under-loaded one
    that is not in the row -- same rack first, another rack only if the partition's count there stays within its band (README.md:
    178-180); a leader whose swap with a follower of its row changes no weight hands over its role.  Partitions in ascending order,
    first fit.  In place; returns the number of moves."""
    B, R, P, RF = t.n_brokers, t.n_racks, t.n_partitions, t.rf
    bd = lp_bands(t)      # the implied ends: same feasible set (kao_round.cpp does the same)
    lo, hi, llo, lhi, phi = bd["rep_lo"], bd["rep_hi"], bd["lead_lo"], bd["lead_hi"], bd["prack_hi"]
    rack = [int(r) for r in np.asarray(t.rack_of)]
    load = [0] * B; lead = [0] * B
    for p in range(P):
        for k in range(RF): load[int(A[p, k])] += 1
        lead[int(A[p, 0])] += 1
    if all(lo <= load[b] <= hi and llo <= lead[b] <= lhi for b in range(B)):
        return 0
    bwv = np.zeros(B, dtype=np.int64) if getattr(t, "broker_w", None) is None else np.asarray(t.broker_w, dtype=np.int64)
    bwlv = np.zeros(B, dtype=np.int64) if getattr(t, "broker_wl", None) is None else np.asarray(t.broker_wl, dtype=np.int64)
    w = t.weights

    def wts(p, b):        # (leader weight, follower weight) of broker b on partition p
        wl = int(bwv[b]) + int(bwlv[b]); wf = int(bwv[b])
        for j in range(t.rf_cur):
            if int(t.current[p, j]) == b:
                wl += w[0 if j == 0 else 1][0]; wf += w[0 if j == 0 else 1][1]
        return wl, wf

    moves = 0
    over = [b for b in range(B) if load[b] > hi]
    if over or any(load[b] < lo for b in range(B)):
        holds = [[] for _ in range(B)]
        srcs = set(over) if over else set()
        for p in range(P):
            for k in range(1, RF):
                if int(A[p, k]) in srcs: holds[int(A[p, k])].append((p, k))
        for b1 in over:
            while load[b1] > hi:
                moved = False
                targets = [b for b in range(B) if load[b] < lo] or [b for b in range(B) if load[b] < hi and b != b1]
                targets.sort(key=lambda b: (rack[b] != rack[b1], b))
                for b2 in targets:
                    for (p, k) in holds[b1]:
                        if int(A[p, k]) != b1: continue
                        row = [int(x) for x in A[p]]
                        if b2 in row or wts(p, b1)[1] != wts(p, b2)[1]: continue
                        if rack[b2] != rack[b1] and sum(1 for x in row if rack[x] == rack[b2]) >= phi: continue
                        A[p, k] = b2; load[b1] -= 1; load[b2] += 1; moves += 1; moved = True
                        break
                    if moved: break
                if not moved: break
        # what is left costs weight: the cheapest follower move of every broker still over its band (first minimal loss in (p, k), target order)
        for b1 in over:
            while load[b1] > hi:
                targets = [b for b in range(B) if load[b] < lo] or [b for b in range(B) if load[b] < hi and b != b1]
                targets.sort(key=lambda b: (rack[b] != rack[b1], b))
                bestm = None
                for (p, k) in holds[b1]:
                    if int(A[p, k]) != b1: continue
                    row = [int(x) for x in A[p]]
                    w1 = wts(p, b1)[1]
                    for b2 in targets:
                        if b2 in row: continue
                        if rack[b2] != rack[b1] and sum(1 for x in row if rack[x] == rack[b2]) >= phi: continue
                        loss = w1 - wts(p, b2)[1]
                        if bestm is None or loss < bestm[0]: bestm = (loss, p, k, b2)
                if bestm is None: break
                _, p, k, b2 = bestm
                A[p, k] = b2; load[b1] -= 1; load[b2] += 1; moves += 1
    for b1 in [b for b in range(B) if lead[b] > lhi]:
        while lead[b1] > lhi:
            moved = False
            cap2 = llo if any(lead[b] < llo for b in range(B)) else lhi      # the taker is below its band when anyone is, else below the upper end
            for p in range(P):
                if int(A[p, 0]) != b1: continue
                for k in range(1, RF):
                    b2 = int(A[p, k])
                    if lead[b2] >= cap2: continue
                    wl1, wf1 = wts(p, b1); wl2, wf2 = wts(p, b2)
                    if wl1 + wf2 != wl2 + wf1: continue
                    A[p, 0], A[p, k] = b2, b1; lead[b1] -= 1; lead[b2] += 1; moves += 1; moved = True
                    break
                if moved: break
            if not moved: break
    for b1 in [b for b in range(B) if lead[b] > lhi]:      # and the cheapest role swap of every broker still leading too many
        while lead[b1] > lhi:
            cap2 = llo if any(lead[b] < llo for b in range(B)) else lhi
            bestm = None
            for p in range(P):
                if int(A[p, 0]) != b1: continue
                wl1, wf1 = wts(p, b1)
                for k in range(1, RF):
                    b2 = int(A[p, k])
                    if lead[b2] >= cap2: continue
                    wl2, wf2 = wts(p, b2)
                    loss = (wl1 + wf2) - (wl2 + wf1)
                    if bestm is None or loss < bestm[0]: bestm = (loss, p, k, b2)
            if bestm is None: break
            _, p, k, b2 = bestm
            A[p, 0], A[p, k] = b2, b1; lead[b1] -= 1; lead[b2] += 1; moves += 1
    # a broker that still leads too many and shares no partition with one that may take a leadership (2,000 brokers: the usual case):
    # a CHAIN of role swaps, breadth first over "u leads p, v follows in p" (partitions ascending, slots ascending), weight-neutral swaps
    # only in the first attempt, any swap in the second; every broker between the ends keeps its count
    for b1 in [b for b in range(B) if lead[b] > lhi]:
        while lead[b1] > lhi:
            cap2 = llo if any(lead[b] < llo for b in range(B)) else lhi
            path = None
            for neutral_only in (True, False):
                leads_of = [[] for _ in range(B)]
                for p in range(P): leads_of[int(A[p, 0])].append(p)
                parent = {b1: None}; queue = [b1]; end = None; qi = 0
                while qi < len(queue) and end is None:
                    u = queue[qi]; qi += 1
                    for p in leads_of[u]:
                        wl1, wf1 = wts(p, u)
                        for k in range(1, RF):
                            v = int(A[p, k])
                            if v in parent: continue
                            if neutral_only:
                                wl2, wf2 = wts(p, v)
                                if wl1 + wf2 != wl2 + wf1: continue
                            parent[v] = (u, p, k); queue.append(v)
                            if lead[v] < cap2: end = v; break
                        if end is not None: break
                if end is not None:
                    path = []
                    v = end
                    while parent[v] is not None:
                        path.append((parent[v][0], parent[v][1], parent[v][2], v)); v = parent[v][0]
                    break
            if path is None: break
            for (u, p, k, v) in path:
                A[p, 0], A[p, k] = v, u; moves += 1
            lead[b1] -= 1; lead[path[0][3]] += 1
    return moves


RACK_REPAIR_MAX = 32     # replicas the racks may be off for repair_racks to try at all


def repair_racks(t: ko.Topic, A) -> int:
    """With a rigid rack band and loose broker bands the completion may leave one rack a replica over its band and another one under it
    (README.md:173-176).  A follower replica moves from a rack over (else: above the lower end of) its band to one under (else: below the
    upper end of) it, within the broker bands (README.md:158-166) and the partition's per-rack band (README.md:178-180).  Rack pairs (source,
    target) ascending; the first pair that offers a move at all decides: its first move that loses no weight (partition, slot, target broker
    ascending), else its cheapest.  Nothing is tried when the racks are more than RACK_REPAIR_MAX replicas off.  In place; returns the number
    of moves.  (kao_round.cpp, the last block of lp_round_assignment.)"""
    B, R, P, RF = t.n_brokers, t.n_racks, t.n_partitions, t.rf
    bd = lp_bands(t)      # the implied ends: same feasible set (kao_round.cpp does the same)
    lo, hi, rlo, rhi, plo, phi = bd["rep_lo"], bd["rep_hi"], bd["rack_lo"], bd["rack_hi"], bd["prack_lo"], bd["prack_hi"]
    rack = [int(r) for r in np.asarray(t.rack_of)]
    load = [0] * B; tot = [0] * R
    for p in range(P):
        for k in range(RF):
            b = int(A[p, k]); load[b] += 1; tot[rack[b]] += 1
    off = sum(max(0, tot[r] - rhi) + max(0, rlo - tot[r]) for r in range(R))
    if off == 0 or off > RACK_REPAIR_MAX:
        return 0
    bwv = np.zeros(B, dtype=np.int64) if getattr(t, "broker_w", None) is None else np.asarray(t.broker_w, dtype=np.int64)
    w = t.weights
    members = [[b for b in range(B) if rack[b] == r] for r in range(R)]

    def wf_of(p, b):
        wf = int(bwv[b])
        for j in range(t.rf_cur):
            if int(t.current[p, j]) == b: wf += w[0 if j == 0 else 1][1]
        return wf

    moves = 0
    for _ in range(2 * RACK_REPAIR_MAX):
        any_over = any(tot[r] > rhi for r in range(R)); any_under = any(tot[r] < rlo for r in range(R))
        if not any_over and not any_under: break
        src = [r for r in range(R) if (tot[r] > rhi if any_over else tot[r] > rlo)]
        dst = [r for r in range(R) if (tot[r] < rlo if any_under else tot[r] < rhi)]
        best = None; neutral = False
        for r1 in src:
            for r2 in dst:
                if r2 == r1: continue
                for p in range(P):
                    row = [int(x) for x in A[p]]
                    c1 = sum(1 for x in row if rack[x] == r1); c2 = sum(1 for x in row if rack[x] == r2)
                    if c1 == 0 or c1 - 1 < plo or c2 >= phi: continue
                    for k in range(1, RF):
                        b1 = row[k]
                        if rack[b1] != r1 or load[b1] - 1 < lo: continue
                        w1 = wf_of(p, b1)
                        for b2 in members[r2]:
                            if load[b2] + 1 > hi or b2 in row: continue
                            loss = w1 - wf_of(p, b2)
                            if best is None or loss < best[0]: best = (loss, p, k, b2)
                            if loss <= 0: neutral = True; break
                        if neutral: break
                    if neutral: break
                if best is not None: break
            if best is not None: break
        if best is None: break
        _, p, k, b2 = best
        b1 = int(A[p, k]); A[p, k] = b2
        load[b1] -= 1; load[b2] += 1; tot[rack[b1]] -= 1; tot[rack[b2]] += 1; moves += 1
    return moves
