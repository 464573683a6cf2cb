"""CPU prototype: Mehrotra predictor-corrector on the compact LP (generic sparse algebra) -- how many iterations, how good
are the un-crossed-over duals as K-bound multipliers.  Test tooling."""
import os, sys, time
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spl
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from reduced_lp import build, otopic, evalL, sy

def standard_form(t):
    """min c x, A x = b, 0 <= x (<= u)."""
    A, c, lb, ub, rlo, rhi, idx = build(t)
    nr, nv = A.shape
    eq = rlo == rhi
    # slack columns for non-equality rows: A x + s = rhi, 0 <= s <= rhi - rlo
    sl = np.nonzero(~eq)[0]
    S = sp.csr_matrix((np.ones(len(sl)), (sl, np.arange(len(sl)))), shape=(nr, len(sl)))
    A2 = sp.hstack([A, S], format="csr")
    c2 = np.concatenate([-c, np.zeros(len(sl))])
    u2 = np.concatenate([ub, rhi[sl] - rlo[sl]])
    return A2, rhi.copy(), c2, u2, idx

def ipm(A, b, c, u, tol=1e-8, maxit=100, verbose=True):
    m, n = A.shape
    U = np.isfinite(u); uu = np.where(U, u, 0.0)
    AT = A.T.tocsr()
    def solve_normal(theta, rhs):
        M = (A @ sp.diags(theta) @ AT).tocsc() + 1e-10 * sp.identity(m, format="csc")
        lu = spl.splu(M)
        return lu
    # starting point (Mehrotra)
    lu = solve_normal(np.ones(n), None)
    x = AT @ lu.solve(b)
    y = lu.solve(A @ c); s = c - AT @ y
    x = np.maximum(x, 1.0); w = np.where(U, np.maximum(uu - x, 1.0), 1.0)
    x = np.where(U, np.minimum(x, np.maximum(uu * 0.5, 1e-2)), x); w = np.where(U, uu - x, 1.0)
    s = np.maximum(s, 1.0); v = np.where(U, 1.0, 0.0)
    for it in range(maxit):
        rp = b - A @ x
        rd = c - AT @ y - s + v
        ru = np.where(U, uu - x - w, 0.0)
        mu = (x @ s + (w * v)[U].sum()) / (n + U.sum())
        pobj = c @ x; dobj = b @ y - (uu * v)[U].sum()
        rel = abs(pobj - dobj) / (1 + abs(pobj))
        pinf = np.linalg.norm(rp) / (1 + np.linalg.norm(b)); dinf = np.linalg.norm(rd) / (1 + np.linalg.norm(c))
        if verbose: print(f"  it {it:3d} pobj {pobj:.6f} dobj {dobj:.6f} mu {mu:.3e} pinf {pinf:.2e} dinf {dinf:.2e}", flush=True)
        if rel < tol and pinf < tol and dinf < tol: break
        theta = 1.0 / (s / x + np.where(U, v / np.where(U, w, 1.0), 0.0))
        lu = solve_normal(theta, None)
        def direction(rxs, rwv):
            # x s = rxs target residual (i.e. s dx + x ds = rxs), w v similarly
            h = rd - rxs / x + np.where(U, (rwv - v * ru) / np.where(U, w, 1.0), 0.0)
            dy = lu.solve(rp + A @ (theta * h))
            dx = theta * (AT @ dy - h)
            ds = (rxs - s * dx) / x
            dw = np.where(U, ru - dx, 0.0)
            dv = np.where(U, (rwv - v * dw) / np.where(U, w, 1.0), 0.0)
            return dx, dy, ds, dw, dv
        def maxstep(z, dz, mask=None):
            neg = dz < 0
            if mask is not None: neg &= mask
            return min(1.0, (-z[neg] / dz[neg]).min()) if neg.any() else 1.0
        dx, dy, ds, dw, dv = direction(-x * s, np.where(U, -w * v, 0.0))
        ap = min(maxstep(x, dx), maxstep(w, dw, U)); ad = min(maxstep(s, ds), maxstep(v, dv, U))
        mu_aff = ((x + ap * dx) @ (s + ad * ds) + ((w + ap * dw) * (v + ad * dv))[U].sum()) / (n + U.sum())
        sigma = (mu_aff / mu) ** 3
        dx, dy, ds, dw, dv = direction(sigma * mu - x * s - dx * ds, np.where(U, sigma * mu - w * v - dw * dv, 0.0))
        ap = min(maxstep(x, dx), maxstep(w, dw, U)); ad = min(maxstep(s, ds), maxstep(v, dv, U))
        ap = min(1.0, 0.9995 * ap) if ap < 1 else 1.0; ad = min(1.0, 0.9995 * ad) if ad < 1 else 1.0
        x += ap * dx; w += ap * dw; y += ad * dy; s += ad * ds; v += ad * dv
    return x, y, s, it

if __name__ == "__main__":
    B, R, P = (int(a) for a in sys.argv[1:4])
    t = otopic(sy.drift(sy.make_cluster(B, R, 1, P, 3, [], []), 0.2, 1)[0])
    A, b, c, u, (C3, C4, C6) = standard_form(t)
    print(A.shape, "ub vars", np.isfinite(u).sum())
    t0 = time.time()
    x, y, s, it = ipm(A, b, c, u, tol=float(sys.argv[4]) if len(sys.argv) > 4 else 1e-8)
    print(f"{it} iterations, {time.time() - t0:.1f}s, value {-(c @ x):.6f}")
    for sgn in (1.0, -1.0):
        a = np.round(sgn * y[C3] * 65536).astype(np.int64); l = np.round(sgn * y[C4] * 65536).astype(np.int64); g = np.round(sgn * y[C6] * 65536).astype(np.int64)
        print(f"  sign {sgn:+.0f}: exact L = {evalL(t, a, l, g):.4f}")
