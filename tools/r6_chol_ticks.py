"""GPU (round 6): clock ticks of the diagonal tile's phases (kao_chol.hip, KAO_CHOL_DEBUG) + accuracy and time of the dense kernels at 64 and 2112 rows."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["KAO_CHOL_DEBUG"] = "1"
import numpy as np
import kafka_assignment_optimizer_amd as kao
kao.init(0)
rng = np.random.default_rng(1)
for n in (64, 2112):
    G = rng.standard_normal((n, n + 32)); sc = 10.0 ** rng.uniform(-1.5, 1.5, n)
    A = (G @ G.T) * np.outer(sc, sc) + 1e-6 * np.diag(sc * sc); A = (A + A.T) / 2
    rhs = rng.standard_normal(n)
    for _ in range(2):
        d = kao.dense_spd_test(A, rhs)
    L = np.linalg.cholesky(A); x = np.linalg.solve(A, rhs)
    print(f"n={n}: |L-L_np|/|L| {np.abs(np.tril(d['factor']) - L).max() / np.abs(L).max():.2e} |x-x_np|/|x| {np.abs(d['x'] - x).max() / np.abs(x).max():.2e} factor {d['ms_factor']*1e3:.0f} us solve {d['ms_solve']*1e3:.0f} us", flush=True)
