"""GPU (round 6): KAO-LP's dense kernels (kao_chol.hip) through the kao_dense_spd_test hook against numpy: factor, diagonal-tile
inverses, solution; then the interior point on drifted topics with the rack block / broker rows switched between their variants
(KAO_LP_RACK, KAO_LP_BROKER_U).  (Until commit "remove round 5's dense kernels" the "old" mode also ran round 5's Cholesky and
triangular solves, KAO_LP_DENSE=old: profiles/r06_c01 / r06_c02 hold those A/B runs.)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import synthetic as sy
kao.init(0)
rng = np.random.default_rng(7)
for n in (64, 128, 192, 640, 2112):
    G = rng.standard_normal((n, n + 32))
    sc = 10.0 ** rng.uniform(-1.5, 1.5, n)                      # badly scaled, still SPD
    A = (G @ G.T) * np.outer(sc, sc) + 1e-6 * np.diag(sc * sc)
    A = (A + A.T) / 2
    rhs = rng.standard_normal(n)
    d = kao.dense_spd_test(A, rhs)
    L = np.linalg.cholesky(A)
    Ld = np.tril(d["factor"])
    eL = np.abs(Ld - L).max() / np.abs(L).max()
    nt = n // 64
    eU = max((np.abs(d["factor"][j*64:(j+1)*64, i*64:(i+1)*64] - L[i*64:(i+1)*64, j*64:(j+1)*64].T).max() for i in range(nt) for j in range(i)), default=0.0) / np.abs(L).max()
    eI = max(np.abs(d["linv"][k] @ L[k*64:(k+1)*64, k*64:(k+1)*64] - np.eye(64)).max() for k in range(nt))
    upper0 = max(np.abs(np.triu(d["linv"][k], 1)).max() for k in range(nt))
    x = np.linalg.solve(A, rhs)
    eX = np.abs(d["x"] - x).max() / np.abs(x).max()
    res = np.abs(A @ d["x"] - rhs).max() / np.abs(rhs).max()
    print(f"n={n}: |L-L_np|/|L| {eL:.2e}  upper copy {eU:.2e}  |Linv L - I| {eI:.2e} (above diag {upper0:.1e})  |x-x_np|/|x| {eX:.2e}  residual {res:.2e}  cond {np.linalg.cond(A):.1e}"
          f"  factor {d['ms_factor']*1e3:.0f} us  solve {d['ms_solve']*1e3:.0f} us", flush=True)
# a dependent row: the pivot rule pins it (L_jj = 1e64), the solve leaves that component at ~0
n = 128
G = rng.standard_normal((n, n)); A = G @ G.T + np.eye(n); A[70, :] = A[3, :]; A[:, 70] = A[:, 3]; A[70, 70] = A[3, 3]
d = kao.dense_spd_test(A)
print("dependent row 70: L[70,70] =", d["factor"][70, 70], " x[70] =", d["x"][70], " finite:", bool(np.isfinite(d["x"]).all()), flush=True)

def run(pt, tag):
    kao.lp_trace(pt, max_iters=1)
    out = {}
    for mode in ("old", "new"):
        os.environ["KAO_LP_DENSE"] = mode; os.environ["KAO_LP_RACK"] = mode; os.environ["KAO_LP_BROKER_U"] = "4" if mode == "old" else "8"
        t0 = time.perf_counter(); d = kao.lp_trace(pt, max_iters=200); dt = time.perf_counter() - t0
        out[mode] = d
        print(f"{tag} dense={mode}: status {d['status']} it {d['iterations']} primal {d['primal']:.6f} dual {d['dual']:.6f} ipm {d['ms']:.1f} ms = {d['ms']/max(1,d['iterations']):.3f} ms/it (call {dt*1e3:.0f} ms)", flush=True)
    a, b = out["old"]["trace"], out["new"]["trace"]
    m = min(len(a), len(b)); worst = 0.0
    for i in range(m):
        if a[i][0] > 1e-6:
            worst = max(worst, abs(a[i][0] - b[i][0]) / a[i][0], abs(a[i][1] - b[i][1]) / max(1, abs(a[i][1])), abs(a[i][2] - b[i][2]) / max(1, abs(a[i][2])))
    print(f"   worst relative trace deviation new vs old while mu > 1e-6: {worst:.2e}; multipliers max diff {np.abs(out['old']['a'].astype(np.int64) - out['new']['a']).max()}", flush=True)
    os.environ["KAO_LP_DENSE"] = "new"
    if pt.n_partitions >= 30000:
        for u in ("4", "16"):
            os.environ["KAO_LP_BROKER_U"] = u
            d = kao.lp_trace(pt, max_iters=200)
            print(f"   broker U={u}: it {d['iterations']} ipm {d['ms']:.1f} ms = {d['ms']/max(1,d['iterations']):.3f} ms/it", flush=True)
        os.environ["KAO_LP_BROKER_U"] = "8"; os.environ["KAO_LP_RACK"] = "old"
        d = kao.lp_trace(pt, max_iters=200)
        print(f"   rack block by the LDS-tiled kernel: it {d['iterations']} ipm {d['ms']:.1f} ms = {d['ms']/max(1,d['iterations']):.3f} ms/it", flush=True)
        os.environ["KAO_LP_RACK"] = "new"
    b1 = kao.lp_bound(pt); b2 = kao.lp_bound(pt)
    print(f"   lp_bound x2 (new): certificate {b1['bound']} / {b2['bound']}, same bits: {np.array_equal(b1['a'], b2['a']) and b1['best_dual'] == b2['best_dual']}", flush=True)

for (B, R, P) in ((100, 5, 1000), (130, 5, 1000), (500, 10, 5000)):
    run(sy.drift(sy.make_cluster(B, R, 1, P, 3, [], []), 0.2, 1)[0], f"{B}x{P}")
for name in sys.argv[1:] or ["drift30k", "drift100k"]:
    run(sy.north_star_topic(name), name)
