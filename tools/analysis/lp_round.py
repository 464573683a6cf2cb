"""CPU (round 5): a vertex of the compact LP (HiGHS dual simplex) rounded into an assignment by the specification
(oracle/kao_lp.py round_primal) -- the experiment that showed the LP's vertices are integral and attain the optimum.
Usage: lp_round.py B R P [drift_seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kao_oracle as ko
import kao_lp as klp

if __name__ == "__main__":
    from regret import topic
    B, R, P = [int(x) for x in sys.argv[1:4]]
    seed = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    t = topic(B, R, P, seed=seed)
    lp = klp.build(t)
    t0 = time.time()
    val, _, x, _ = klp.solve_highs(lp, method="highs-ds")
    print(f"{B}x{P} d{seed}: simplex vertex value {val:.4f} in {time.time() - t0:.1f} s, fractional variables {int((np.abs(x - np.rint(x)) > 1e-6).sum())} of {len(x)}", flush=True)
    A, rep = klp.round_primal(t, *klp.blocks_from_compact(t, x))
    obj, viol = ko.verify(t, A)
    print(rep, "objective", obj, "violations", [int(v) for v in np.asarray(viol)])
