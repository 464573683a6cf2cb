#!/usr/bin/env python3
"""CPU only.  The host half of KAO-LP's rounding on iterates of the scalar restatement (oracle/kao_lp_port.c): the specification
(oracle/kao_lp.py round_primal) with and without the pattern completion of half-integral vertices, and the product's host code
(kao_round.cpp through kao_lp_round_host) against it.  No device involved: the library is only asked for its host entry point.

    python tools/analysis/lp_round_sweep.py                 # nine drifted topics x six salts, kao_lp_round's perturbation, tol 1e-8
    python tools/analysis/lp_round_sweep.py --loose         # five topics x three salts x tolerances 1e-4 / 1e-5 / 1e-6

Columns: fractional partitions (after rows outside the inflows joined them), objective minus certificate and violated rows without
patterns (search over candidate rows) and with them, product == specification, time of the product's host half.
docs/notes_r05.md section 6 and profiles/r05_cpu_pattern_completion.txt quote this output."""
import argparse
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import kao_oracle as ko          # noqa: E402
import kao_lp as kl              # noqa: E402
from regret import topic         # noqa: E402


def pack(F, L, YF, YL, ZF, ZL):
    """the device's layout of the quantised iterate (k_lp_round): [2*NJ + 2*R][P] centi-units, then the inflows"""
    q = np.concatenate([np.asarray(F).T, np.asarray(L).T, np.asarray(YF).T, np.asarray(YL).T], axis=0).astype(np.uint8)
    return np.ascontiguousarray(q), np.concatenate([np.asarray(ZF), np.asarray(ZL)]).astype(np.int32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--loose", action="store_true")
    a = ap.parse_args()
    import kafka_assignment_optimizer_amd as kao
    from conftest import to_product_topic
    if a.loose:
        topics = [(300, 10, 2000, 1), (200, 8, 3000, 4), (400, 8, 6000, 2), (100, 5, 1000, 1), (60, 6, 400, 2)]
        salts, tols = (0, 1, 2), (1e-4, 1e-5, 1e-6)
    else:
        topics = [(30, 5, 200, 3), (60, 6, 400, 1), (60, 6, 400, 2), (100, 10, 1000, 2), (100, 10, 1000, 3), (300, 10, 2000, 1),
                  (300, 10, 2000, 2), (500, 10, 5000, 1), (200, 8, 3000, 4)]
        salts, tols = (0, 1, 2, 3, 4, 5), (1e-8,)
    parts = kl.PAT_MAX_PARTS
    n = bad = 0; at_cert = [0, 0]; feasible = [0, 0]
    print("B P dseed salt tol | fractional | candidate rows: obj-cert viol | patterns: obj-cert viol | product==spec ms")
    for (B, R, P, ds) in topics:
        t = topic(B, R, P, seed=ds); pt = to_product_topic(t)
        r0 = kl.port_solve(t)
        cert = math.floor(kl.exact_dual_value(t, r0["a"], r0["l"], r0["g"]) + 1e-9)
        for salt in salts:
            for tol in tols:
                r = kl.port_solve(t, tol=tol, maxit=200, primal=True, pert=min(1e-2, 100.0 / (P * t.rf)), salt=salt)
                blocks = kl.primal_blocks(t, r["x"], r["xg"])
                res = []
                for k, mp in enumerate((0, parts)):
                    kl.PAT_MAX_PARTS = mp
                    A, rep = kl.round_primal(t, *blocks)
                    obj, viol = ko.verify(t, A)
                    v = int(np.asarray(viol).sum())
                    res.append((obj - cert, v)); feasible[k] += v == 0; at_cert[k] += v == 0 and obj == cert
                t0 = time.time(); d = kao.lp_round_host(pt, *pack(*blocks)); ms = (time.time() - t0) * 1e3
                same = d["assignment"].tolist() == A.tolist() and (d["fractional"], d["over_inflow"]) == (rep["fractional"], rep["over_inflow"])
                n += 1; bad += not same
                if rep["fractional"]:
                    print(B, P, ds, salt, tol, "|", rep["fractional"], "|", *res[0], "|", *res[1], "|", same, round(ms, 1), flush=True)
    print("iterates", n, "feasible without / with patterns", feasible, "at the certificate", at_cert, "product != specification", bad)


if __name__ == "__main__":
    main()
