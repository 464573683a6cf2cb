"""CPU analysis (test tooling): the cheapest improving change of a stuck incumbent, by local branching with HiGHS around it.
Loads an incumbent saved by tools/r3_probe.py dump, maximises the README objective subject to the model rows plus
"at most K incumbent slots change", with the candidate brokers of a partition restricted (mode) so that the MILP stays small:
  tight  : incumbent row + current row of the partition
  zero   : ... + every broker for slots whose incumbent replica earns no weight (free rebalancing moves)
  full   : every broker
Prints the changed partitions (old row -> new row, weights)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "oracle"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import kao_oracle as ko
from scipy.optimize import Bounds, LinearConstraint, milp

path, K, mode = sys.argv[1], int(sys.argv[2]), (sys.argv[3] if len(sys.argv) > 3 else "tight")
tl = float(sys.argv[4]) if len(sys.argv) > 4 else 900.0
d = np.load(path)
inc = d["assignment"].astype(np.int64)
cur = d["current"]; rack = d["rack_of"]
P, RF = inc.shape; B = len(rack); R = int(rack.max()) + 1
t = ko.Topic(name="t", broker_ids=np.arange(B, dtype=np.int32), rack_of=rack.astype(np.uint8), n_racks=R, n_partitions=P, rf=RF, current=cur.astype(np.uint16))
obj0, viol = ko.verify(t, inc.astype(np.uint16))
print(f"instance B={B} R={R} P={P}: incumbent {obj0} (viol {viol[0]}), certificate {int(d['upper_bound'])}", flush=True)
A, lo, hi = ko._sparse_model(t)
c = ko.objective_vector(t).astype(float)
n = 2 * B * P
ub = np.zeros(n)
W = t.weight_matrix()
for p in range(P):
    allowed = set(int(b) for b in inc[p]) | set(int(b) for b in cur[p] if b < B)
    if mode == "full" or (mode == "zero" and any(W[p, int(b), 0 if k == 0 else 1] == 0 for k, b in enumerate(inc[p]))):
        allowed = set(range(B))
    for b in allowed:
        ub[ko.var_index(t, b, p, False)] = 1; ub[ko.var_index(t, b, p, True)] = 1
x0 = np.zeros(n)
for p in range(P):
    for k, b in enumerate(inc[p]):
        x0[ko.var_index(t, int(b), p, k == 0)] = 1
cons = [LinearConstraint(A, lo, hi), LinearConstraint(x0.reshape(1, -1), P * RF - K, np.inf)]
print(f"free variables {int(ub.sum())} of {n}; local branching radius {K}", flush=True)
t0 = time.perf_counter()
cobj = c * 1000.0 + x0 if os.environ.get('LB_MINCHANGE') else c
res = milp(-cobj, constraints=cons, integrality=np.ones(n), bounds=Bounds(0, ub), options={"time_limit": tl, "mip_rel_gap": 0.0, "disp": False})
print(f"HiGHS status {res.status} in {time.perf_counter() - t0:.1f}s objective {None if res.x is None else int(round(float(c @ np.rint(res.x))))}", flush=True)
if res.x is not None:
    new = ko.decode_solution(t, res.x).astype(np.int64)
    ch = [p for p in range(P) if sorted(new[p][1:]) != sorted(inc[p][1:]) or new[p][0] != inc[p][0]]
    print(f"{len(ch)} partitions change")
    c_rep = np.bincount(inc.reshape(-1), minlength=B); c_lead = np.bincount(inc[:, 0], minlength=B)
    bd = t.bounds()
    for p in ch:
        wo = [int(W[p, int(b), 0 if k == 0 else 1]) for k, b in enumerate(inc[p])]
        wn = [int(W[p, int(b), 0 if k == 0 else 1]) for k, b in enumerate(new[p])]
        print(f"  p{p}: {inc[p].tolist()} w{wo} -> {new[p].tolist()} w{wn}  (current {cur[p].tolist()}) racks {rack[inc[p]].tolist()} -> {rack[new[p]].tolist()}  gain {sum(wn) - sum(wo)}")
    print("bands", bd)
    np.save(path.replace(".npz", f"_lb{K}_{mode}.npy"), new)
