"""GPU: KAO-CX against oracle/kao_cycle.py on drifted topics (matrices, seed table, whole rounds) -- test tooling."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import synthetic as sy
import kao_oracle as ko, kao_cycle as kc, kao_port as kp
kp.build()
kao.init(0)
B, R, P = (int(v) for v in sys.argv[1:4])
launches = int(sys.argv[4]) if len(sys.argv) > 4 else 20
pt = sy.drift(sy.make_cluster(B, R, 1, P, 3, [], []), 0.2, 1)[0]
t = ko.Topic(name=pt.name, broker_ids=np.array(pt.broker_ids), rack_of=np.array(pt.rack_of), n_racks=pt.n_racks,
             n_partitions=pt.n_partitions, rf=pt.rf, current=np.array(pt.current), weights=pt.weights)
A = kp.port_search(t, 3, 0, launches, 512)["best"]
print("start", kc.evaluate(t, A))
rd = kc.Round(t, A)
ok = True
for layer, (Ds, Ms, E) in enumerate(((rd.DF, rd.MF, rd.EF), (rd.DS, rd.MS, rd.ES), (rd.DL, rd.ML, rd.EL))):
    for lev in range(4):
        d, m, s = kao.cycle_matrices(pt, A, layer, lev)
        same = np.array_equal(d, Ds[lev])
        if lev:
            same &= np.array_equal(m, Ms[lev])
        else:
            es = (E & np.uint64(0xFFFFFFFF)).astype(np.uint32)
            same &= np.array_equal(s[:B, :B], es[:B, :B])
        print("layer", layer, "level", lev, "equal", bool(same))
        ok &= bool(same)
if not rd.cycle_candidates():
    tab = kao.cycle_seeds(pt, A)
    same = np.array_equal(tab, rd.seed_table())
    print("seed table equal", same); ok &= same
t0 = time.time(); Xo, hist = kc.improve(t, A, 3); t1 = time.time()
Xg = np.array(A, dtype=np.uint16).copy()
for _ in range(3):
    Xn, obj, st = kao.improve_cycles(pt, Xg, 1)
    if st["improving_rounds"] == 0: break
    Xg = Xn
t2 = time.time()
print("oracle", [h.get("objective") for h in hist], "%.1fs" % (t1 - t0), "device", kc.evaluate(t, Xg), "%.2fs" % (t2 - t1))
same = np.array_equal(Xo.reshape(-1), Xg.reshape(-1)); print("3 rounds equal", same); ok &= same
t0 = time.time(); Xf, obj, st = kao.improve_cycles(pt, A, 0); print("device to fixpoint", obj, st, "%.2fs" % (time.time() - t0))
print("ALL OK" if ok else "MISMATCH")
