import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import synthetic
from kafka_assignment_optimizer_amd.solver import decode_key
kao.init(0)
rng = synthetic.SplitMix64(synthetic.CONFIG_SEED + 5)
rm = rng.sample(list(range(1000)), 50); add = [(1000 + i, b % 20) for i, b in enumerate(rm)]  # rack-preserving: stays feasible
P = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
pt = synthetic.make_cluster(1000, 20, 1, P, 3, rm, add, bounds_override={"rep_hi": -(-P*3//1000)+1})[0]
print("ub", kao.upper_bound(pt), kao.derive_bounds(pt))
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 256
s = kao.Session([pt], seed=5, restarts=int(sys.argv[3]) if len(sys.argv) > 3 else 64, iters_per_launch=iters)
t0 = time.perf_counter()
for L in range(40):
    s.step(1); k = s.best_keys()[0]
    st = s.restart_state(0, 0)
    print(L, f"{time.perf_counter()-t0:.3f}s key viol/obj/rho", decode_key(k), "restart0 V", st["V"], "obj", st["obj"], "acc", st["n_accept"], flush=True)
    if L % 8 == 7 or L == 0:
        print("   viol breakdown [total,C1..C7]:", kao.evaluate(pt, st["final"])[1].tolist(), flush=True)
    if decode_key(k)[0] == 0 and decode_key(k)[1] >= kao.upper_bound(pt): break
print(s.stats())
