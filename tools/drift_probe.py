"""GPU: drifted single topics (tools/drift_scale.py sizes) with the search-price feedback and the elite launches switched
on and off / varied -- what each buys inside a fixed budget (test tooling)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import synthetic as sy
kao.init(0)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
sizes = {"a": (100, 5, 1000), "b": (300, 6, 2000), "c": (500, 10, 5000), "d": (500, 10, 10000), "e": (1000, 20, 30000)}
pick = sys.argv[2] if len(sys.argv) > 2 else "abc"
names = sys.argv[3].split(",") if len(sys.argv) > 3 else None
def auto_ep(P, rf=3, iters=128):
    lg = min(16, max(8, (2 * P * rf).bit_length() - 1))
    return max(1, (1 << lg) // iters)
for key in pick:
    B, R, P = sizes[key]
    ep = auto_ep(P)
    variants = {
        "base": (dict(use_prices=-1, elite_period=-1), {}),
        "elite": (dict(use_prices=-1), {}),
        "elite/4": (dict(use_prices=-1, elite_period=max(1, ep // 4)), {}),
        "elite/16": (dict(use_prices=-1, elite_period=max(1, ep // 16)), {}),
        "both-rec": (dict(), {"KAO_X_PRICE_SRC": "1"}),
        "both-last": (dict(), {"KAO_X_PRICE_SRC": "2"}),
        "both-rec/4": (dict(elite_period=max(1, ep // 4)), {"KAO_X_PRICE_SRC": "1"}),
        "r128": (dict(restarts=128), {}), "r256": (dict(restarts=256), {}), "r512": (dict(restarts=512), {}),
        "r1024": (dict(restarts=1024), {}), "r2048": (dict(restarts=2048), {}),
        "i512": (dict(iters_per_launch=512), {}), "r512i512": (dict(restarts=512, iters_per_launch=512), {}),
        "default": (dict(), {}),
    }
    t = sy.drift(sy.make_cluster(B, R, 1, P, 3, [], []), 0.2, 1)[0]
    kao.solve([t], seed=1, max_launches=1)  # warm the arena cache
    for name, (kw, env) in variants.items():
        if names and name not in names:
            continue
        for k in ("KAO_X_PRICE_SRC",):
            os.environ.pop(k, None)
        os.environ.update(env)
        t0 = time.perf_counter()
        r = kao.solve([t], seed=3, stop_at_bound=1, time_limit_s=budget, **kw)[0]
        dt = time.perf_counter() - t0
        tm = kao.last_solve_timing()
        print(f"B={B} P={P} {name:10s}: {r.status} objective {r.objective} certificate {r.upper_bound} gap {r.upper_bound - r.objective} "
              f"t_best {tm['time_to_best']:.2f}s launches {int(tm['launches'])} total {dt:.2f}s", flush=True)
