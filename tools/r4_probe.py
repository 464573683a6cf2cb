"""GPU probe of round 4 (measurement tooling, not product).
  r4_probe.py fill <which> "<restarts,...>" "<waves,...>" [launches]   K-search launch time of a topic that lives in HBM against the
                                                                       number of restarts and restarts per workgroup (KAO_GLOBAL_WAVES)
  r4_probe.py team <which> "<restarts,...>" "<team,...>" [launches]    the same against the team size (KAO_TEAM; 1 = one wavefront per restart)
  r4_probe.py solve <which> "<team,...>" [seconds] [seed]              one kao_solve per team size: objective, certificate, timing
"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import synthetic as sy

kao.init(0)
mode = sys.argv[1]
if mode == "solve":
    which = sys.argv[2]
    t = sy.north_star_topic(which)
    budget = float(sys.argv[4]) if len(sys.argv) > 4 else 3.0
    seed = int(sys.argv[5]) if len(sys.argv) > 5 else 3
    kao.solve([t], seed=1, max_launches=1)          # arenas, code objects
    for team in [int(v) for v in sys.argv[3].split(",")]:
        os.environ["KAO_TEAM"] = str(team)
        t0 = time.perf_counter()
        kw = {"restarts": int(os.environ["R4_RESTARTS"])} if os.environ.get("R4_RESTARTS") else {}
        r = kao.solve([t], seed=seed, stop_at_bound=1, time_limit_s=budget, **kw)[0]
        tm = kao.last_solve_timing()
        print(json.dumps({"which": which, "team": team, "restarts": os.environ.get("R4_RESTARTS", "auto"), "status": str(r.status), "objective": int(r.objective), "certificate": int(r.upper_bound),
                          "gap": int(r.upper_bound - r.objective), "seconds": round(time.perf_counter() - t0, 3),
                          "timing": {k: (round(v, 4) if isinstance(v, float) else v) for k, v in tm.items()}}), flush=True)
if mode in ("fill", "team"):
    which = sys.argv[2]
    launches = int(sys.argv[5]) if len(sys.argv) > 5 else 4
    t = sy.north_star_topic(which)
    sy.north_star_topic = lambda w, _t=t: _t     # one generation of the instance for every row
    for waves in [int(v) for v in sys.argv[4].split(",")]:
        for restarts in [int(v) for v in sys.argv[3].split(",")]:
            os.environ["KAO_TEAM" if mode == "team" else "KAO_GLOBAL_WAVES"] = str(waves)
            st = sy.north_star_steps(kao, which, launches=launches, restarts=restarts)
            it_s = st["iters_per_launch"] / (st["k_search_ms_per_launch"] * 1e-3)
            print(json.dumps({"which": which, "waves": waves, "restarts": st["restarts"], "workgroups": st["k_search_workgroups"],
                              "k_search_ms": round(st["k_search_ms_per_launch"], 3), "k_eval_ms": round(st["k_eval_ms_per_launch"], 3),
                              "neighbours_per_s": st["neighbours_per_launch"] / (st["k_search_ms_per_launch"] * 1e-3),
                              "alg_GBps": st["k_search_algorithmic_bytes_per_launch"] / (st["k_search_ms_per_launch"] * 1e-3) / 1e9,
                              "iters_per_s_per_restart": it_s, "team": waves if mode == "team" else 1, "lds": st["k_search_lds_bytes"], "obj": st["objective_after"], "viol": st["violation_after"]}), flush=True)
