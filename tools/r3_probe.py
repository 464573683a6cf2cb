"""GPU probe of round 3 (test tooling): the drifted families under the deterministic and the wall-clock schedule, a determinism
check of kao_solve, per-launch timing traces for the schedule constants, and incumbents saved for the offline KAO-CX analysis.
Usage: r3_probe.py <what>[,<what>...] [budget_s]   what: family | scale | seeds | goldens | determinism | trace | dump
       r3_probe.py solve B R P drift_seed solver_seeds(csv) [budget_s]     one line per solver seed (KAO_* test hooks from the environment)
       r3_probe.py onetrace B R P drift_seed [budget_s] [solver_seed]      one solve under KAO_SOLVE_TRACE=1 (per-launch lines on stderr)
Environment: R3_SCHEDS (schedules, default "0,1"), R3_SEEDS (solver seeds of `family`), R3_HARD=1 (second half of the family).
(Replaces the one-off probes of rounds 2-3: cx_ab, cx_long, cx_seeds, cx_sweep, drift_family, drift_probe, dump_incumbent, one_big, ...)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import synthetic as sy

what = set((sys.argv[1] if len(sys.argv) > 1 else "family").split(","))
if "onetrace" in what:
    os.environ["KAO_SOLVE_TRACE"] = "1"
budget = float(sys.argv[2]) if len(sys.argv) > 2 and not (what & {"solve", "onetrace"}) else 3.0
SCHEDS = [int(v) for v in os.environ.get("R3_SCHEDS", "0,1").split(",")]
kao.init(0)
OUT = "gpurun_out"
os.makedirs(OUT, exist_ok=True)
FAMILY = [(50, 5, 1000), (100, 5, 1000), (120, 4, 1200), (150, 6, 1500), (90, 3, 1500), (200, 8, 1600), (200, 5, 2000), (250, 10, 2000),
          (300, 6, 2000), (160, 4, 2400), (400, 8, 3000), (250, 5, 4000)]
if os.environ.get("R3_HARD"):   # the shapes one 3-s solve does not always prove
    FAMILY = FAMILY[5:]
SCALE = [(100, 5, 1000), (300, 6, 2000), (400, 8, 3000), (500, 10, 5000), (500, 10, 10000), (1000, 20, 30000)]


def topic(B, R, P, dseed=1):
    return sy.drift(sy.make_cluster(B, R, 1, P, 3, [], []), 0.2, dseed)[0]


def run(t, **kw):
    t0 = time.perf_counter()
    r = kao.solve([t], stop_at_bound=1, **kw)[0]
    dt = time.perf_counter() - t0
    return r, dt, kao.last_solve_timing()


def line(tag, B, R, P, dseed, r, dt, tm):
    gap = r.upper_bound - r.objective
    return (f"{tag} B={B:4d} R={R:2d} P={P:5d} d{dseed}: {r.status:15s} obj {r.objective} cert {r.upper_bound} gap {gap} {dt:.2f}s "
            f"t_best {tm['time_to_best']:.2f} launches {tm['launches']} its {tm['search_iters']} K-bound {tm['bound_launches']}/{tm['bound_iters']} cx {tm['cx_calls']}/{tm['cx_gains']} (+{tm.get('cx_further_starts', 0)} starts) gens {tm['generations']}")


if "family" in what:
    for sched in SCHEDS:
        for sd in [int(v) for v in os.environ.get("R3_SEEDS", "3").split(",")]:   # solver seeds
            proven, gaps = 0, []
            for (B, R, P) in FAMILY:
                for dseed in (1, 2):
                    r, dt, tm = run(topic(B, R, P, dseed), seed=sd, time_limit_s=budget, schedule=sched)
                    proven += r.status == "OPTIMAL_PROVEN"
                    gaps.append(r.upper_bound - r.objective)
                    print(line(f"family sched{sched} seed {sd}", B, R, P, dseed, r, dt, tm), flush=True)
            print(f"family sched{sched} seed {sd}: proven {proven}/{len(gaps)}; gaps of the others {sorted(g for g in gaps if g)}", flush=True)

if "scale" in what:
    for sched in SCHEDS:
        for (B, R, P) in SCALE:
            r, dt, tm = run(topic(B, R, P), seed=3, time_limit_s=budget, schedule=sched)
            print(line(f"scale sched{sched}", B, R, P, 1, r, dt, tm), flush=True)

if "seeds" in what:
    # would independent re-runs (another seed) inside the same budget prove more topics than one long run?
    n_any = n_long = 0
    for (B, R, P) in FAMILY:
        for dseed in (1, 2):
            t = topic(B, R, P, dseed)
            got = []
            for sd in (3, 4, 5):
                r, dt, tm = run(t, seed=sd, time_limit_s=budget / 3.0, schedule=0)
                got.append((r.status == "OPTIMAL_PROVEN", r.objective, r.upper_bound, round(dt, 2)))
            n_any += any(g[0] for g in got)
            print(f"seeds B={B:4d} R={R:2d} P={P:5d} d{dseed}: three runs of {budget / 3.0:.2f}s: {got}", flush=True)
    print(f"seeds: proven by any of three short runs {n_any}/24", flush=True)

if "goldens" in what:
    # the rows of tests/golden/drift_scale.json the GPU tests assert: time to the proof over three seeds
    for (B, R, P, lim) in ((100, 5, 1000, 8.0), (200, 5, 2000, 8.0), (300, 6, 2000, 8.0), (400, 8, 3000, 3.0), (250, 5, 4000, 3.0)):
        for sd in (1, 2, 3):
            r, dt, tm = run(topic(B, R, P), seed=sd, time_limit_s=lim, schedule=0)
            print(line(f"goldens seed {sd} gens {tm['generations']}", B, R, P, 1, r, dt, tm), flush=True)

if "determinism" in what:
    for (B, R, P, ml) in ((100, 5, 1000, 400), (300, 6, 2000, 300), (500, 10, 5000, 120)):
        t = topic(B, R, P)
        outs = []
        for rep in range(3):
            r, dt, tm = run(t, seed=7, time_limit_s=60.0, max_launches=ml, schedule=0)
            outs.append((r.objective, r.upper_bound, r.status, r.assignment.tobytes(), tm['launches'], tm['bound_iters'], tm['cx_calls']))
            print(f"determinism B={B} P={P} rep {rep}: obj {r.objective} cert {r.upper_bound} {r.status} launches {tm['launches']} K-bound its {tm['bound_iters']} cx {tm['cx_calls']} {dt:.2f}s", flush=True)
        print(f"determinism B={B} P={P}: identical = {all(o == outs[0] for o in outs)}", flush=True)

if "trace" in what:
    # per-launch wall times of the wall-clock schedule (KAO_SOLVE_TRACE=1 prints to stderr): the constants of the deterministic one
    os.environ["KAO_SOLVE_TRACE"] = "1"
    for (B, R, P) in [(100, 5, 1000), (300, 6, 2000), (500, 10, 5000), (500, 10, 10000), (1000, 20, 30000)]:
        print(f"trace B={B} P={P}", file=sys.stderr, flush=True)
        r, dt, tm = run(topic(B, R, P), seed=3, time_limit_s=1.0, schedule=SCHEDS[0])
        print(line("trace", B, R, P, 1, r, dt, tm), flush=True)
    ts = sy.drift(sy.make_config(4), 0.2, 1)
    print("trace cfg4 drifted batch", file=sys.stderr, flush=True)
    t0 = time.perf_counter(); rs = kao.solve(ts, seed=3, time_limit_s=2.0, schedule=SCHEDS[0]); dt = time.perf_counter() - t0
    print(f"trace cfg4 batch: proven {sum(r.status == 'OPTIMAL_PROVEN' for r in rs)}/200 in {dt:.3f}s {kao.last_solve_timing()}", flush=True)
    del os.environ["KAO_SOLVE_TRACE"]

if "dump" in what:
    os.makedirs(f"{OUT}/incumbents", exist_ok=True)
    for (B, R, P, dseed) in [(120, 4, 1200, 1), (120, 4, 1200, 2), (100, 5, 1000, 1), (100, 5, 1000, 2), (300, 6, 2000, 1), (200, 5, 2000, 1), (200, 5, 2000, 2), (250, 10, 2000, 1),
                             (250, 10, 2000, 2), (150, 6, 1500, 1), (150, 6, 1500, 2), (90, 3, 1500, 1), (90, 3, 1500, 2), (200, 8, 1600, 1), (200, 8, 1600, 2), (160, 4, 2400, 1),
                             (160, 4, 2400, 2), (400, 8, 3000, 1), (400, 8, 3000, 2), (250, 5, 4000, 1), (250, 5, 4000, 2), (50, 5, 1000, 1), (50, 5, 1000, 2), (300, 6, 2000, 2),
                             (500, 10, 5000, 1)]:
        t = topic(B, R, P, dseed)
        r, dt, tm = run(t, seed=3, time_limit_s=budget, schedule=0)
        print(line("dump", B, R, P, dseed, r, dt, tm), flush=True)
        np.savez_compressed(f"{OUT}/incumbents/inc_{B}_{R}_{P}_d{dseed}.npz", assignment=r.assignment, objective=r.objective, upper_bound=r.upper_bound,
                            current=np.asarray(t.current), rack_of=np.asarray(t.rack_of))

if "solve" in what or "onetrace" in what:
    B, R, P, d = (int(v) for v in sys.argv[2:6])
    t = topic(B, R, P, d)
    if "solve" in what:
        seeds = [int(v) for v in sys.argv[6].split(",")]
        lim = float(sys.argv[7]) if len(sys.argv) > 7 else 3.0
    else:
        lim = float(sys.argv[6]) if len(sys.argv) > 6 else 3.0
        seeds = [int(sys.argv[7]) if len(sys.argv) > 7 else 3]
    for sd in seeds:
        r, dt, tm = run(t, seed=sd, time_limit_s=lim, schedule=SCHEDS[0])
        print(line(f"solve seed {sd}", B, R, P, d, r, dt, tm), flush=True)
