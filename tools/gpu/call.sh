#!/bin/bash
# round 5, call 50: the huge instances with the two-phase band repair; LP tests
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
T=r05_c50
timeout 900 python -m pytest tests/test_gpu_lp.py -m gpu -q -s 2>&1 | grep -E "passed|failed|FAILED|golden families"
timeout 900 python - > gpurun_out/${T}_huge.log 2>&1 <<'P'
import sys, time, os
sys.path.insert(0, '.')
import kafka_assignment_optimizer_amd as kao
from kafka_assignment_optimizer_amd import synthetic as sy
kao.init(0)
cases = [('1000x100000 drift 0.2 seed 1', lambda: sy.north_star_topic('drift100k')),
         ('1000x100000 drift 0.2 seed 2', lambda: sy.drift(sy.make_cluster(1000, 20, 1, 100000, 3, [], []), 0.2, 2)[0]),
         ('1000x100000 drift 0.2 seed 3', lambda: sy.drift(sy.make_cluster(1000, 20, 1, 100000, 3, [], []), 0.2, 3)[0]),
         ('1000x100000 drift 0.4 seed 1', lambda: sy.drift(sy.make_cluster(1000, 20, 1, 100000, 3, [], []), 0.4, 1)[0]),
         ('2000x100000 drift 0.2 seed 1', lambda: sy.drift(sy.make_cluster(2000, 20, 1, 100000, 3, [], []), 0.2, 1)[0]),
         ('1000x200000 drift 0.2 seed 1', lambda: sy.drift(sy.make_cluster(1000, 20, 1, 200000, 3, [], []), 0.2, 1)[0]),
         ('1000x30000 drift 0.2 seed 2', lambda: sy.drift(sy.make_cluster(1000, 20, 1, 30000, 3, [], []), 0.2, 2)[0]),
         ('500x10000 drift 0.3 seed 4', lambda: sy.drift(sy.make_cluster(500, 10, 1, 10000, 3, [], []), 0.3, 4)[0])]
for name, mk in cases:
    try:
        t = mk()
        kao.solve([t], seed=1, max_launches=1)
        t0 = time.perf_counter(); r = kao.solve([t], seed=3, stop_at_bound=1, time_limit_s=6.0)[0]; dt = time.perf_counter() - t0
        tm = kao.last_solve_timing(); lp = kao.last_solve_lp()
        print(f"{name}: {r.status} objective {r.objective} certificate {r.upper_bound} gap {r.upper_bound - r.objective} read back {tm['results_read_back']:.3f}s launches {tm['launches']} cx {tm['cx_calls']} lp {lp}", flush=True)
    except Exception as e:
        print(name, 'ERROR', repr(e)[:200], flush=True)
P
cat gpurun_out/${T}_huge.log | cut -c1-300
