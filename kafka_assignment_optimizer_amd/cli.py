"""`python -m kafka_assignment_optimizer_amd.cli` -- same flags and JSON in/out as cli/kao-cli (README.md:52-78),
through the ctypes binding.  All computation happens in libkao.so on the GPU."""
from __future__ import annotations

import argparse
import json
import sys

from . import assignment_to_json, canonicalize, init, solve, solve_multi, topics_from_json


def _racks(arg: str) -> dict:
    if ":" in arg and "{" not in arg and not arg.endswith(".json"):
        return {int(k): v for k, v in (kv.split(":") for kv in arg.split(",") if kv)}
    with open(arg) as f:
        return {int(k): str(v) for k, v in json.load(f).items()}


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="kafka_assignment_optimizer_amd.cli")
    ap.add_argument("--current", required=True, help="reassignment JSON (README.md:52-63), '-' = stdin")
    ap.add_argument("--broker-list", required=True, help="target brokers, CSV (README.md:48)")
    ap.add_argument("--racks", required=True, help='{"<brokerId>": "<rack>"} JSON file or id:rack,id:rack')
    ap.add_argument("--rf", type=int, default=0)
    ap.add_argument("--weights", default="4,1,2,2", help="LL,LF,FL,FF")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--time-limit", type=float, default=10.0)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--gpus", type=int, default=1, help="solve on devices device .. device+gpus-1 (kao_solve_multi)")
    ap.add_argument("--no-canonical", action="store_true")
    ap.add_argument("--out", default="")
    ap.add_argument("--report", action="store_true")
    ap.add_argument("--require-optimal", action="store_true",
                    help="withhold (exit status 4) any topic whose plan is feasible but not PROVEN optimal; "
                         "without it such plans are emitted with a warning on stderr")
    a = ap.parse_args(argv)
    doc = json.load(sys.stdin if a.current == "-" else open(a.current))
    w = [int(x) for x in a.weights.split(",")]
    topics = topics_from_json(doc, [int(b) for b in a.broker_list.split(",") if b], _racks(a.racks), rf=a.rf or None,
                              weights=((w[0], w[1]), (w[2], w[3])))
    init(a.device)
    if a.gpus > 1:
        res = solve_multi(topics, list(range(a.device, a.device + a.gpus)), seed=a.seed, time_limit_s=a.time_limit, stop_at_bound=1)
    else:
        res = solve(topics, seed=a.seed, time_limit_s=a.time_limit, stop_at_bound=1)
    ok_topics, assigns, rc = [], [], 0
    for t, r in zip(topics, res):
        if r.status in ("NO_FEASIBLE", "INFEASIBLE_PROVEN"):
            print(f"topic {t.name}: " + ("This problem is infeasible" if r.status == "INFEASIBLE_PROVEN"
                                          else "no feasible assignment found within the time limit"), file=sys.stderr)
            rc = 3
            continue
        if r.status != "OPTIMAL_PROVEN":
            # lp_solve only returns the exact optimum (README.md:135-136): never emit a possibly suboptimal plan silently
            print(f"warning: topic {t.name}: plan is feasible but NOT proven optimal ({r.status}): objective={r.objective} "
                  f"bound={r.upper_bound} gap={r.upper_bound - r.objective}"
                  + ("; withheld (--require-optimal)" if a.require_optimal else ""), file=sys.stderr)
            if a.require_optimal:
                rc = rc or 4
                continue
        ok_topics.append(t)
        assigns.append(r.assignment if a.no_canonical else canonicalize(t, r.assignment))
        if a.report:
            print(f"topic {t.name}: status={r.status} objective={r.objective} bound={r.upper_bound} "
                  f"seconds_to_best={r.seconds_to_best:.4f}", file=sys.stderr)
    text = json.dumps(assignment_to_json(ok_topics, assigns))
    if a.out:
        with open(a.out, "w") as f:
            f.write(text + "\n")
    else:
        print(text)
    return rc


if __name__ == "__main__":
    sys.exit(main())
