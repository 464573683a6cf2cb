#!/usr/bin/env python3
"""bench.py -- candidate assignments/s + time-to-optimal of the HIP solver on BASELINE config 4
("10k-partition reassign": 500 brokers / 10 racks, 200 topics x 50 partitions RF 3, rolling replace
of 50 brokers), at 1/2/4/8 GPUs.

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path over this rank's topics: one K-search launch (iters_per_launch
local-search iterations x 64 neighbours for every restart of every topic) + one K-eval launch (full
evaluation of every restart's best snapshot, wavefront/workgroup min-reduce into one packed key per
topic) + the host read-back of those keys [+ for N > 1 the RCCL min-allreduce of the global best].
Topics shard across ranks (independent sub-problems, README.md:146-184) and every rank fills its
own GPU with restarts, so per-GPU work is fixed as N grows: "scaling": "weak".

value = (delta-evaluated neighbours + fully evaluated candidates) of ALL ranks / wall time of the K
timed steps (max over ranks).  Inputs (instance tables, restart states) are resident in HBM before
the timed region.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable


def host_cores() -> int:
    """CPUs this process may actually use: the affinity mask capped by the cgroup CPU quota (a GPU box with 256 hardware
    threads may grant the container 16 CPUs' worth of time)."""
    n = max(1, len(os.sched_getaffinity(0)))
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:        # cgroup v2: "<quota> <period>" or "max <period>"
            q, per = f.read().split()[:2]
            if q != "max":
                n = min(n, max(1, int(int(q) / int(per))))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                q, per = int(f.read()), int(g.read())
                if q > 0:
                    n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(topics, restarts, iters, budget_s=15.0):
    """The oracle's scalar C port of the same search (oracle/kao_port.c), on every host core, on a
    bounded sample of the same workload.  This is the ONLY place bench.py touches oracle/."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import kao_oracle as ko
    import kao_port as kp

    kp.build()
    ots = [ko.Topic(name=t.name, broker_ids=t.broker_ids, rack_of=t.rack_of, n_racks=t.n_racks,
                    n_partitions=t.n_partitions, rf=t.rf, current=t.current, weights=t.weights,
                    bounds_override=dict(t.bounds_override)) for t in topics]
    # 1 thread first (the scalar figure), then every host core: restarts are independent and the port keeps no global
    # state; the threads are native (pthreads inside oracle/kao_port.c), no Python in the loop
    t0 = time.perf_counter()
    n1 = 0
    k1 = 0
    while time.perf_counter() - t0 < budget_s / 5:
        n1 += kp.port_search(ots[0], 1, k1 & 1023, 1, iters)["n_eval"]
        k1 += 1
    rate1 = n1 / (time.perf_counter() - t0)
    cores = host_cores()
    n_eval = 0
    done_topics = 0
    per_call = max(4 * cores, -(-restarts // cores) * cores)   # a multiple of the thread count, >= 4 restarts per thread
    t0 = time.perf_counter()
    while True:   # whole passes over the topic list until the budget is used (a 256-thread host finishes one pass in ~1 s)
        for ot in ots:
            n_eval += kp.port_search_throughput(ot, 1, per_call, 1, iters, cores)
            done_topics += 1
            if time.perf_counter() - t0 > budget_s:
                break
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    # exact CPU solve (HiGHS on the README model; lp_solve itself is not installed) of one topic
    te = time.perf_counter()
    ex = ko.solve_exact(ots[0], 120)
    exact_s = time.perf_counter() - te
    return {"value": n_eval / dt, "unit": "candidates/s", "cores": cores, "kind": "port",
            "sample": f"{done_topics} topic passes (of the {len(ots)}-topic list, repeated) x {per_call} restarts x {iters} iterations, "
                      f"oracle/kao_port.c scalar replay of the same search on {cores} native host threads, {dt:.1f} s",
            "value_one_thread": rate1,
            "exact_solver": "HiGHS (scipy.optimize.milp) on the README model; lp_solve 5.5 not installed",
            "exact_seconds_per_topic": exact_s, "exact_objective_topic0": ex.objective}


def load_profile_constants(config, iters, restarts_total):
    """Per-launch PMC figures of the committed rocprofv3 run of THIS command (profiles/pmc_constants.json);
    returned only when workload, iterations and restart count match, else None (-> traffic: null)."""
    path = os.path.join(ROOT, "profiles", "pmc_constants.json")
    try:
        with open(path) as f:
            for e in json.load(f):
                if e["config"] == config and e["iters_per_launch"] == iters and e["restarts_total"] == restarts_total:
                    return e
    except (OSError, ValueError, KeyError):
        pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=4, help="BASELINE config (2..5); the metric is quoted on 4")
    ap.add_argument("--topics", type=int, default=0, help="truncate the topic list (debug)")
    ap.add_argument("--iters", type=int, default=512, help="local-search iterations per K-search launch")
    ap.add_argument("--restarts", type=int, default=0,
                    help="restarts per topic (0 = four full rounds of resident wavefronts: 256 CUs x 32 x 4 / topics)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--eval-bench", type=int, default=1, help="also time K-eval alone on a resident batch")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import kafka_assignment_optimizer_amd as kao
    from kafka_assignment_optimizer_amd import multigpu, synthetic

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the solver has no CPU path")
    # KAO_BENCH_SHARE_DEVICE=1 (test mode for a 1-GPU box): every rank uses device 0 and the collective runs
    # over gloo, so the N > 1 code path can be exercised without N GPUs.  Never set by the driver.
    share = os.environ.get("KAO_BENCH_SHARE_DEVICE", "0") == "1"
    # a launcher may mask devices per rank (HIP_VISIBLE_DEVICES): then every rank sees one device, index 0
    dev_index = 0 if share else local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    kao.init(dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
    dev = torch.device("cuda", dev_index)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- workload: identical synthetic instance on every rank, topics sharded (LPT) -------------
    topics_all = synthetic.make_config(args.config, n_topics=args.topics or None)
    sizes = [t.n_brokers * t.n_partitions for t in topics_all]
    if len(topics_all) >= world:
        shards = multigpu.shard_topics(sizes, world)
    else:  # fewer topics than GPUs: every rank searches every topic with its own seed; the min-allreduce picks the best
        shards = [list(range(len(topics_all))) for _ in range(world)]
    owned = shards[rank]
    topics = [topics_all[i] for i in owned]
    n_topics = len(topics_all)

    restarts = args.restarts
    if restarts <= 0:
        # throughput batch: ~4 rounds of resident wavefronts, so that fill/drain and the per-launch prologue/epilogue
        # amortize (measured: 3.8e11 cand/s at one partial round, 5.1e11 at three rounds)
        cus = torch.cuda.get_device_properties(dev_index).multi_processor_count
        restarts = max(8, (cus * 32 * 4 // max(1, len(topics))) // 4 * 4)
        restarts = min(restarts, 8192)
    sess = kao.Session(topics, seed=0xB0B + rank, restarts=restarts, iters_per_launch=args.iters, profile=1)

    def one_step():
        sess.step(1)
        keys = sess.best_keys()  # syncs the session stream
        if world > 1:
            return multigpu.allreduce_best(keys, owned, n_topics, rank, device=dev)
        return keys

    for _ in range(args.warmup):
        one_step()
    st0 = sess.stats()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    barrier()
    dt = time.perf_counter() - t0
    st1 = sess.stats()

    d_delta = st1["delta_candidates"] - st0["delta_candidates"]
    d_full = st1["full_candidates"] - st0["full_candidates"]
    ms_search = st1["ms_search"] - st0["ms_search"]
    ms_eval = st1["ms_eval"] - st0["ms_eval"]
    sb = st1["search_bytes_algo"] - st0["search_bytes_algo"]
    eb = st1["eval_bytes_algo"] - st0["eval_bytes_algo"]

    # aggregate over ranks: max time, summed candidates
    agg = torch.tensor([dt, float(d_delta), float(d_full)], dtype=torch.float64, device=dev)
    if world > 1:
        tmax = agg[:1].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(agg, op=dist.ReduceOp.SUM)
        agg[0] = tmax[0]
    dt_max, tot_delta, tot_full = (float(x) for x in agg.cpu())

    # ---- solution quality after the timed steps (all ranks' topics) -----------------------------
    res = sess.best()
    feasible = sum(1 for r in res if r.violations[0] == 0 and r.objective >= 0)
    proven = sum(1 for r in res if r.status == "OPTIMAL_PROVEN")
    drift = st1["drift"]
    sess.close()

    # ---- time-to-optimal: a fresh whole job (create + H2D + launches until every topic is proven
    #      optimal + D2H), wall clock from kao_solve entry
    barrier()
    t1 = time.perf_counter()
    sol = kao.solve(topics, seed=0x5EED + rank, iters_per_launch=64, stop_at_bound=1, time_limit_s=20.0)
    tto_wall = time.perf_counter() - t1
    tm = kao.last_solve_timing()
    all_proven = all(r.status == "OPTIMAL_PROVEN" for r in sol)
    tto = torch.tensor([tm["results_read_back"], 0.0 if all_proven else 1.0, tto_wall, tm["time_to_best"], float(tm["launches"])],
                       dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tto, op=dist.ReduceOp.MAX)
    tto_s, tto_fail, tto_wall_s, tto_best_s, tto_launches = (float(x) for x in tto.cpu())

    # ---- the same reassign after the cluster has drifted (20 % of the slots on random brokers): the closed-form bound
    #      has a gap on every topic, so "optimal" is proven by the Lagrangian dual kernel (K-bound) running beside K-search
    drifted = synthetic.drift(topics, 0.2, 1)
    barrier()
    kao.solve(drifted[:1], seed=1, iters_per_launch=64, max_launches=1)  # warm-up of the K-bound code path
    sold = kao.solve(drifted, seed=0xD21F + rank, iters_per_launch=64, stop_at_bound=1, time_limit_s=20.0)
    tmd = kao.last_solve_timing()
    dr = torch.tensor([tmd["results_read_back"], float(sum(r.status != "OPTIMAL_PROVEN" for r in sold)), float(tmd["launches"])],
                      dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(dr, op=dist.ReduceOp.MAX)
    dr_s, dr_unproven, dr_launches = (float(x) for x in dr.cpu())

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    out = {
        "metric": "candidate assignments/sec (+ time_to_optimal_s), 10k-partition reassign",
        "value": (tot_delta + tot_full) / dt_max,
        "unit": "candidates/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * dt_max / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "int32",
        "data": "synthetic",
        "config": {"workload": synthetic.WORKLOADS[args.config], "topics_total": n_topics,
                   "topics_per_rank": [len(s) for s in shards], "restarts_per_topic_rank0": st1["n_restarts_total"] // max(1, len(topics)),
                   "iters_per_launch": args.iters,
                   "neighbours_per_iteration": "REPLACE: B brokers of one slot (scan) or 64x4 (sample), EXCHANGE: P*rf partner slots, "
                                               "LEADER-SWAP: 64x(rf-1); pattern RRXRLRXR",
                   "parallelism": f"topic-sharded x{world}" if world > 1 else "single GPU"},
        "delta_candidates_per_s": tot_delta / dt_max,
        "full_candidates_per_s": tot_full / dt_max,
        "time_to_optimal_s": None if tto_fail else tto_s,
        "time_to_optimal_note": "seconds from kao_solve entry (instance in host memory) to results in host memory: instance "
                                "preparation + H2D + K-search/K-eval launches until every topic's objective equals its upper "
                                "bound (OPTIMAL_PROVEN) + gather + D2H; max over ranks",
        "time_to_optimal_detail": {"python_wall_s": tto_wall_s, "last_improving_launch_done_s": tto_best_s, "launches": int(tto_launches)},
        "time_to_optimal_drifted": {"workload": "the same topics after a 20 % drift (synthetic.drift): every topic has a "
                                                "closed-form bound gap, optimality is proven by K-bound (Lagrangian dual) on its own stream",
                                    "seconds": dr_s, "unproven_topics_max_over_ranks": int(dr_unproven), "launches": int(dr_launches)},
        "quality_after_timed_steps": {"topics_rank0": len(res), "feasible": feasible, "proven_optimal": proven, "drift": drift},
    }
    # ---- roofline of the dominant kernel (K-search), from HIP events on the session stream -------
    launches = st1["launches"] - st0["launches"]
    achieved = sb / (ms_search * 1e-3) / 1e9 if ms_search > 0 else None
    out["roofline"] = {"kernel": "k_search", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": achieved / HBM_PEAK_GBS if achieved else None, "traffic": None,
                       "algorithmic_bytes_per_launch": sb // max(1, launches),
                       "avg_launch_ms": ms_search / max(1, launches),
                       "note": "algorithmic bytes = neighbours x (8*RF+10) B (SURVEY.md 8d); the working set is "
                               "LDS-resident, so this kernel is VALU/LDS-issue bound, not HBM bound (DESIGN.md section 6)"}
    prof = load_profile_constants(args.config, args.iters, st1["n_restarts_total"])
    if prof:
        out["roofline"]["traffic"] = prof["k_search_hbm_bytes_per_launch"]
        out["roofline"]["traffic_note"] = ("rocprofv3 --pmc FETCH_SIZE (x2 gfx950 wide-read correction) + WRITE_SIZE per launch, "
                                           "separate passes, from " + prof["source"])
        valu = prof["k_search_valu_insts_per_launch"]
        # integer VALU issue peak, MEASURED on this chip (tools/microbench/valu_rate.hip, profiles/r01_valu_issue_microbench.txt):
        # 540-671 G wave64-instructions/s depending on the op (~4 cycles per instruction per SIMD); best class used as the roof
        peak = 671.3e9
        out["roofline_valu_issue"] = {"kernel": "k_search", "bound": "valu-issue", "insts_per_launch": valu,
                                      "achieved": valu / (ms_search / max(1, launches) * 1e-3) / 1e9, "peak": peak / 1e9,
                                      "unit": "G wave-instructions/s",
                                      "frac": valu / (ms_search / max(1, launches) * 1e-3) / peak,
                                      "note": "SQ_INSTS_VALU per launch from " + prof["source"] + "; peak = best measured integer-VALU class "
                                              "(profiles/r01_valu_issue_microbench.txt); this, not HBM, is the binding roof"}
    ach_e = eb / (ms_eval * 1e-3) / 1e9 if ms_eval > 0 else None
    out["roofline_eval_in_step"] = {"kernel": "k_eval", "achieved": ach_e, "unit": "GB/s", "avg_launch_ms": ms_eval / max(1, launches),
                                    "algorithmic_bytes_per_launch": eb // max(1, launches)}

    # ---- K-eval alone on a large resident batch: the genuinely HBM-streaming kernel ----------------
    if args.eval_bench:
        t = topics[0]
        n = 1 << 18
        per = t.n_partitions * t.rf
        g = torch.Generator(device=dev)
        g.manual_seed(1)
        cand = torch.randint(0, t.n_brokers, (n, per), dtype=torch.int32, device=dev, generator=g).to(torch.int16)
        obj = torch.empty(n, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        plan = kao.EvalPlan(t)
        for _ in range(2):
            plan.run(cand.data_ptr(), n, obj.data_ptr())
            plan.sync()
        ms = []
        for _ in range(5):
            plan.run(cand.data_ptr(), n, obj.data_ptr())
            ms.append(plan.sync())
        plan.close()
        ms_e = sum(ms) / len(ms)
        bytes_e = n * (2 * t.rf * t.n_partitions + 2 * t.rf_cur * t.n_partitions + t.n_brokers)
        out["roofline_eval_stream"] = {"kernel": "k_eval", "bound": "hbm", "candidates": n, "avg_launch_ms": ms_e,
                                       "achieved": bytes_e / (ms_e * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": bytes_e / (ms_e * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                       "full_candidates_per_s": n / (ms_e * 1e-3)}

    if world == 1 and not args.no_cpu_baseline:
        rpt = st1["n_restarts_total"] // max(1, len(topics))
        out["cpu_baseline"] = cpu_baseline(topics, rpt, args.iters)
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
