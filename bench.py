#!/usr/bin/env python3
"""bench.py -- candidate assignments/s + time-to-optimal of the HIP solver on BASELINE config 4
("10k-partition reassign": 500 brokers / 10 racks, 200 topics x 50 partitions RF 3, rolling replace
of 50 brokers), at 1/2/4/8 GPUs.

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path over ONE FRESH BATCH of synthetic input: the config-4 cluster after a 20 % drift
(synthetic.drift, a different drift seed for every step), i.e. 200 reassignment problems nobody has solved yet, all
resident in HBM before the timed region.  A step runs, for this rank's topics, one K-search launch (best-insertion init +
iters_per_launch local-search iterations for every restart of every topic) + one K-eval launch (full evaluation of every
restart's best snapshot, wavefront/workgroup min-reduce into one packed key per topic) + the host read-back of those keys
[+ for N > 1 the min-allreduce of the global best over RCCL].  (Round 1 timed 20 launches on ONE batch, i.e. mostly on
topics that were already solved; that figure is still reported, as `steady_state`.)
Topics shard across ranks (independent sub-problems, README.md:146-184) and every rank fills its own GPU with
restarts, so per-GPU work is fixed as N grows: "scaling": "weak".

value = (delta-evaluated neighbours + fully evaluated candidates) of ALL ranks / wall time of the K timed steps (max over
ranks); `value_non_null` discounts the null proposals (same proposals as the scalar replay, which counts them).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ROUND_TAG = "r06"        # counter constants are taken from profiles/ of THIS round only: the kernels change between rounds
F64_PEAK_TFLOPS = 78.6   # MI355X f64 (vector = matrix) datasheet peak; the microarch guide has no f64 row
HBM_PEAK_GBS = 8000.0    # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable
LDS_PEAK_GBS = 256 * 128 * 2.4   # 128 B/clk/CU x 256 CUs x 2.4 GHz = 78.6 TB/s for ds_read_b128-class accesses (guide, LDS table)


FINAL_LINE_LIMIT = 4096   # the driver keeps a bounded tail of stdout: the result line must stay far below it


def _num(d, keys):
    """{k: d[k]} for the keys that are present and are numbers / short strings / None (no prose in the result line)."""
    o = {}
    for k in keys:
        if isinstance(d, dict) and k in d:
            v = d[k]
            if v is None or isinstance(v, (bool, int, float)) or (isinstance(v, str) and len(v) <= 96):
                o[k] = v
    return o


def compact_line(out):
    """The ONE result line: numbers only, < FINAL_LINE_LIMIT bytes.  Everything else (notes, probes, per-topic rows) goes to
    gpurun_out/bench_extras.json (emit())."""
    top = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
           "dtype", "data", "value_non_null", "delta_candidates_per_s", "full_candidates_per_s", "time_to_optimal_s")
    line = _num(out, top)
    cfg = out.get("config", {})
    line["config"] = {"workload": cfg.get("workload", "")[:200]}
    line["config"].update(_num(cfg, ("topics_total", "restarts_per_topic_rank0", "iters_per_launch", "parallelism")))
    if isinstance(cfg.get("topics_per_rank"), list) and len(cfg["topics_per_rank"]) <= 16:
        line["config"]["topics_per_rank"] = cfg["topics_per_rank"]          # (tools/summarize_prof.py keys the counter constants on it)
    roof_keys = ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "avg_launch_ms",
                 "traffic_source", "peak_source", "binding_bound", "binding_frac", "algorithmic_gbps")
    for k in ("roofline", "roofline_valu_issue", "roofline_lds", "roofline_eval_stream", "roofline_lp"):
        if k in out:
            line[k] = _num(out[k], roof_keys + ("valu_insts_per_neighbour", "ms_per_iteration", "iterations", "hbm_frac", "f64_frac",
                                                "hbm_bytes_per_iteration", "f64_flops_per_iteration", "chol_ms_per_iteration"))
    if "cpu_baseline" in out:
        line["cpu_baseline"] = _num(out["cpu_baseline"], ("value", "unit", "cores", "kind", "sample", "value_one_thread", "non_null_fraction",
                                                          "exact_solver", "exact_topics_solved", "exact_topics_total",
                                                          "exact_seconds_per_topic", "exact_wall_seconds"))
    if "north_star" in out:
        line["north_star"] = _num(out["north_star"], ("workload", "time_limit_s", "seconds", "status", "objective", "certificate",
                                                      "kao_lp_iterations", "seconds_unlimited", "status_unlimited"))
    if "extras_file" in out:
        line["extras_file"] = out["extras_file"]
    return line


def emit(out):
    """Write the full record to gpurun_out/bench_extras.json (scratch on the GPU box, merged back by gpurun) and print the compact
    result line LAST on stdout."""
    d = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
        path = os.path.join(d, "bench_extras.json")
        with open(path, "w") as f:
            json.dump(out, f, indent=1)
        out["extras_file"] = "gpurun_out/bench_extras.json"
    except OSError as ex:
        print("bench.py: could not write the extras file: %r" % (ex,), file=sys.stderr)
    line = json.dumps(compact_line(out))
    assert len(line) < FINAL_LINE_LIMIT, len(line)
    sys.stdout.flush()
    print(line, flush=True)


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


def _exact_one(args):
    """(process-pool worker) HiGHS on the README model of one topic; returns (seconds, objective or None)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import kao_oracle as ko
    d, limit = args
    t = ko.topic_from_dict(d)
    t0 = time.perf_counter()
    ex = ko.solve_exact(t, limit)
    return time.perf_counter() - t0, ex.objective if ex.status == "optimal" else None


def cpu_baseline(topics, restarts, iters, budget_s=12.0, exact_budget_s=45.0):
    """The oracle's scalar C port of the same search (oracle/kao_port.c), on every host core, on a bounded sample of the
    same workload, and the exact CPU solver (HiGHS; lp_solve is not installed) over the bench topics with a process pool.
    This is the ONLY place bench.py touches oracle/."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import multiprocessing as mp
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
    non_null = sum(kp.port_valid_fraction(ots[i % len(ots)], 1, i, 1, iters) for i in range(8)) / 8
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
    # exact CPU solve of the bench topics: HiGHS (scipy.optimize.milp) on the README model, one process per core
    te = time.perf_counter()
    secs, objs, n_done = [], [], 0
    with mp.Pool(min(cores, len(ots))) as pool:
        it = pool.imap(_exact_one, [(ko.topic_to_dict(t), 120.0) for t in ots])
        for _ in range(len(ots)):
            try:
                s, o = it.next(timeout=max(1.0, exact_budget_s - (time.perf_counter() - te)))
            except mp.TimeoutError:
                break
            secs.append(s); objs.append(o); n_done += 1
        pool.terminate()
    exact_wall = time.perf_counter() - te
    return {"value": n_eval / dt, "unit": "candidates/s", "cores": cores, "kind": "port",
            "sample": f"{done_topics} topic passes x {per_call} restarts x {iters} iters, {dt:.1f} s",
            "sample_detail": f"{done_topics} topic passes (of the {len(ots)}-topic list, repeated) x {per_call} restarts x {iters} iterations, "
                             f"oracle/kao_port.c scalar replay of the same search on {cores} native host threads, {dt:.1f} s",
            "value_one_thread": rate1, "non_null_fraction": non_null,
            "exact_solver": "HiGHS (scipy milp) on the README model; lp_solve 5.5 not installed",
            "exact_seconds_per_topic": (sum(secs) / n_done) if n_done else None,
            "exact_topics_solved": n_done, "exact_topics_total": len(ots), "exact_pool_processes": min(cores, len(ots)),
            "exact_wall_seconds": exact_wall, "exact_cpu_seconds_sum": sum(secs),
            "exact_cpu_seconds_per_topic_mean": (sum(secs) / n_done) if n_done else None,
            "exact_seconds_total_all_topics_est": (sum(secs) / n_done * len(ots) / min(cores, len(ots))) if n_done else None,
            "exact_objectives": objs}


def load_profile_constants(tag, iters, restarts_total):
    """Per-launch PMC figures of the committed rocprofv3 run of THIS command (profiles/pmc_constants.json);
    returned only when workload, iterations and restart count match, else None (-> traffic: null)."""
    path = os.path.join(ROOT, "profiles", "pmc_constants.json")
    try:
        with open(path) as f:
            for e in json.load(f):
                if e.get("workload_tag") == tag and e["iters_per_launch"] == iters and e["restarts_total"] == restarts_total:
                    return e
    except (OSError, ValueError, KeyError):
        pass
    return None


def load_big_constants(which, restarts):
    """Per-kernel counter figures of the north-star workloads (tools/profile_big.sh -> profiles/r0N_<tag>_big_<which>_constants.json):
    the latest profile taken with THIS restart count (traffic per launch scales with it), else None (-> traffic: null)."""
    import glob
    c = path = None
    for cand in sorted(glob.glob(os.path.join(ROOT, "profiles", ROUND_TAG + "*_big_%s_constants.json" % which)), reverse=True):   # this round's, the latest tag first
        try:
            with open(cand) as f:
                cc = json.load(f)
        except (OSError, ValueError):
            continue
        if cc.get("steps", {}).get("restarts") == restarts:
            c, path = cc, cand
            break
    if c is None:
        return None
    out = {"source": "profiles/" + os.path.basename(path)}
    for k, v in c.items():
        if isinstance(v, dict) and "hbm_bytes" in v:
            out["k_search" if k.startswith("k_search") else ("k_eval" if k.startswith("k_eval") else k)] = v
    return out


def load_lp_constants(workload):
    """Per-iteration counter figures of KAO-LP (tools/profile_lp_pmc.sh -> profiles/<round tag>*_lp_pmc_constants.json), this round's
    latest for the workload, else None (-> traffic: null)."""
    import glob
    for cand in sorted(glob.glob(os.path.join(ROOT, "profiles", ROUND_TAG + "*_lp_pmc_constants.json")), reverse=True):
        try:
            with open(cand) as f:
                c = json.load(f)
        except (OSError, ValueError):
            continue
        if c.get("workload") == workload:
            c["file"] = "profiles/" + os.path.basename(cand)
            return c
    return None


def write_cli_inputs(topics, path_prefix):
    """current.json / racks.json / broker list for cli/kao-cli from product topics (replicas on removed brokers get ids
    outside the target list, as in README.md:52-63 where broker 19 is about to be removed)."""
    t0 = topics[0]
    parts = []
    for t in topics:
        for p in range(t.n_partitions):
            reps = [int(t.broker_ids[b]) if b != 0xFFFF else 900000 + k for k, b in enumerate(t.current[p].tolist())]
            parts.append({"topic": t.name, "partition": p, "replicas": reps})
    with open(path_prefix + "current.json", "w") as f:
        json.dump({"version": 1, "partitions": parts}, f)
    with open(path_prefix + "racks.json", "w") as f:
        json.dump({str(int(b)): f"r{int(r)}" for b, r in zip(t0.broker_ids, t0.rack_of)}, f)
    return ",".join(str(int(b)) for b in t0.broker_ids)


def cold_cli(topics, device):
    """Wall time of a COLD cli/kao-cli process (exec + HIP init + first hipMalloc + solve + JSON out) on the workload."""
    exe = os.path.join(ROOT, "cli", "kao-cli")
    if not os.path.exists(exe):
        return None
    with tempfile.TemporaryDirectory() as d:
        pre = os.path.join(d, "b_")
        brokers = write_cli_inputs(topics, pre)
        t0 = time.perf_counter()
        out = subprocess.run([exe, "--current", pre + "current.json", "--broker-list", brokers, "--racks", pre + "racks.json",
                              "--device", str(device), "--time-limit", "20", "--out", pre + "out.json", "--no-canonical"],
                             capture_output=True, text=True)
        wall = time.perf_counter() - t0
        warn = sum(1 for ln in out.stderr.splitlines() if "NOT proven optimal" in ln)
        return {"wall_s": wall, "exit_code": out.returncode, "topics_not_proven": warn}


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
    ap.add_argument("--no-extras", action="store_true", help="timed steps only (what tools/profile.sh wraps in rocprofv3)")
    ap.add_argument("--eval-bench", type=int, default=1, help="also time K-eval alone on a resident batch")
    ap.add_argument("--in-library", action="store_true",
                    help="ONE process, --gpus devices, through kao_solve_multi (what kao-cli --gpus N ships: topics dealt LPT, RCCL only "
                         "when topics < devices) instead of one process per GPU; not the driver's launch mode")
    ap.add_argument("--devices", type=str, default="", help="--in-library: device list (csv; repeats = logical shards on one GPU)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import kafka_assignment_optimizer_amd as kao
    from kafka_assignment_optimizer_amd import multigpu, synthetic

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and not args.in_library:
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
    if args.in_library:
        # the in-library multi-GPU path: every timed step is one kao_solve_multi call on a FRESH drifted batch (upload, search to
        # the proof on every device, read-back) -- whole solves, not single launches, so `value` counts the neighbours evaluated
        # until every topic was proven per wall second of the whole call
        if world != 1:
            raise SystemExit("--in-library runs in ONE process (do not launch it through torch.distributed.run)")
        devices = [int(v) for v in args.devices.split(",")] if args.devices else list(range(args.gpus))
        batches = [synthetic.drift(topics_all, 0.2, 1 + i) for i in range(args.warmup + args.steps)]
        done = []
        for i, b in enumerate(batches):
            if i == args.warmup:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            rs = kao.solve_multi(b, devices, seed=0xB0B + i, time_limit_s=20.0)
            if i >= args.warmup:
                tm = kao.last_solve_timing()
                done.append((sum(r.status == "OPTIMAL_PROVEN" for r in rs), tm["delta_candidates"], tm["launches"], tm["elite_exchanges"]))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        cand = sum(d[1] for d in done)
        print(json.dumps({"metric": "candidate assignments/sec (+ time_to_optimal_s), 10k-partition reassign", "value": cand / dt,
                          "unit": "candidates/s", "n_gpus": len(devices), "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": dt / max(1, args.steps) * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                          "dtype": "int32", "data": "synthetic",
                          "config": {"workload": "BASELINE config %d, 20 %% drift, a fresh batch per step, each step one kao_solve_multi call to the "
                                                 "proof (upload + search + certificate + read-back)" % args.config,
                                     "parallelism": "in-library: one process, devices %s, topics dealt LPT" % devices},
                          "time_to_optimal_s": dt / max(1, args.steps), "topics": len(topics_all),
                          "topics_proven_per_step": [d[0] for d in done], "launches_per_step": [d[2] for d in done],
                          "elite_exchanges": sum(d[3] for d in done),
                          "note": "unmeasured on more than one physical GPU (the driver launches one process per GPU; repeated device ids are "
                                  "logical shards of one GPU)"}))
        return
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

    # one FRESH drifted batch per step (warm-up steps included), every session resident in HBM before the clock starts
    n_batches = args.warmup + args.steps
    batches = [synthetic.drift(topics, 0.2, 1 + i) for i in range(n_batches)]
    sessions = [kao.Session(b, seed=0xB0B + rank + 7919 * i, restarts=restarts, iters_per_launch=args.iters, profile=1)
                for i, b in enumerate(batches)]

    def one_step(sess):
        sess.step(1)
        if world > 1 and not share:   # min-allreduce on the resident key buffer (RCCL), no host round trip
            return multigpu.allreduce_best_resident(sess, owned, n_topics, rank)
        keys = sess.best_keys()  # syncs the session stream
        if world > 1:
            return multigpu.allreduce_best(keys, owned, n_topics, rank, device=dev)
        return keys

    for i in range(args.warmup):
        one_step(sessions[i])
    barrier()
    t0 = time.perf_counter()
    for i in range(args.warmup, n_batches):
        one_step(sessions[i])
    barrier()
    dt = time.perf_counter() - t0
    timed = sessions[args.warmup:]
    stats = [s.stats() for s in timed]

    def tot(k):
        return sum(st[k] for st in stats)
    d_delta, d_full = tot("delta_candidates"), tot("full_candidates")
    ms_search, ms_eval = tot("ms_search"), tot("ms_eval")
    sb, eb = tot("search_bytes_algo"), tot("eval_bytes_algo")
    launches = tot("launches")
    drift_ctr = tot("drift")
    restarts_total = stats[0]["n_restarts_total"]

    # aggregate over ranks: max time, summed candidates
    agg = torch.tensor([dt, float(d_delta), float(d_full)], dtype=torch.float64, device=dev)
    if world > 1:
        tmax = agg[:1].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(agg, op=dist.ReduceOp.SUM)
        agg[0] = tmax[0]
    dt_max, tot_delta, tot_full = (float(x) for x in agg.cpu())

    # ---- what the timed steps achieved: every batch's topics after its ONE step ------------------
    feasible = n_res = 0
    obj_after_step = []
    for s in timed:
        res = s.best()
        n_res += len(res)
        feasible += sum(1 for r in res if r.violations[0] == 0 and r.objective >= 0)
        obj_after_step.append(sum(int(r.objective) for r in res))

    out_extra = {}
    if not args.no_extras:
        # ---- steady state (round 1's figure): more launches on a batch that is already solved ----
        ss = timed[-1]
        sa = ss.stats()
        barrier()
        t1 = time.perf_counter()
        for _ in range(5):
            one_step(ss)
        barrier()
        dts = time.perf_counter() - t1
        sbb = ss.stats()
        out_extra["steady_state"] = {
            "candidates_per_s_rank0": (sbb["delta_candidates"] - sa["delta_candidates"] + sbb["full_candidates"] - sa["full_candidates"]) / dts,
            "note": "5 further launches on the last batch, whose topics are at or near their optimum already: machinery "
                    "throughput, not useful work (this is what round 1 reported as `value`)"}
    for s in sessions:
        s.close()

    tto = {}
    if not args.no_extras:
        # ---- time-to-optimal: fresh whole jobs (create + H2D + launches until every topic is PROVEN optimal + D2H),
        #      wall clock from kao_solve entry.  (a) the drifted batch of the first timed step: the closed-form bound has a
        #      gap on every topic, optimality is proven by K-bound beside K-search; (b) the balanced config as generated.
        def run_solve(tp, seed):
            barrier()
            t1 = time.perf_counter()
            sol = kao.solve(tp, seed=seed, stop_at_bound=1, time_limit_s=20.0)
            wall = time.perf_counter() - t1
            tm = kao.last_solve_timing()
            unproven = sum(r.status != "OPTIMAL_PROVEN" for r in sol)
            v = torch.tensor([tm["results_read_back"], float(unproven), wall, tm["time_to_best"], float(tm["launches"]),
                              float(tm["delta_candidates"]), float(sum(int(r.objective) for r in sol))], dtype=torch.float64, device=dev)
            if world > 1:
                vmax = v.clone(); dist.all_reduce(vmax, op=dist.ReduceOp.MAX)
                vsum = v.clone(); dist.all_reduce(vsum, op=dist.ReduceOp.SUM)
                v = vmax; v[5] = vsum[5]; v[6] = vsum[6]; v[1] = vsum[1]
            return [float(x) for x in v.cpu()]
        kao.solve(batches[0][:1], seed=1, max_launches=1)  # warm-up of the K-bound code path / arena cache
        d = run_solve(batches[args.warmup], 0xD21F + rank)
        b = run_solve(topics, 0x5EED + rank)
        tto = {"drifted": d, "balanced": b}

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    non_null = None
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
        "config": {"workload": synthetic.WORKLOADS[args.config] + "; every step a FRESH batch: the cluster after a 20 % drift "
                               "(synthetic.drift, seed 1 + step), all batches resident in HBM before the timed region",
                   "topics_total": n_topics, "topics_per_rank": [len(s) for s in shards],
                   "restarts_per_topic_rank0": restarts_total // max(1, len(topics)), "iters_per_launch": args.iters,
                   "neighbours_per_iteration": "REPLACE: B brokers of one slot (scan) or 64x4 (sample), EXCHANGE: P*rf partner slots, "
                                               "LEADER-SWAP: 64x(rf-1); pattern RRXRLRXR",
                   "parallelism": f"topic-sharded x{world}" if world > 1 else "single GPU"},
        "delta_candidates_per_s": tot_delta / dt_max,
        "full_candidates_per_s": tot_full / dt_max,
        "quality_after_one_step": {"batches": len(timed), "topics_rank0": n_res, "feasible": feasible, "drift": drift_ctr,
                                   "objective_sum_per_batch_rank0": obj_after_step},
    }
    out.update(out_extra)
    if tto:
        d, b = tto["drifted"], tto["balanced"]
        out["time_to_optimal_s"] = None if d[1] else d[0]
        out["time_to_optimal_note"] = ("seconds from kao_solve entry (instance in host memory) to results in host memory: instance "
                                       "preparation + H2D + K-search/K-eval launches (+ K-bound on its own stream) until every topic's "
                                       "objective equals its certificate (OPTIMAL_PROVEN) + gather + D2H; max over ranks; workload = "
                                       "the drifted batch of the first timed step (no topic is solved by the initial state)")
        out["time_to_optimal_detail"] = {"unproven_topics": int(d[1]), "python_wall_s": d[2], "last_improving_launch_done_s": d[3],
                                         "launches": int(d[4]), "delta_candidates": d[5], "objective_sum": d[6],
                                         "useful_candidates_per_s": d[5] / d[0] if d[0] > 0 else None,
                                         "objective_sum_after_one_timed_step_rank0": obj_after_step[0]}
        out["time_to_optimal_balanced"] = {"workload": "config 4 as generated (balanced start: the best-insertion init is optimal, "
                                                       "proof by the closed-form bound)", "seconds": None if b[1] else b[0],
                                           "python_wall_s": b[2], "launches": int(b[4]), "delta_candidates": b[5]}
        out["cold_cli"] = {"balanced": cold_cli(topics, dev_index), "drifted": cold_cli(batches[args.warmup], dev_index),
                           "note": "cli/kao-cli process start -> JSON written (exec, HIP init, first hipMalloc, JSON parse, solve, JSON out), "
                                   "rank 0's topics"}

    # ---- exactness at scale: single drifted topics of >= 1000 partitions, rank 0 only ----
    if not args.no_extras and rank == 0:
        probes = []
        # (brokers, racks, partitions, budget, HiGHS MILP optimum or None, HiGHS LP relaxation value or None)
        for (B_, R_, P_, budget, known, lp) in ((100, 5, 1000, 3.0, 7430, 7430.0), (300, 6, 2000, 3.0, 14826, 14826.0),
                                                (400, 8, 3000, 3.0, None, 22586.0), (450, 9, 3500, 3.0, None, 26330.0),
                                                (500, 10, 5000, 3.0, None, 37558.0), (1000, 20, 30000, 3.0, None, None)):
            tp = synthetic.drift(synthetic.make_cluster(B_, R_, 1, P_, 3, [], []), 0.2, 1)[0]
            t0 = time.perf_counter()
            r = kao.solve([tp], seed=3, stop_at_bound=1, time_limit_s=budget)[0]
            tm = kao.last_solve_timing()
            probes.append({"brokers": B_, "partitions": P_, "rf": 3, "status": str(r.status), "objective": int(r.objective),
                           "certificate": int(r.upper_bound), "exact_optimum_highs": known, "lp_relaxation_highs": lp,
                           "seconds": time.perf_counter() - t0, "seconds_to_best": float(r.seconds_to_best),
                           "k_bound_iterations": int(tm["bound_iters"]), "k_bound_launches": int(tm["bound_launches"]),
                           "kao_cx_calls": int(tm["cx_calls"]), "kao_cx_further_starts": int(tm["cx_further_starts"]), "generations": int(tm["generations"]),
                           "kao_lp_solves": int(tm["lp_solves"]), "kao_lp_iterations": int(tm["lp_iters"])})
        out["exactness_probe"] = {"topics": probes,
                                  "note": "one kao_solve call per topic (K-search + K-bound + KAO-CX), 20 % drift, tools/drift_scale.py's "
                                          "instances; exact references from tests/golden/drift_scale.json (HiGHS: MILP optimum where branch-and-"
                                          "bound finished, value of the LP relaxation where only that did; none for the largest).  Round 5: the "
                                          "certificate comes from KAO-LP (interior point on the compact LP, kao_lp.hip) wherever K-bound has not closed the topic"}
        # ---- KAO-LP alone: the LP relaxation on the device (kao_lp_bound): value, certificate = K-bound's exact dual value at the LP's duals, time ----
        lps = []
        for name, tp in [("450x3500", synthetic.drift(synthetic.make_cluster(450, 9, 1, 3500, 3, [], []), 0.2, 1)[0]),
                         ("500x5000", synthetic.drift(synthetic.make_cluster(500, 10, 1, 5000, 3, [], []), 0.2, 1)[0]),
                         ("drift30k", synthetic.north_star_topic("drift30k")), ("drift100k", synthetic.north_star_topic("drift100k"))]:
            kao.lp_trace(tp, max_iters=1)        # allocation / code-object warm-up
            t0 = time.perf_counter()
            b = kao.lp_bound(tp)
            mc = 3 * tp.n_racks + 2 * tp.n_brokers
            mcp = (mc + 63) // 64 * 64
            flops = b["iterations"] * (mcp ** 3 / 3.0)
            lps.append({"workload": name, "brokers": tp.n_brokers, "partitions": tp.n_partitions, "certificate": b["bound"], "lp_value": b["dual"],
                        "exact_dual_value_at_the_lp_duals": b["best_dual"] / 65536.0, "iterations": b["iterations"], "status": b["status"],
                        "interior_point_ms": b["ms"], "whole_call_ms": 1e3 * (time.perf_counter() - t0), "schur_rows": mc,
                        "cholesky_f64_flops": flops, "cholesky_gflops_over_the_whole_solve": flops / (b["ms"] * 1e-3) / 1e9 if b["ms"] > 0 else None})
            # the primal side: the iterate of the LP with perturbed costs, rounded and scored exactly (kao_lp_round)
            t0 = time.perf_counter()
            rr = kao.lp_round(tp)
            lps[-1]["rounded_iterate"] = {"objective": rr["objective"], "violations": rr["violations"][0], "equals_certificate": bool(rr["violations"][0] == 0 and rr["objective"] == b["bound"]),
                                          "perturbation": rr["pert"], "iterations": rr["iterations"], "status": rr["status"], "fractional_partitions": rr["fractional"],
                                          "interior_point_ms": rr["ms_lp"], "rounding_ms": rr["ms_round"], "whole_call_ms": 1e3 * (time.perf_counter() - t0)}
        out["lp_certificate"] = {"topics": lps,
                                 "note": "kao_lp_bound: Mehrotra predictor-corrector on the compact LP relaxation (new placements pooled per partition and rack), "
                                         "block elimination per partition, Schur complement of the 3R + 2B coupling rows gathered in fixed order, blocked f64 "
                                         "Cholesky (64 x 64 tiles); certificate = floor(K-bound's integer dual value at the rounded row duals).  The solve is a "
                                         "chain of ~250 small dependent kernels per iteration (latency-bound: profiles/r05_*_lp_*): the Cholesky flops over the "
                                         "whole solve time are ~1 % of the f64 vector peak -- reported, not a roofline claim.  rounded_iterate (kao_lp_round): the same solve with costs "
                                         "perturbed by eps * hash(variable) converges to ONE optimal vertex, which is rounded on the host and scored by K-eval: where its objective "
                                         "equals the certificate it IS an optimum of the model, found without a search",
                                 "reference": "HiGHS on the full model needed 2,876 s (450 x 3500, LP 26330) and 10,008 s (500 x 5000, LP 37558): tests/golden/drift_scale.json"}

    # ---- the north-star solve under the north-star's own budget, and the roofline of what decides it: a KAO-LP iteration ----
    if not args.no_extras and rank == 0:
        tp = synthetic.north_star_topic("drift100k")
        kao.solve([tp], seed=1, max_launches=1)               # arenas / code objects
        t0 = time.perf_counter()
        r = kao.solve([tp], seed=3, stop_at_bound=1, time_limit_s=1.0)[0]
        wall = time.perf_counter() - t0
        tm = kao.last_solve_timing(); lpi = kao.last_solve_lp()
        out["north_star"] = {"workload": "drift100k: 1000 brokers x 100,000 partitions RF 3, 20 % drift, one topic", "time_limit_s": 1.0,
                             "seconds": tm["results_read_back"], "python_wall_s": wall, "status": str(r.status), "objective": int(r.objective),
                             "certificate": int(r.upper_bound), "kao_lp_iterations": int(lpi["iterations"]), "kao_lp_solves": int(lpi["solves"]),
                             "fractional_partitions": int(lpi["fractional_partitions"]), "k_search_launches": int(tm["launches"]), "kao_cx_calls": int(tm["cx_calls"])}
        t0 = time.perf_counter()
        r3 = kao.solve([tp], seed=3, stop_at_bound=1, time_limit_s=3.0)[0]
        out["north_star"]["seconds_unlimited"] = kao.last_solve_timing()["results_read_back"]
        out["north_star"]["status_unlimited"] = str(r3.status)
        kao.lp_trace(tp, max_iters=1)
        b = kao.lp_bound(tp)
        ms_it = b["ms"] / max(1, b["iterations"])
        lc = load_lp_constants("drift100k")
        rl = {"kernel": "KAO-LP iteration (k_lp_*, k_chol_*, k_trsv)", "bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "ms_per_iteration": ms_it,
              "iterations": b["iterations"], "achieved": None, "frac": None, "traffic": None, "hbm_frac": None, "f64_frac": None,
              "note": "one interior-point iteration of the certificate's LP at 100,000 partitions, HIP-event time of the whole solve / iterations (captured graph); "
                      "bytes and counted f64 flops per iteration from this round's rocprofv3 passes (plain launches)"}
        if lc:
            rl.update(traffic=lc["hbm_bytes_per_iteration"], hbm_bytes_per_iteration=lc["hbm_bytes_per_iteration"],
                      f64_flops_per_iteration=lc["f64_flops_per_iteration_counted"], traffic_source=lc["file"],
                      kernel_ms_per_iteration_in_the_profile=lc["kernel_ms_per_iteration"])
            rl["achieved"] = lc["hbm_bytes_per_iteration"] / (ms_it * 1e-3) / 1e9
            rl["frac"] = rl["hbm_frac"] = rl["achieved"] / HBM_PEAK_GBS
            rl["f64_frac"] = lc["f64_flops_per_iteration_counted"] / (ms_it * 1e-3) / 1e12 / F64_PEAK_TFLOPS
            ks = lc.get("kernels_ms_per_iteration", {})
            rl["chol_ms_per_iteration"] = sum(v for k, v in ks.items() if k.startswith("k_chol"))
        out["roofline_lp"] = rl

    # ---- the north-star regime (BASELINE config 5): one LARGE topic, assignment words in HBM/L2 -- the kernel variants that
    #      run there (k_search<true, ...>, the cooperative k_eval) against the HBM peak; traffic from profiles/ (rocprofv3 --pmc) ----
    if not args.no_extras and rank == 0:
        big = []
        for which in ("drift30k", "drift100k", "cfg5one"):
            st = synthetic.north_star_steps(kao, which, launches=4)     # automatic restart count: 4 per compute unit (round 4)
            prof_b = load_big_constants(which, st["restarts"])
            e = {"workload": which, "brokers": st["brokers"], "partitions": st["partitions"], "restarts": st["restarts"],
                 "iters_per_launch": st["iters_per_launch"], "wall_ms_per_step": st["wall_ms_per_launch"], "drift": st["drift"]}
            for kern, ms, algo in (("k_search", st["k_search_ms_per_launch"], st["k_search_algorithmic_bytes_per_launch"]),
                                   ("k_eval", st["k_eval_ms_per_launch"], st["k_eval_algorithmic_bytes_per_launch"])):
                r = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "avg_launch_ms": ms, "algorithmic_bytes_per_launch": algo,
                     "algorithmic_gbps": algo / (ms * 1e-3) / 1e9 if ms > 0 else None, "achieved": None, "frac": None, "traffic": None}
                pk = prof_b.get(kern) if prof_b else None
                if pk and ms > 0:
                    r["traffic"] = pk["hbm_bytes"]
                    r["achieved"] = pk["hbm_bytes"] / (ms * 1e-3) / 1e9
                    r["frac"] = r["achieved"] / HBM_PEAK_GBS
                    r["traffic_over_algorithmic"] = pk["hbm_bytes"] / algo if algo else None
                    r["valu_insts_per_launch"] = pk.get("SQ_INSTS_VALU")
                    r["traffic_note"] = "rocprofv3 --pmc FETCH_SIZE (x2 gfx950 wide-read correction) + WRITE_SIZE per launch, separate passes, " + prof_b["source"]
                e[kern] = r
            if which in ("drift30k", "drift100k"):    # one 3-s kao_solve (K-search + K-bound + KAO-CX + KAO-LP), gap to the certificate, and K-search AS THE SOLVE RUNS IT
                tp = synthetic.north_star_topic(which)
                kao.solve([tp], seed=1, max_launches=1)
                t0 = time.perf_counter()
                r = kao.solve([tp], seed=3, stop_at_bound=1, time_limit_s=3.0, profile=1)[0]
                tm = kao.last_solve_timing()
                pf = kao.last_solve_profile()
                if pf["search_launches"] > 0 and pf["ms_search"] > 0:
                    ms_l = pf["ms_search"] / pf["search_launches"]
                    e["k_search"]["in_solve"] = {"avg_launch_ms": ms_l, "launches": pf["search_launches"], "restarts": pf["restarts"],
                                                 "algorithmic_bytes_per_launch": pf["search_bytes_algo"] / pf["search_launches"],
                                                 "algorithmic_gbps": pf["search_bytes_algo"] / (pf["ms_search"] * 1e-3) / 1e9,
                                                 "algorithmic_frac_of_hbm_peak": pf["search_bytes_algo"] / (pf["ms_search"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                                 "note": "HIP-event time of every K-search launch of the 3-s kao_solve below (kao_opts.profile, kao_last_solve_profile): the "
                                                         "solve keeps two restarts per compute unit on topics that live in HBM and launches K-search beside K-bound / KAO-LP, so "
                                                         "this -- not the session figure above (4 restarts per compute unit, nothing beside it) -- is what the product's solve "
                                                         "path runs.  Beyond 131,072 slots K-search is not launched at all between the first feasible incumbent and the end of the LP"}
                e["solve_3s"] = {"status": str(r.status), "objective": int(r.objective), "certificate": int(r.upper_bound),
                                 "gap": int(r.upper_bound - r.objective), "closed_form_bound": int(kao.upper_bound(tp)),
                                 "seconds": time.perf_counter() - t0, "seconds_to_best": float(r.seconds_to_best),
                                 "launches": int(tm["launches"]), "k_bound_iterations": int(tm["bound_iters"]), "kao_cx_calls": int(tm["cx_calls"]),
                                 "kao_lp_solves": int(tm["lp_solves"]), "kao_lp_iterations": int(tm["lp_iters"]), "kao_lp_rounding": kao.last_solve_lp(),
                                 "note": "no exact solver reaches this size: the certificate is K-bound's integer dual value at the multipliers of the LP relaxation "
                                         "solved on the device (KAO-LP, round 5; round 4: 782,627 from K-bound's own subgradient iteration); the incumbent is the rounded "
                                         "iterate of the same (perturbed) LP -- OPTIMAL_PROVEN means its objective equals that certificate"}
            big.append(e)
        out["roofline_big_topic"] = {"topics": big,
                                     "note": "K-search + K-eval steps of one session on a single large topic (synthetic.north_star_topic): "
                                             "the assignment words live in HBM/L2 (16 B per partition per restart, updated in place); k_search<true,...> is "
                                             "bound by the latency of dependent global loads at one wavefront per SIMD, not by bandwidth"}

    # ---- cluster-wide per-broker caps (BASELINE config 5 wording; kao_solve_capped) on the medium golden: wall time, plan
    #      against the exact joint optimum (HiGHS, tests/golden/capped_medium.json: numbers only, nothing of oracle/ runs here) ----
    if not args.no_extras and rank == 0:
        try:
            with open(os.path.join(ROOT, "tests", "golden", "capped_medium.json")) as f:
                cases = json.load(f)["cases"]
            rows = []
            for c in cases:
                tps = [kao.Topic.from_dict(d) for d in c["topics"]]
                t0 = time.perf_counter()
                res, lb = kao.solve_capped(tps, c["replica_cap"], seed=c["seed"], time_limit_s=10.0, max_rounds=60)
                rows.append({"topics": len(tps), "brokers": tps[0].n_brokers, "partitions_per_topic": tps[0].n_partitions,
                             "wall_s": time.perf_counter() - t0, "plan_objective": int(sum(int(r.objective) for r in res)),
                             "exact_joint_optimum_highs": c["objective"], "lagrangian_bound": lb, "objective_without_caps": c["objective_without_caps"]})
            out["capped_cluster"] = {"cases": rows, "note": "kao_solve_capped: Lagrangian prices on the capped brokers over independent per-topic solves"}
        except (OSError, ValueError, KeyError, AttributeError) as ex:
            out["capped_cluster"] = {"error": repr(ex)}

    # ---- roofline of the dominant kernel (K-search), duration from HIP events on the session stream -------
    avg_ms = ms_search / max(1, launches)
    algo_per_launch = sb // max(1, launches)
    prof = load_profile_constants("cfg%d-drift-fresh" % args.config, args.iters, restarts_total)
    roof = {"kernel": "k_search", "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "avg_launch_ms": avg_ms,
            "achieved": None, "frac": None, "traffic": None, "algorithmic_bytes_per_launch": algo_per_launch,
            "algorithmic_lds_served": {"bytes_per_launch": algo_per_launch, "gbps": algo_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else None,
                                       "note": "SURVEY.md 8(d) algorithmic bytes = neighbours x (8*RF+10) B; they are served from LDS (the "
                                               "restart state is LDS-resident), NOT from HBM, so they are not priced against the HBM peak"},
            "note": "achieved = HBM bytes per launch from the PMC counters / HIP-event launch time: the kernel touches HBM to load and store "
                    "restart states (25 MB read x2 correction + 23 MB written per launch) and, since round 5's occupancy floor (80 VGPRs for 6 "
                    "wavefronts per SIMD), to park ~20 VGPRs per lane in scratch around the iteration loop (+156 MB written per launch = 2.05 M "
                    "lanes x 20 x 4 B; read back mostly from L2): 230 MB in 11.9 ms = 19 GB/s.  It is bound by VALU issue (see roofline_valu_issue), "
                    "not by HBM"}
    if prof:
        roof["traffic"] = prof["k_search_hbm_bytes_per_launch"]
        roof["achieved"] = prof["k_search_hbm_bytes_per_launch"] / (avg_ms * 1e-3) / 1e9
        roof["frac"] = roof["achieved"] / HBM_PEAK_GBS
        roof["traffic_source"] = prof["source"].split(" ")[0]
        roof["traffic_note"] = ("rocprofv3 --pmc FETCH_SIZE (x2 gfx950 wide-read correction) + WRITE_SIZE per launch, separate passes, from "
                                + prof["source"])
        valu = prof["k_search_valu_insts_per_launch"]
        peak = prof.get("valu_issue_peak_winst_per_s", 671.3e9)
        out["roofline_valu_issue"] = {"kernel": "k_search", "bound": "valu-issue", "insts_per_launch": valu,
                                      "achieved": valu / (avg_ms * 1e-3) / 1e9, "peak": peak / 1e9, "unit": "G wave-instructions/s",
                                      "frac": valu / (avg_ms * 1e-3) / peak, "peak_source": "builder microbench (tools/microbench/valu_rate.hip), not datasheet",
                                      "valu_insts_per_neighbour": valu / max(1.0, d_delta / max(1, launches)),
                                      "note": "SQ_INSTS_VALU per launch from " + prof["source"] + "; peak: " + prof.get("valu_issue_peak_note", "")
                                              + ".  The peak is the fastest instruction class the microbench measured at the clock it "
                                              "measured, not a datasheet figure; frac is not capped.  Until round 5 the kernel held 97 VGPRs = 4 wavefronts per "
                                              "SIMD and was bound by latency, not by issue (3.7 % fewer VALU instructions changed nothing, "
                                              "profiles/r05_c28_scan_loop_ab.txt); with the register allocator held to 6 wavefronts per SIMD "
                                              "(amdgpu_waves_per_eu, 17-20 VGPRs spilled outside the loop) the same instruction stream runs "
                                              "11 % faster and fewer instructions pay again (profiles/r05_c29 / c30 / c31)",
                                      "occupancy": {"waves_per_simd": 6, "vgprs": 80}}
        if "k_search_lds_insts_per_launch" in prof:
            lds_b = prof["k_search_lds_insts_per_launch"] * 64 * prof.get("lds_bytes_per_lane_avg", 4)
            out["roofline_lds"] = {"kernel": "k_search", "bound": "lds", "achieved": lds_b / (avg_ms * 1e-3) / 1e9, "peak": LDS_PEAK_GBS,
                                   "unit": "GB/s", "frac": lds_b / (avg_ms * 1e-3) / 1e9 / LDS_PEAK_GBS,
                                   "note": "SQ_INSTS_LDS x 64 lanes x average access width; peak 128 B/clk/CU"}
    # the HBM figure above is the contract's; the roof that binds this LDS-resident kernel is VALU issue: both in the one block (numbers only)
    roof["algorithmic_gbps"] = roof["algorithmic_lds_served"]["gbps"]
    if "roofline_valu_issue" in out:
        roof["binding_bound"] = "valu-issue"; roof["binding_frac"] = out["roofline_valu_issue"]["frac"]
    out["roofline"] = roof
    ach_e = eb / (ms_eval * 1e-3) / 1e9 if ms_eval > 0 else None
    out["roofline_eval_in_step"] = {"kernel": "k_eval", "achieved": ach_e, "unit": "GB/s", "avg_launch_ms": ms_eval / max(1, launches),
                                    "algorithmic_bytes_per_launch": eb // max(1, launches)}

    # ---- K-eval alone on a large resident batch: the genuinely HBM-streaming kernel ----------------
    if args.eval_bench and not args.no_extras:
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

    if world == 1 and not args.no_cpu_baseline and not args.no_extras:
        rpt = restarts_total // max(1, len(topics))
        cb = cpu_baseline(batches[args.warmup], rpt, args.iters)
        out["cpu_baseline"] = cb
        non_null = cb["non_null_fraction"]
        out["value_non_null"] = out["value"] * non_null
        objs = [o for o in cb.pop("exact_objectives") if o is not None]
        if tto and objs and len(objs) == len(topics):   # the exact solver finished every topic: the certified optimum agrees
            out["time_to_optimal_detail"]["objective_sum_exact_cpu"] = sum(objs)
    emit(out)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
