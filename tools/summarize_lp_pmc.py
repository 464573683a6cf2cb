"""Condense tools/profile_lp_pmc.sh output (rocprofv3, rocpd sqlite) into per-kernel lines and constants.json for bench.py's roofline_lp.
usage: summarize_lp_pmc.py gpurun_out/prof_lp_pmc_<tag> <tag>
HBM traffic = FETCH_SIZE x 2 (gfx950: FETCH_SIZE under-reports wide coalesced reads by 2x, MI355X_MICROARCH.md HBM section; the raw figure is kept
beside it: most of these kernels read 8 B per lane, for which the guide gives no calibration) + WRITE_SIZE, both in KiB, separate --pmc passes.
f64 flops per iteration are COUNTED, not measured: Cholesky n^3 / 3 (n = Schur rows padded to 64), four pairs of triangular solves 4 x 2 n^2,
the rack block 3 x 2 x (2R)^2 x P, the broker rows ~20 flops per (incidence, column, row), ~60 flops per variable per pass over the variables
(h, dir, residuals, updates: ~12 passes)."""
import glob, json, os, sqlite3, sys
from collections import defaultdict

out, tag = sys.argv[1], sys.argv[2]


def dbs(sub):
    return sorted(glob.glob(os.path.join(out, sub, "**", "*.db"), recursive=True))


def short(name):
    return name.replace("(anonymous namespace)::", "").split("(")[0].replace("kao::", "").replace("void ", "")


run = json.loads(open(os.path.join(out, "run_trace.json")).read().strip().splitlines()[-1])
its = sum(r["iterations"] for r in run["runs"])
print("== workload ==")
print(json.dumps(run))
dur, calls = defaultdict(float), defaultdict(int)
for db in dbs("trace"):
    c = sqlite3.connect(db)
    for name, d in c.execute("select name, duration from kernels"):
        if "kao::" in name:
            dur[short(name)] += d; calls[short(name)] += 1
ctr = defaultdict(lambda: defaultdict(float))
for sub in ("pmc_fetch", "pmc_write"):
    for db in dbs(sub):
        c = sqlite3.connect(db)
        for name, cn, val in c.execute("select kernel_name, counter_name, value from counters_collection"):
            if "kao::" in name:
                ctr[short(name)][cn] += val
tot_ns = sum(dur.values())
fetch = sum(v.get("FETCH_SIZE", 0.0) for v in ctr.values()) * 1024
write = sum(v.get("WRITE_SIZE", 0.0) for v in ctr.values()) * 1024
print(f"== {its} interior-point iterations over {len(run['runs'])} solves (starting points included in the totals) ==")
print(f"{'kernel':40s} {'calls':>7s} {'ms/iteration':>13s} {'share':>6s} {'FETCHx2 MB/it':>14s} {'WRITE MB/it':>12s}")
for k in sorted(dur, key=lambda q: -dur[q]):
    f = ctr[k].get("FETCH_SIZE", 0.0) * 1024 * 2 / its / 1e6
    w = ctr[k].get("WRITE_SIZE", 0.0) * 1024 / its / 1e6
    print(f"{k[:40]:40s} {calls[k]:7d} {dur[k] / its / 1e6:13.4f} {100 * dur[k] / tot_ns:5.1f}% {f:14.2f} {w:12.2f}")
B, R, P, RF = run["brokers"], run["racks"], run["partitions"], run["rf"]
n = (3 * R + 2 * B + 63) // 64 * 64
nv = (3 * RF + 3 * R) * P
flops = n ** 3 / 3.0 + 4 * 2.0 * n * n + 3 * 2.0 * (2 * R) ** 2 * P + 20.0 * (RF * P) * (2 * RF + 2 * R) * 2 + 60.0 * nv * 12
ms_it = tot_ns / its / 1e6
hbm = (2 * fetch + write) / its
const = {"tag": tag, "workload": run["workload"], "brokers": B, "partitions": P, "iterations": its, "kernel_ms_per_iteration": ms_it,
         "fetch_bytes_raw_per_iteration": fetch / its, "write_bytes_per_iteration": write / its, "hbm_bytes_per_iteration": hbm,
         "f64_flops_per_iteration_counted": flops, "schur_rows_padded": n,
         "kernels_ms_per_iteration": {k: dur[k] / its / 1e6 for k in dur},
         "source": f"profiles/{tag}_lp_pmc_summary.txt (tools/profile_lp_pmc.sh {tag})"}
print(f"== per iteration: {ms_it:.3f} ms of kernels, HBM {hbm / 1e9:.3f} GB (FETCH x2 {2 * fetch / its / 1e9:.3f} + WRITE {write / its / 1e9:.3f}) = {hbm / (ms_it * 1e-3) / 1e12:.3f} TB/s "
      f"= {hbm / (ms_it * 1e-3) / 8e12:.4f} of 8 TB/s; counted f64 flops {flops / 1e9:.2f} G = {flops / (ms_it * 1e-3) / 1e12:.3f} TFLOP/s = {flops / (ms_it * 1e-3) / 78.6e12:.4f} of 78.6 TFLOP/s ==")
json.dump(const, open(os.path.join(out, "constants.json"), "w"), indent=1)
