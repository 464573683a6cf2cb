"""Condense rocprofv3 (ROCm 7.2, rocpd sqlite output) results into a small text summary for profiles/.

usage: python tools/summarize_prof.py gpurun_out/prof_<tag>
Reads <dir>/trace/*.db (kernel trace), <dir>/pmc_fetch, pmc_write, pmc_sq (counter passes).
FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x
(MI355X_MICROARCH.md, HBM section), so both the raw and the 2x-corrected read figure are printed.
"""
import glob
import os
import sqlite3
import sys
from collections import defaultdict

out = sys.argv[1]


def dbs(sub):
    return sorted(glob.glob(os.path.join(out, sub, "**", "*.db"), recursive=True))


def short(name):
    return name.split("(")[0].replace("kao::", "")


print("== kernel trace (rocprofv3 --kernel-trace --stats): dispatches grouped by kernel / grid / LDS ==")
for db in dbs("trace"):
    c = sqlite3.connect(db)
    rows = c.execute("select name, grid_x, workgroup_x, lds_size, vgpr_count, sgpr_count, duration from kernels").fetchall()
    groups = defaultdict(list)
    for name, gx, wx, lds, vg, sg, dur in rows:
        if "kao::" in name:
            groups[(short(name), gx // wx, lds, vg, sg)].append(dur)
    for (name, blocks, lds, vg, sg), d in sorted(groups.items()):
        d = sorted(d)
        print(f"{name:10s} workgroups={blocks:6d} lds={lds:6d}B vgpr={vg} sgpr={sg} calls={len(d):3d} "
              f"avg={sum(d)/len(d)/1e3:9.1f}us min={d[0]/1e3:9.1f}us median={d[len(d)//2]/1e3:9.1f}us max={d[-1]/1e3:9.1f}us")
    print("-- top kernels (all, % of GPU time) --")
    for name, calls, total, avg, pct in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels").fetchall()[:6]:
        print(f"{short(name)[:60]:60s} calls={calls:5d} total={total:10.1f}us avg={avg:9.1f}us {pct:5.1f}%")

for sub, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    print(f"== {ctr} per dispatch (KiB), separate --pmc pass ==")
    for db in dbs(sub):
        c = sqlite3.connect(db)
        groups = defaultdict(list)
        for name, gs, wg, val in c.execute("select kernel_name, grid_size, workgroup_size, value from counters_collection where counter_name=?", (ctr,)):
            if "kao::" in name:
                groups[(short(name), gs // wg)].append(val)
        for (name, blocks), v in sorted(groups.items()):
            mean = sum(v) / len(v)
            extra = f"  (x2 gfx950 wide-read correction: {2*mean*1024/1e6:.3f} MB)" if ctr == "FETCH_SIZE" else ""
            print(f"{name:10s} workgroups={blocks:6d} n={len(v):3d} mean={mean:12.1f} KiB = {mean*1024/1e6:9.3f} MB  min={min(v):.1f} max={max(v):.1f}{extra}")

print("== SQ counters per dispatch (mean), separate --pmc pass ==")
for db in dbs("pmc_sq"):
    c = sqlite3.connect(db)
    groups = defaultdict(lambda: defaultdict(list))
    for name, gs, wg, cn, val in c.execute("select kernel_name, grid_size, workgroup_size, counter_name, value from counters_collection"):
        if "kao::" in name:
            groups[(short(name), gs // wg)][cn].append(val)
    for (name, blocks), d in sorted(groups.items()):
        print(f"{name} workgroups={blocks}: " + ", ".join(f"{k}={sum(v)/len(v):.4g}" for k, v in sorted(d.items())))

# ---- machine-readable per-launch constants of the full-size K-search launch (largest duration group) ----
import json
try:
    const = {}
    c = sqlite3.connect(dbs("trace")[0])
    rows = c.execute("select name, grid_x, workgroup_x, duration from kernels").fetchall()
    ks = [(gx // wx, dur) for name, gx, wx, dur in rows if "k_search" in name]
    blocks = max(b for b, _ in ks)
    durs = sorted(d for b, d in ks if b == blocks)
    full = [d for d in durs if d > 0.7 * durs[len(durs) // 2]]  # drop the short kao_solve launches
    const["k_search_workgroups"] = blocks
    const["k_search_avg_us_trace"] = sum(full) / len(full) / 1e3

    def per_launch(sub, ctr):
        cc = sqlite3.connect(dbs(sub)[0])
        v = [val for name, val in cc.execute("select kernel_name, value from counters_collection where counter_name=?", (ctr,)) if "k_search" in name]
        v = sorted(v)
        return v[len(v) // 2]  # median: the first launch (init) and the short kao_solve launches are outliers
    fetch = per_launch("pmc_fetch", "FETCH_SIZE") * 1024 * 2
    write = per_launch("pmc_write", "WRITE_SIZE") * 1024
    const["k_search_hbm_bytes_per_launch"] = int(fetch + write)
    const["k_search_valu_insts_per_launch"] = int(per_launch("pmc_sq", "SQ_INSTS_VALU"))
    const["k_search_lds_insts_per_launch"] = int(per_launch("pmc_sq", "SQ_INSTS_LDS"))
    meta = {}
    try:
        with open(os.path.join(out, "bench_trace.json")) as f:
            b = json.loads(f.read().strip().splitlines()[-1])
        meta = {"config": int(b["config"]["workload"][3]), "workload_tag": "cfg%d-drift-fresh" % int(b["config"]["workload"][3]),
                "iters_per_launch": b["config"]["iters_per_launch"],
                "restarts_total": b["config"]["restarts_per_topic_rank0"] * b["config"]["topics_per_rank"][0],
                "bench_hip_event_avg_launch_ms": b["roofline"]["avg_launch_ms"]}
    except Exception as e:  # noqa: BLE001
        print("bench json unavailable:", e)
    const.update(meta)
    tag = os.path.basename(out.rstrip("/"))
    tag = tag[len("prof_"):] if tag.startswith("prof_") else tag
    const["source"] = "profiles/%s_final_rocprof_summary.txt (tools/profile.sh %s; summary.txt of gpurun_out/prof_%s, copied)" % (tag, tag, tag)
    print("== constants ==")
    print(json.dumps(const))
    with open(os.path.join(out, "pmc_constants.json"), "w") as f:
        json.dump([const], f)
except Exception as e:  # noqa: BLE001
    print("constants unavailable:", e)
