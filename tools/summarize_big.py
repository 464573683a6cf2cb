"""Condense tools/profile_big.sh output (rocprofv3, rocpd sqlite) into per-kernel roofline lines for profiles/ and
constants.json for bench.py.  usage: summarize_big.py gpurun_out/prof_big_<tag>/<workload>
HBM traffic per launch = FETCH_SIZE x 2 (gfx950: FETCH_SIZE under-reports wide coalesced reads by 2x, MI355X_MICROARCH.md HBM
section) + WRITE_SIZE, both in KiB, from separate --pmc passes; peaks: HBM 8 TB/s, L2 34.5 TB/s."""
import glob, json, os, sqlite3, sys
from collections import defaultdict

out = sys.argv[1]
HBM, L2 = 8000.0, 34500.0


def dbs(sub):
    return sorted(glob.glob(os.path.join(out, sub, "**", "*.db"), recursive=True))


def short(name):
    return name.replace("(anonymous namespace)::", "").split("(")[0].replace("kao::", "").replace("void ", "")


steps = {}
try:
    steps = json.loads(open(os.path.join(out, "steps_trace.json")).read().strip().splitlines()[-1])
    print("== workload ==")
    print(json.dumps(steps))
except Exception as e:  # noqa: BLE001
    print("steps json unavailable:", e)

dur = defaultdict(list)
for db in dbs("trace"):
    c = sqlite3.connect(db)
    for name, d in c.execute("select name, duration from kernels"):
        if "kao::" in name:
            dur[short(name)].append(d)
ctr = defaultdict(lambda: defaultdict(list))
for sub in ("pmc_fetch", "pmc_write", "pmc_sq"):
    for db in dbs(sub):
        c = sqlite3.connect(db)
        for name, cn, val in c.execute("select kernel_name, counter_name, value from counters_collection"):
            if "kao::" in name:
                ctr[short(name)][cn].append(val)


def steady(v):
    """median of the dispatches after the first (launch 0 builds the initial state)"""
    v = list(v)[1:] if len(v) > 1 else list(v)
    v = sorted(v)
    return v[len(v) // 2] if v else None


const = {"workload_tag": steps.get("workload"), "steps": steps}
print("== per kernel, steady launches (median after the first dispatch) ==")
for k in sorted(dur):
    us = steady(dur[k]) / 1e3
    line = f"{k:44s} dispatches={len(dur[k]):3d} {us:10.1f} us"
    cc = ctr.get(k, {})
    entry = {"us": us}
    if "FETCH_SIZE" in cc and "WRITE_SIZE" in cc:
        hbm = steady(cc["FETCH_SIZE"]) * 1024 * 2 + steady(cc["WRITE_SIZE"]) * 1024
        gbs = hbm / (us * 1e-6) / 1e9
        entry.update(hbm_bytes=int(hbm), hbm_gbs=gbs, frac_hbm=gbs / HBM, frac_l2=gbs / L2)
        line += f"  HBM {hbm / 1e6:9.3f} MB/launch = {gbs:8.1f} GB/s = {gbs / HBM:6.4f} of 8 TB/s ({gbs / L2:6.4f} of the 34.5 TB/s L2)"
    for cn in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES"):
        if cn in cc:
            entry[cn] = steady(cc[cn])
    if "SQ_INSTS_VALU" in entry:
        g = entry["SQ_INSTS_VALU"] / (us * 1e-6) / 1e9
        entry["valu_ginst_s"] = g
        line += f"  VALU {entry['SQ_INSTS_VALU']:.3e}/launch = {g:7.1f} G wave-instr/s ({g / 673.2:5.3f} of the measured issue roof)"
    const[k] = entry
    print(line)
if steps:
    ks = [k for k in const if k.startswith("k_search")]
    if ks and "hbm_bytes" in const[ks[0]]:
        e = const[ks[0]]
        alg = steps["k_search_algorithmic_bytes_per_launch"]
        print(f"k_search: algorithmic bytes/launch {alg:.3e} (neighbours x (8 RF + 10)) = {alg / (e['us'] * 1e-6) / 1e9:.1f} GB/s; measured HBM traffic / algorithmic = {e['hbm_bytes'] / alg:.3f}")
print("== one kao_solve: kernels by GPU time ==")
try:
    print(open(os.path.join(out, "solve.json")).read().strip().splitlines()[-1])
except Exception as e:  # noqa: BLE001
    print("solve json unavailable:", e)
tot = defaultdict(lambda: [0, 0])
for db in dbs("solve_trace"):
    c = sqlite3.connect(db)
    for name, d in c.execute("select name, duration from kernels"):   # duration in ns
        e = tot[short(name)[:58]]
        e[0] += 1
        e[1] += d
gpu_ns = sum(e[1] for e in tot.values()) or 1
solve_share = {}
for name, (calls, ns) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:16]:
    solve_share[name] = {"calls": calls, "total_ms": ns / 1e6, "avg_us": ns / calls / 1e3, "share": ns / gpu_ns}
    print(f"{name:58s} calls={calls:6d} total={ns / 1e6:9.3f} ms avg={ns / calls / 1e3:10.2f} us {100.0 * ns / gpu_ns:5.1f}%")
const["solve_kernel_share"] = solve_share
with open(os.path.join(out, "constants.json"), "w") as f:
    json.dump(const, f, indent=1)
