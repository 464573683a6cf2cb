"""Merge the per-launch PMC constants tools/profile.sh wrote (gpurun_out/prof_<tag>/pmc_constants.json) into
profiles/pmc_constants.json, which bench.py quotes: the new entry replaces the one of the same workload / launch shape and
inherits the measured VALU issue peak (tools/microbench) from it.  usage: merge_pmc.py <tag>"""
import json, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
new = json.load(open(os.path.join(root, "gpurun_out", "prof_" + tag, "pmc_constants.json")))[0]
path = os.path.join(root, "profiles", "pmc_constants.json")
old = json.load(open(path))
key = lambda e: (e.get("workload_tag"), e.get("iters_per_launch"), e.get("restarts_total"))
for e in old:
    if key(e) == key(new):
        for k in ("valu_issue_peak_winst_per_s", "valu_issue_peak_note", "lds_bytes_per_lane_avg"):
            if k in e and k not in new:
                new[k] = e[k]
new["source"] = f"profiles/{tag}_final_rocprof_summary.txt (tools/profile.sh {tag})"
out = [new] + [e for e in old if key(e) != key(new)]
json.dump(out, open(path, "w"), indent=1)
print("merged", key(new), "VALU/launch", new.get("k_search_valu_insts_per_launch"), "avg us", new.get("k_search_avg_us_trace"))
