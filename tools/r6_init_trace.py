"""Kernels of tools/r6_init_probe.py longer than 150 us (and every K-init), in launch order, from the rocprofv3 database."""
import glob, sqlite3, sys
out = sys.argv[1]
dbs = glob.glob(out + "/trace/**/*.db", recursive=True)
if not dbs: print("no rocprofv3 database under", out); sys.exit(0)
c = sqlite3.connect(dbs[0])
for n, s, e, g, w in c.execute("select name,start,end,grid_x,workgroup_x from kernels order by start").fetchall():
    us = (e - s) / 1e3
    if us > 150 or "k_init" in n:
        print(f"{n.replace('kao::', '').replace('(anonymous namespace)::', '').replace('void ', '')[:64]:64s} {us:10.1f} us  grid {g} wg {w}")
