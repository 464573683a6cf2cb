"""Static view of a gfx950 kernel's ISA (hipcc -S output): basic blocks in layout order with VALU / SALU / LDS / VMEM counts,
v_readlane / v_writelane (SGPR spill traffic) and branch targets.  Usage: blocks.py file.s [kernel-name-substring]"""
import re, sys
src = open(sys.argv[1]).read().split("\n")
want = sys.argv[2] if len(sys.argv) > 2 else None
start = 0; end = len(src)
if want:
    for i, l in enumerate(src):
        if l.startswith("_Z") and want in l and l.rstrip().endswith(":") is False and ":" in l:
            start = i; break
    for i in range(start + 1, len(src)):
        if src[i].startswith("\t.section") or src[i].strip().startswith(".amdhsa_kernel"):
            end = i; break
blocks = []; cur = {"name": "entry", "ins": [], "line": start}
for i in range(start, end):
    l = src[i]
    m = re.match(r"^(\.LBB[0-9_]+):", l)
    if m:
        blocks.append(cur); cur = {"name": m.group(1), "ins": [], "line": i}
        continue
    s = l.strip()
    if not s or s.startswith(";") or s.startswith(".") or s.endswith(":"): continue
    cur["ins"].append(s.split(";")[0].strip())
blocks.append(cur)
tot = dict(valu=0, salu=0, lds=0, vmem=0, rl=0, wl=0)
for b in blocks:
    c = dict(valu=0, salu=0, lds=0, vmem=0, rl=0, wl=0, other=0); br = []
    for ins in b["ins"]:
        op = ins.split()[0]
        if op.startswith("v_readlane") or op.startswith("v_readfirstlane"): c["rl"] += 1; c["valu"] += 1
        elif op.startswith("v_writelane"): c["wl"] += 1; c["valu"] += 1
        elif op.startswith("v_"): c["valu"] += 1
        elif op.startswith("s_cbranch") or op.startswith("s_branch"): br.append(ins.split()[-1]); c["salu"] += 1
        elif op.startswith("s_waitcnt") or op.startswith("s_nop"): c["other"] += 1
        elif op.startswith("s_"): c["salu"] += 1
        elif op.startswith("ds_"): c["lds"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")): c["vmem"] += 1
        else: c["other"] += 1
    for k in tot: tot[k] += c[k]
    print(f"{b['name']:12s} L{b['line'] - start:5d} n={len(b['ins']):4d} valu={c['valu']:3d} (rl {c['rl']:2d} wl {c['wl']:2d}) salu={c['salu']:3d} lds={c['lds']:2d} vmem={c['vmem']:2d} -> {' '.join(br)}")
print(tot)
