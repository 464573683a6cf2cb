"""Per basic block of one kernel (hipcc -S -gline-tables-only): instruction-class counts and the source lines (.loc) the
block's instructions come from, as a histogram.  Usage: annotate.py file.s mangled-kernel-name-prefix [first-line last-line]"""
import re, sys, collections
src = open(sys.argv[1]).read().split("\n")
want = sys.argv[2]
start = next(i for i, l in enumerate(src) if l.startswith(want) and ":" in l)
end = next(i for i in range(start + 1, len(src)) if src[i].strip().startswith(".amdhsa_kernel") or src[i].startswith("\t.section\t.rodata"))
cur_line = 0; cur_file = 0
blocks = []; cur = dict(name="entry", ins=[], lines=collections.Counter())
for i in range(start, end):
    l = src[i]
    m = re.match(r"^(\.LBB[0-9_]+):", l)
    if m:
        blocks.append(cur); cur = dict(name=m.group(1), ins=[], lines=collections.Counter()); continue
    s = l.strip()
    m = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
    if m: cur_file, cur_line = int(m.group(1)), int(m.group(2)); continue
    if not s or s.startswith(";") or s.startswith(".") or s.endswith(":"): continue
    ins = s.split(";")[0].strip()
    cur["ins"].append((ins, cur_line))
    cur["lines"][cur_line] += 1
blocks.append(cur)
def cls(op):
    if op.startswith(("v_readlane", "v_readfirstlane")): return "rl"
    if op.startswith("v_writelane"): return "wl"
    if op.startswith("v_"): return "valu"
    if op.startswith(("s_waitcnt", "s_nop")): return "wait"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    return "other"
for b in blocks:
    c = collections.Counter(cls(i.split()[0]) for i, _ in b["ins"])
    br = [i.split()[-1] for i, _ in b["ins"] if i.startswith(("s_cbranch", "s_branch"))]
    top = " ".join(f"{ln}x{n}" for ln, n in sorted(b["lines"].items()))
    print(f"{b['name']:11s} n={len(b['ins']):4d} V={c['valu'] + c['rl'] + c['wl']:3d} (rl{c['rl']:3d} wl{c['wl']:3d}) S={c['salu']:3d} L={c['lds']:2d} M={c['vmem']:2d} -> {','.join(br):28s} | {top}")
