"""Innermost loops of a kernel's ISA (-S output): length, DS / VALU / scratch instruction counts, a signature.
usage: isa_loops.py file.s"""
import re, sys
lines = open(sys.argv[1]).read().splitlines()
labels = {}
for i, l in enumerate(lines):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m: labels[m.group(1)] = i
loops = []
for i, l in enumerate(lines):
    m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", l)
    if m:
        t = m.group(1) or m.group(2)
        if t in labels and labels[t] < i: loops.append((labels[t], i))
# innermost only
inner = [lp for lp in loops if not any(o != lp and lp[0] <= o[0] and o[1] <= lp[1] for o in loops)]
for a, b in sorted(inner):
    body = [x.strip() for x in lines[a:b + 1] if x.strip() and not x.strip().startswith((";", ".", "//"))]
    n = len(body)
    ds = sum(x.startswith("ds_") for x in body); sc_l = sum(x.startswith("scratch_load") for x in body); sc_s = sum(x.startswith("scratch_store") for x in body)
    valu = sum(x.startswith("v_") for x in body); dpp = sum("row_newbcast" in x for x in body); f64 = sum(("_f64" in x) for x in body); acc = sum(x.startswith("v_accvgpr") for x in body)
    gl = sum(x.startswith(("global_", "flat_")) for x in body)
    if n >= 20: print(f"lines {a}-{b}: {n:5d} instr  valu {valu:4d} (f64 {f64:4d}) ds {ds:3d} dpp {dpp:3d} scratch ld/st {sc_l:3d}/{sc_s:3d} accvgpr {acc:3d} global {gl:3d}")
