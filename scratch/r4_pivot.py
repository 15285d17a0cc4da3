"""Pivot of scratch/r4_ab.py JSON lines: one row per (case, batch, entry), one column per library build: kernel ms, '=' when the
answers' checksum equals the first column's, otherwise the checksum's head; for fp32/mixed the largest scaled distance from fp64."""
import collections
import json
import sys

rows = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")]
libs = []
data = collections.OrderedDict()
for r in rows:
    lib = r["lib"].replace("liblmpc_", "").replace(".so", "")
    if lib not in libs:
        libs.append(lib)
    if "error" in r:
        print("ERROR", lib, r)
        continue
    data.setdefault((r["case"], r["B"], r["prec"], r.get("reg", False)), {})[lib] = r
print("| case | B | entry | " + " | ".join(libs) + " |")
print("|---|---|---|" + "---|" * len(libs))
for (case, B, prec, reg), d in data.items():
    ref = d.get(libs[0])
    cells = []
    for lib in libs:
        r = d.get(lib)
        if not r:
            cells.append("")
            continue
        c = "%.3f" % r["qp_ms"]
        c += " =" if ref and r["sha"] == ref["sha"] else " " + r["sha"][:5]
        if prec != "f64":
            c += " e%.0e" % r.get("err_max", 0)
        if ref and r["status"] != ref["status"]:
            c += " ST%s" % r["status"]
        cells.append(c)
    print("| %s%s | %d | %s | " % (case, "+reg" if reg else "", B, prec) + " | ".join(cells) + " |")
