#!/usr/bin/env python3
"""Per-loop instruction census of a gfx950 ISA listing (hipcc -S): which loops carry scratch traffic, SGPR-spill lane
moves, LDS and VALU instructions.  Uses the loop annotations the AMDGPU asm printer leaves on the block labels.

usage: isa_loops.py file.s [min_instructions]
"""
import collections
import re
import sys


def classify(op):
    if op.startswith("scratch_load"):
        return "scr_ld"
    if op.startswith("scratch_store"):
        return "scr_st"
    if op.startswith("v_readlane") or op.startswith("v_writelane"):
        return "lane"
    if op.startswith("ds_"):
        return "ds"
    if op.startswith("global_") or op.startswith("flat_") or op.startswith("buffer_"):
        return "vmem"
    if op.startswith("v_accvgpr"):
        return "acc"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path = sys.argv[1]
    floor = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    label_re = re.compile(r"^(\.LBB\d+_\d+):\s*(;.*)?$")
    own_re = re.compile(r"=>\s*This (Inner )?Loop Header: Depth=(\d+)")
    hdr_re = re.compile(r"Header=BB(\d+_\d+)")
    blocks = []  # (label, loop header label or None, depth, Counter, first_line)
    cur = None
    fn = None
    raw = open(path).read().splitlines()
    # a label's loop annotation may continue on the following comment-only lines: fold them onto the label line
    folded = []
    last_label = -1  # index in `folded` of a label line whose comment block is still open
    for line in raw:
        st = line.strip()
        if last_label >= 0 and st.startswith(";"):
            folded[last_label] = folded[last_label].rstrip() + " " + st
            folded.append("")
            continue
        folded.append(line)
        last_label = len(folded) - 1 if label_re.match(line.rstrip()) else -1
    for ln, line in enumerate(folded, 1):
        s = line.strip()
        if s.endswith(":") and s.startswith("_Z") and not s.startswith("."):
            fn = s[:-1]
        m = label_re.match(line.rstrip())
        if m:
            lab, com = m.group(1), m.group(2) or ""
            hdr = None
            depth = 0
            mo = own_re.search(com)
            if mo:
                hdr = lab[1:].replace("LBB", "BB")
                depth = int(mo.group(2))
            else:
                mh = hdr_re.search(com)
                if mh:
                    hdr = "BB" + mh.group(1)
                    md = re.search(r"Depth=(\d+)", com)
                    depth = int(md.group(1)) if md else 0
            cur = [lab, hdr, depth, collections.Counter(), ln, fn]
            blocks.append(cur)
            continue
        if cur is None or not s or s.startswith(";") or s.startswith(".") or s.startswith("//"):
            continue
        op = s.split()[0]
        if op.endswith(":"):
            continue
        cur[3][classify(op)] += 1
    loops = collections.OrderedDict()
    for lab, hdr, depth, cnt, ln, fn_ in blocks:
        if hdr is None:
            continue
        key = (fn_, hdr)
        if key not in loops:
            loops[key] = [depth, collections.Counter(), ln, 0]
        loops[key][1].update(cnt)
        loops[key][3] += 1
    cols = ["valu", "ds", "salu", "lane", "scr_ld", "scr_st", "acc", "vmem", "wait", "nop"]
    print("%-12s %5s %4s %6s " % ("loop", "line", "dep", "total") + " ".join("%6s" % c for c in cols))
    for (fn_, hdr), (depth, cnt, ln, nb) in loops.items():
        tot = sum(cnt.values())
        if tot < floor:
            continue
        print("%-12s %5d %4d %6d " % (hdr, ln, depth, tot) + " ".join("%6d" % cnt[c] for c in cols))
    allc = collections.Counter()
    for b in blocks:
        allc.update(b[3])
    print("%-12s %5s %4s %6d " % ("ALL", "", "", sum(allc.values())) + " ".join("%6d" % allc[c] for c in cols))


if __name__ == "__main__":
    main()
