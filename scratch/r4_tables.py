"""Round 4: the committed tables under profiles/ from the JSON lines scratch/r4_ab.py wrote on the GPU box (gpurun_out/)."""
import collections
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
G = ROOT / "gpurun_out"


def rows(name):
    p = G / name
    return [json.loads(l) for l in open(p) if l.startswith("{")] if p.exists() else []


def polish_forms():
    libs = [("r3", "round 3 (lambda inlined, no fresh lane)"), ("inl_nf", "function inlined"), ("call_nf", "call"),
            ("inl", "function inlined + FRESH_LANE"), ("call", "call + FRESH_LANE")]
    data = collections.OrderedDict()
    for lib, _ in libs:
        for r in rows("r4b_ab_%s.jsonl" % lib):
            if "error" not in r:
                data.setdefault((r["case"], r["B"], r["prec"]), {})[lib] = r
    out = ["# Forms of the active-set polish x FRESH_LANE, every instantiation (round 4, one MI355X, one gpurun call)", "",
           "`scratch/r4_build_variants.sh` builds the library five ways, `scratch/r4_ab.py` times the QP kernel(s) of each entry point",
           "(HIP events inside `lmpc_solve_batch*`, median of 9 calls, ms per batch) and hashes the answers (X, U, dU, status, iters).",
           "`=`: the fp64 answers are bit for bit those of the round-3 build; `!`: they are not; fp32 / mixed answers are compared by their",
           "largest scaled distance from the fp64 answers instead (`e`).  `ST[...]`: statuses differ from the round-3 build's.", "",
           "| problem | batch | entry | kernel | " + " | ".join(t for _, t in libs) + " |", "|---|---|---|---|" + "---|" * len(libs)]
    kq = lambda n: next(k for k in (2, 4, 7, 11, 14) if (11 * n + 63) // 64 <= k)
    for (case, B, prec), d in data.items():
        n = int("".join(c for c in case if c.isdigit()))
        ks = 0 if not case.startswith("lmpc") else 3
        kern = "<%s, %d, %d>" % ("double" if prec == "f64" else "float", max(kq(n), 4) if prec != "f64" or ks else kq(n), ks)
        ref = d.get("r3")
        cells = []
        for lib, _ in libs:
            r = d.get(lib)
            if not r:
                cells.append("-")
                continue
            c = "%.3f" % r["qp_ms"]
            if prec == "f64":
                c += " =" if ref and r["sha"] == ref["sha"] else " **!**"
            else:
                c += " e %.0e" % r.get("err_max", 0)
            if ref and r["status"] != ref["status"]:
                c += " ST%s" % r["status"]
            cells.append(c)
        out.append("| %s | %d | %s | %s | " % (case, B, prec, kern) + " | ".join(cells) + " |")
    out += ["", "Timings of one build differ by 2-10 % between gpurun calls (another box, another clock state): compare within a row.",
            "`lmpc20` rows with batch 4096 appear twice: 160 safe-set points (KS = 3) and 96 (KS = 2, `lmpc96` case: N = 20 and the 2048-batch at N = 40)."]
    (ROOT / "profiles" / "r04_polish_forms.md").write_text("\n".join(out) + "\n")


def bisect():
    out = ["# The miscomputing <double, 7, 0> build: which sweep's FRESH_LANE, which compiler switch (round 4)", "",
           "Builds with the polish inlined (`-DLMPC_POLISH_CALL=0`) and FRESH_LANE (`asm volatile(\"\" : \"+v\"(lane))` at the top of a",
           "sweep function) enabled per function by `-DLMPC_FRESH_MASK` (bit 0 `riccati_factor`, bit 3 `riccati_solve_lds`, bit 4",
           "`feedback_rollout`); `scratch/r4_ab.py trk40 iac`: BARC N = 40 batch 4096 and IAC N = 40 batch 8192, fp64, status histogram",
           "[optimal, max-iter, infeasible, unverified], mean iterations, checksum of all answers.", "",
           "| build | BARC N = 40 | IAC N = 40 | verdict |", "|---|---|---|---|"]
    names = [("r3", "round-3 build"), ("rc_m01", "mask 0x01: factor only"), ("rc_m08", "mask 0x08: vector solve only"), ("rc_m10", "mask 0x10: rollout only"),
             ("rc_m11", "mask 0x11"), ("rc_m18", "mask 0x18"), ("rc_m09", "mask 0x09: factor + vector solve"),
             ("rc_wait0", "mask 0x7f, `-mllvm -amdgpu-waitcnt-forcezero`"), ("rc_nopost", "mask 0x7f, `-mllvm -enable-post-misched=false`"),
             ("rc_nosgpr", "mask 0x7f, `-mllvm -amdgpu-spill-sgpr-to-vgpr=false`"), ("rc_O2", "mask 0x7f, `-O2`")]
    good = None
    for lib, title in names:
        rr = {r["case"]: r for r in rows("r4b_rc_%s.jsonl" % lib) if r.get("prec") == "f64"}
        if not rr:
            continue
        a, b = rr.get("barc40"), rr.get("iac40")
        if lib == "r3":
            good = (a["sha"], b["sha"])
        ok = (a["sha"], b["sha"]) == good
        out.append("| %s | %s it %.2f `%s` %.2f ms | %s it %.2f `%s` %.2f ms | %s |" % (title, a["status"], a["iters_mean"], a["sha"], a["qp_ms"], b["status"], b["iters_mean"],
                                                                              b["sha"], b["qp_ms"], "same bits" if ok else "**wrong**"))
    (ROOT / "profiles" / "r04_d70_bisect.md").write_text("\n".join(out) + "\n")


if __name__ == "__main__":
    polish_forms()
    bisect()
    print(open(ROOT / "profiles" / "r04_polish_forms.md").read())
    print(open(ROOT / "profiles" / "r04_d70_bisect.md").read())
