import sqlite3, sys, json
out = sys.argv[1]
print("# rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline\n")
con = sqlite3.connect(f"{out}/trace/run_results.db"); cur = con.cursor()
print("| kernel | calls | total us | avg us | % |\n|---|---|---|---|---|")
meta = {}
for r in cur.execute("select * from top_kernels"):
    print("| %s | %d | %.1f | %.3f | %.2f |" % (r[0][:70], r[1], r[2], r[3], r[4]))
    if "lmpc_solve_kernel" in r[0]:
        meta = {"qp_kernel_ms": r[3] * 1e-3, "qp_kernel_calls": r[1], "clock_ghz": 2.3}
print("\n# rocprofv3 --pmc <counters> (separate passes; same bench command with --steps 5 --warmup 1), average per launch\n")
print("| kernel | counter | avg per launch |\n|---|---|---|")
res = {}
for name in ("pmc1", "pmc2", "pmc3", "pmc4"):
    con = sqlite3.connect(f"{out}/{name}/run_results.db"); cur = con.cursor()
    for r in cur.execute("select kernel_name, counter_name, avg(value) from counters_collection group by kernel_name, counter_name"):
        if "lmpc" in r[0]:
            kn = r[0].split("(")[0].replace("void ", "")
            print("| %s | %s | %.6g |" % (kn, r[1], r[2]))
            res.setdefault(kn, {})[r[1]] = r[2]
res["_meta"] = meta
json.dump(res, open(f"{out}/pmc.json", "w"), indent=1)
# launch resources as the runtime saw them (LDS bytes per workgroup = per problem; residency = min(160 KB / LDS, VGPR limit))
try:
    con = sqlite3.connect(f"{out}/trace/run_results.db"); cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    disp = next((t for t in tabs if "kernel_dispatch" in t), None)
    sym = next((t for t in tabs if "kernel_symbol" in t), None)
    if disp:
        cols = [r[1] for r in cur.execute(f"pragma table_info({disp})")]
        scols = [r[1] for r in cur.execute(f"pragma table_info({sym})")] if sym else []
        print("\n# launch resources (rocprofv3 kernel trace)\n")
        want_d = [c for c in ("lds_block_size", "scratch_size", "private_segment_size", "group_segment_size", "workgroup_size_x", "grid_size_x", "workgroup_size", "grid_size") if c in cols]
        want_s = [c for c in ("arch_vgpr_count", "accum_vgpr_count", "sgpr_count", "group_segment_size", "private_segment_size") if c in scols]
        if sym and "kernel_id" in cols and "id" in scols:
            namecol = "display_name" if "display_name" in scols else "kernel_name"
            q = "select s.%s, %s from %s d join %s s on d.kernel_id = s.id where s.%s like '%%lmpc%%' group by s.%s" % (
                namecol, ", ".join(["d." + c for c in want_d] + ["s." + c for c in want_s]), disp, sym, namecol, namecol)
            print("| kernel | " + " | ".join(want_d + want_s) + " |\n|---|" + "---|" * len(want_d + want_s))
            for r in cur.execute(q):
                print("| %s | %s |" % (r[0][:70], " | ".join(str(v) for v in r[1:])))
except Exception as e:  # schema differs between rocprofv3 versions: the tables above are the evidence, this one is a convenience
    print("\n(launch resources not available: %s)" % e)
