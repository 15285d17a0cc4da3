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
