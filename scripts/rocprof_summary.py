"""Turn a rocprofv3 results .db (ROCm 7.2 default output) into a small text summary for profiles/.
    python scripts/rocprof_summary.py <rocprof-output-dir> profiles/r02_x.txt ["title"]
Kernel table = the `top_kernels` view of the rocpd schema (same numbers as `--stats`), with long names shortened.  Below it the
dispatches of k_hist_build are split in time: a default bench.py run contains the benchmark's 2^20-row launches AND the small launches
of the predict_cfg5 leg (10 000 trees grown on 4096-row minibatches) under the same kernel name; the roofline figure in bench.py's
JSON line is the benchmark-shape group."""
import glob, os, re, sqlite3, sys

src, dst = sys.argv[1], sys.argv[2]
title = sys.argv[3] if len(sys.argv) > 3 else src
dbs = sorted(glob.glob(os.path.join(src, "**", "*.db"), recursive=True))
assert dbs, "no .db under " + src
out = ["# " + title, "# source: rocprofv3 (ROCm 7.2) sqlite output, view top_kernels; durations in microseconds", ""]
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*$", "", n)
    return n[:90]
for db in dbs:
    con = sqlite3.connect(db)
    cur = con.cursor()
    rows = cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    out.append("%-60s %8s %14s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for n, c, t, a, p in rows:
        if p < 0.01: continue
        out.append("%-60s %8d %14.1f %12.2f %6.2f%%" % (short(n), c, t, a, p))
    out.append("")
    # The predict_cfg5 leg (10 000 trees grown on 4096-row minibatches, categorical columns) launches the same k_hist_build template
    # on small inputs.  Everything before its first categorical kernel ran on 2^20 x 128 rows (timed workload, ensemble growth, cfg3 leg).
    t_split = cur.execute("select min(start) from kernels where name like '%k_cat_distinct_insert%'").fetchone()[0]
    if t_split is not None:
        out.append("# k_hist_build split at the start of the predict_cfg5 leg (first k_cat_distinct_insert dispatch):")
        out.append("%-60s %8s %12s %14s" % ("group", "calls", "avg_us", "total_us"))
        for label, cond in (("k_hist_build on 2^20 x 128 rows (bench shape: timed steps, ensemble growth, cfg3 leg)", "start < ?"),
                            ("k_hist_build on 4096-row minibatches (predict_cfg5 leg)", "start >= ?")):
            c, a, t = cur.execute("select count(*), avg(end - start) / 1000.0, sum(end - start) / 1000.0 from kernels where name like '%k_hist_build%' and " + cond, (t_split,)).fetchone()
            out.append("%-60s %8d %12.2f %14.1f" % (label[:60], c or 0, a or 0.0, t or 0.0))
            out.append("#   ^ " + label)
    con.close()
os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
open(dst, "w").write("\n".join(out) + "\n")
print("\n".join(out[:12]))
