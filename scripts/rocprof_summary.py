"""Turn a rocprofv3 results .db (ROCm 7.2 default output) into a small text summary for profiles/.
    python scripts/rocprof_summary.py <rocprof-output-dir> profiles/r02_x.txt ["title"]
Kernel table = the `top_kernels` view of the rocpd schema (same numbers as `--stats`), with long names shortened.  Below it the
dispatches of k_hist_build are broken down by grid size: a default bench.py run contains the benchmark's 2^20-row launches AND the
small launches of the predict_cfg5 leg (10 000 trees grown on 4096-row minibatches) under the same kernel name; the roofline figure in
bench.py's JSON line is the benchmark-shape group."""
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
    out.append("# dispatches by (kernel, grid size): the same kernel name covers the benchmark's 2^20-row launches and the small-batch legs")
    out.append("%-60s %12s %8s %12s %12s" % ("kernel", "grid_x", "calls", "avg_us", "total_us"))
    for pat in ("k_hist_build", "k_predict_obl2"):
        q = cur.execute("select name, grid_x, count(*), avg(end - start) / 1000.0, sum(end - start) / 1000.0 from kernels where name like ? group by name, grid_x order by sum(end - start) desc", ("%" + pat + "%",)).fetchall()
        for n, g, c, a, t in q[:8]:
            out.append("%-60s %12d %8d %12.2f %12.1f" % (short(n), g, c, a, t))
    con.close()
os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
open(dst, "w").write("\n".join(out) + "\n")
print("\n".join(out[:12]))
