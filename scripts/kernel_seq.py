"""Per-dispatch kernel durations, in launch order, from a rocprofv3 --kernel-trace output directory:
    python scripts/kernel_seq.py <dir> <kernel-substring> [group]      (group: print the mean of every `group`-th dispatch, e.g. 6 = per tree level)"""
import glob, sqlite3, sys
d, pat = sys.argv[1], sys.argv[2]
group = int(sys.argv[3]) if len(sys.argv) > 3 else 0
db = sorted(glob.glob(d + "/**/*.db", recursive=True))[0]
cur = sqlite3.connect(db).cursor()
cols = [c[1] for c in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else "kernel_name"
rows = cur.execute("select %s, start, end from kernels order by start" % name_col).fetchall()
dur = [(e - s) / 1e3 for n, s, e in rows if pat in n]
print("%d dispatches of *%s*: mean %.2f us" % (len(dur), pat, sum(dur) / max(1, len(dur))))
if group:
    tail = dur[len(dur) % group:]
    for k in range(group):
        v = tail[k::group]
        print("  position %d of %d: mean %.2f us  min %.2f  (n=%d)" % (k, group, sum(v) / len(v), min(v), len(v)))
