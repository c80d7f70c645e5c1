"""Kernel timeline of the last step in a rocprofv3 --kernel-trace directory: name, start offset, duration, gap to the previous kernel (us).
    python scripts/kernel_timeline.py <dir> [n_last]"""
import glob, re, sqlite3, sys
d = sys.argv[1]; n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 60
db = sorted(glob.glob(d + "/**/*.db", recursive=True))[0]
cur = sqlite3.connect(db).cursor()
rows = cur.execute("select name, start, end from kernels order by start").fetchall()
rows = rows[-n_last:]
t0 = rows[0][1]; prev_end = rows[0][1]
for n, s, e in rows:
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n); n = re.sub(r"\(.*$", "", n)
    print("%-42s start %9.1f  dur %8.1f  gap %6.1f" % (n[:42], (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3))
    prev_end = e
