"""Per-kernel PMC averages from one or more rocprofv3 --pmc output directories:
    python scripts/pmc_table.py <kernel-substring> dir1 [dir2 ...]"""
import glob, re, sqlite3, sys
pat = sys.argv[1]
vals = {}
for d in sys.argv[2:]:
    for db in sorted(glob.glob(d + "/**/*.db", recursive=True)):
        cur = sqlite3.connect(db).cursor()
        try:
            rows = cur.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name").fetchall()
        except Exception as e:
            print("skip", db, e); continue
        for k, c, v, n in rows:
            if pat in k:
                vals[c] = (v / max(n, 1), n)
for c in sorted(vals):
    print("%-28s %16.1f  (calls %d)" % (c, vals[c][0], vals[c][1]))
