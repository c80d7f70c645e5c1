"""Per-kernel average durations from a rocprofv3 sqlite result (short names):  python scripts/kernel_times.py <dir> [pattern]"""
import glob, re, sqlite3, sys
db = sorted(glob.glob(sys.argv[1] + "/**/*.db", recursive=True))[0]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
cur = sqlite3.connect(db).cursor()
agg = {}
for name, dur, lds in cur.execute("select name, end - start, lds_size from kernels order by start"):
    n = re.sub(r"\(anonymous namespace\)::|^void |gbrl::kern::", "", name)
    n = re.sub(r"\(.*$", "", n)[:48]
    if pat and pat not in n: continue
    key = (n, lds)
    a = agg.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += dur / 1e3
for (n, lds), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-50s lds=%-7d calls=%-5d avg_us=%9.2f total_us=%10.1f" % (n, lds, c, t / c, t))
