"""Per-kernel per-call averages of the PMC counters of one rocprofv3 --pmc run:  python scripts/pmc_table.py <dir> <out.txt> "<title>" """
import glob, re, sqlite3, sys
src, dst, title = sys.argv[1], sys.argv[2], sys.argv[3]
db = sorted(glob.glob(src + "/**/*.db", recursive=True))[-1]
cur = sqlite3.connect(db).cursor()
def short(n):
    n = re.sub(r"\(anonymous namespace\)::|^void ", "", n); return re.sub(r"\(.*$", "", n)[:46]
tab, ctrs = {}, []
for k, c, v, n in cur.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name"):
    if "gbrl" not in k: continue
    tab.setdefault(short(k), {})[c] = (v / max(n, 1), n)
    if c not in ctrs: ctrs.append(c)
lines = ["# " + title, "# per-call averages", "%-46s %6s " % ("kernel", "calls") + " ".join("%20s" % c for c in ctrs)]
for k, d in sorted(tab.items(), key=lambda kv: -sum(x[0] * x[1] for x in kv[1].values())):
    calls = max(x[1] for x in d.values())
    lines.append("%-46s %6d " % (k, calls) + " ".join("%20.0f" % d.get(c, (0, 0))[0] for c in ctrs))
open(dst, "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:16]))
