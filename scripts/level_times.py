"""Per-tree-level durations of the level-loop kernels from a rocprofv3 --kernel-trace database (launch order = level order).
    python scripts/level_times.py <dir-with-db> <levels> <out.txt> "<title>" """
import glob, sqlite3, sys
d, levels, out, title = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4]
db = sorted(glob.glob(d + "/**/*.db", recursive=True))[0]
cur = sqlite3.connect(db).cursor()
lines = ["# " + title, "%-34s %8s " % ("kernel", "calls") + " ".join("level%d_us" % l for l in range(levels)) + "   sum_us"]
for pat in ("k_hist_build", "k_hist_reduce", "k_score", "k_argmax_stage1", "k_resolve_splits", "k_partition"):
    rows = cur.execute("select start, end - start, grid_x * grid_y * grid_z from kernels where name like ? order by start", ("%" + pat + "%",)).fetchall()
    if not rows:
        continue
    per = [[] for _ in range(levels)]
    for k, (st, du, g) in enumerate(rows):
        per[k % levels].append(du / 1000.0)
    means = [sum(p) / max(1, len(p)) for p in per]
    lines.append("%-34s %8d " % (pat, len(rows)) + " ".join("%9.1f" % m for m in means) + "   %7.1f" % sum(means))
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
