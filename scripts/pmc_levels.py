"""Per-tree-level counter attribution for a kernel that is launched once per level (k_hist_build: 6 launches per tree at depth 6).

    python scripts/pmc_levels.py <kernel-substring> <levels> <out.txt> "<title>" dir1 [dir2 ...]

Every directory is one `rocprofv3 --pmc ...` run (counters that fit one pass; several runs for several counter sets).  Dispatches of
the kernel are ordered by dispatch id; position = index modulo <levels> = tree level (the bench grows whole trees only).  For every
counter: mean per launch and per level; the dispatch duration the counter rows carry is reported too (durations under --pmc are a
few % above a plain run)."""
import glob, sqlite3, sys

pat, levels, out, title = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4]
lines = ["# " + title]
for d in sys.argv[5:]:
    for db in sorted(glob.glob(d + "/**/*.db", recursive=True)):
        cur = sqlite3.connect(db).cursor()
        rows = cur.execute("select dispatch_id, counter_name, sum(value), max(end - start), max(grid_size) from counters_collection where kernel_name like ? "
                           "group by dispatch_id, counter_name order by dispatch_id", ("%" + pat + "%",)).fetchall()
        if not rows:
            continue
        # keep the launches of the big shape only (the largest grid): the bench's other legs reuse the kernel on small inputs
        gmax = max(r[4] for r in rows)
        rows = [r for r in rows if r[4] == gmax]
        ids = sorted({r[0] for r in rows})
        pos = {i: k % levels for k, i in enumerate(ids)}
        ctrs = sorted({r[1] for r in rows})
        lines.append("## %s: %d launches, counters %s" % (d, len(ids), " ".join(ctrs)))
        dur = {l: [] for l in range(levels)}
        seen = set()
        for i, c, v, du, g in rows:
            if i not in seen:
                seen.add(i); dur[pos[i]].append(du / 1000.0)
        lines.append("%-28s %14s  " % ("counter", "mean/launch") + "  ".join("level %d      " % l for l in range(levels)))
        lines.append("%-28s %14.1f  " % ("duration_us", sum(sum(x) for x in dur.values()) / max(1, len(ids))) +
                     "  ".join("%13.1f" % (sum(dur[l]) / max(1, len(dur[l]))) for l in range(levels)))
        for c in ctrs:
            per = {l: [] for l in range(levels)}
            for i, cc, v, du, g in rows:
                if cc == c:
                    per[pos[i]].append(v)
            tot = sum(sum(x) for x in per.values()) / max(1, len(ids))
            lines.append("%-28s %14.1f  " % (c, tot) + "  ".join("%13.1f" % (sum(per[l]) / max(1, len(per[l]))) for l in range(levels)))
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
