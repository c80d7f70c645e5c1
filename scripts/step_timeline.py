"""Timeline of ONE boosting step from a rocprofv3 kernel trace (.db): every dispatch of the step in start order with its duration and the
idle gap in front of it, averaged over the last steps of the run; plus busy time (union of the dispatch intervals) against the step's
span.  Steps are cut at the first dispatch of `marker` (default k_transpose_count: one per step at the bench shape).
    python3 scripts/step_timeline.py <rocprof-output-dir> out.txt [marker] [n_last_steps]
Run on the GPU box after  rocprofv3 --kernel-trace -d DIR -o t -- python3 bench.py --no-cpu-baseline --no-extra-legs --steps 20 --warmup 5"""
import glob, os, re, sqlite3, sys

src, dst = sys.argv[1], sys.argv[2]
marker = sys.argv[3] if len(sys.argv) > 3 else "k_transpose_count"
n_last = int(sys.argv[4]) if len(sys.argv) > 4 else 10
db = sorted(glob.glob(os.path.join(src, "**", "*.db"), recursive=True))[0]
cur = sqlite3.connect(db).cursor()
rows = cur.execute("select name, start, end from kernels order by start").fetchall()
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n); n = re.sub(r"\(.*$", "", n)
    return n.replace("gbrl::kern::", "")[:56]
cuts = [i for i, r in enumerate(rows) if marker in r[0]]
steps = [rows[cuts[i]:cuts[i + 1]] for i in range(len(cuts) - 1)]
# keep the steps that have the most common dispatch sequence among the last ones (the timed steps)
steps = steps[-(n_last + 1):-1] if len(steps) > n_last + 1 else steps
sig = {}
for s in steps: sig.setdefault(tuple(short(r[0]) for r in s), []).append(s)
seq, same = max(sig.items(), key=lambda kv: len(kv[1]))
out = ["# %s: %d steps with the same %d-dispatch sequence (of the last %d); microseconds" % (os.path.basename(dst), len(same), len(seq), len(steps)),
       "%4s %-56s %9s %9s %9s" % ("#", "kernel", "start", "dur", "gap")]
tot_busy = tot_span = 0.0
agg = {}
for j, name in enumerate(seq):
    st = sum(s[j][1] - s[0][1] for s in same) / len(same) / 1e3
    du = sum(s[j][2] - s[j][1] for s in same) / len(same) / 1e3
    gp = 0.0 if j == 0 else sum(max(0, s[j][1] - max(r[2] for r in s[:j])) for s in same) / len(same) / 1e3
    out.append("%4d %-56s %9.1f %9.1f %9.1f" % (j, name, st, du, gp))
    a = agg.setdefault(name, [0, 0.0, 0.0]); a[0] += 1; a[1] += du; a[2] += gp
for s in same:
    busy, hi = 0.0, s[0][1]
    for _, a, b in s:
        if b > hi: busy += b - max(a, hi); hi = b
    tot_busy += busy / 1e3; tot_span += (hi - s[0][1]) / 1e3
# the span of a step runs to the next step's first dispatch: include the idle time behind the last dispatch
nxt = []
for s in same:
    i = rows.index(s[0]); k = i + len(s)
    if k < len(rows): nxt.append((rows[k][1] - s[0][1]) / 1e3)
out.append("")
out.append("busy (union of dispatches) %.1f us of %.1f us first-to-last dispatch; step period (first dispatch to the next step's) %.1f us"
           % (tot_busy / len(same), tot_span / len(same), sum(nxt) / max(1, len(nxt))))
out.append("")
out.append("%-56s %5s %10s %10s" % ("per kernel", "calls", "dur_us", "gap_us"))
for name, (c, du, gp) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    out.append("%-56s %5d %10.1f %10.1f" % (name, c, du, gp))
out.append("%-56s %5d %10.1f %10.1f" % ("total", sum(a[0] for a in agg.values()), sum(a[1] for a in agg.values()), sum(a[2] for a in agg.values())))
os.makedirs(os.path.dirname(os.path.abspath(dst)), exist_ok=True)
open(dst, "w").write("\n".join(out) + "\n")
print("\n".join(out[-(len(agg) + 6):]))
