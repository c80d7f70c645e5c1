#!/bin/bash
# Run on the GPU box: kernel table of an RL-sized step (rocprofv3 --kernel-trace --stats), calls per step.
#   bash scripts/small_step_profile.sh N F D depth policy out.txt
set -u
R="$GRAFT_REPO_ROOT"; W=/tmp/gbrl_small_prof; rm -rf "$W"; mkdir -p "$W" "$R/gpurun_out/evidence"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$W/t" -o t -- python3 "$R/scripts/small_step_trace.py" "$1" "$2" "$3" "$4" "$5" > "$W/out.txt" 2>&1
python3 - "$W/t" "$R/$6" "$1 rows x $2 features, D $3, depth $4, $5: 120 steps; $(tail -1 $W/out.txt)" <<'PY'
import glob, os, re, sqlite3, sys
src, dst, title = sys.argv[1], sys.argv[2], sys.argv[3]
db = sorted(glob.glob(os.path.join(src, "**", "*.db"), recursive=True))[0]
cur = sqlite3.connect(db).cursor()
rows = cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n); n = re.sub(r"\(.*$", "", n); return n[:70]
out = ["# " + title, "%-72s %8s %10s %10s %7s" % ("kernel", "calls/step", "avg_us", "us/step", "pct")]
tot = 0.0; nk = 0.0
for n, c, t, a, p in rows:
    out.append("%-72s %8.2f %10.2f %10.2f %6.2f%%" % (short(n), c / 120.0, a / 1e3 if a > 1e3 else a, t / 120.0 / (1e3 if a > 1e3 else 1), p))
    tot += t / 120.0; nk += c / 120.0
out.append("total kernel time per step %.1f (raw units), launches per step %.1f" % (tot, nk))
open(dst, "w").write("\n".join(out) + "\n"); print("\n".join(out))
PY
