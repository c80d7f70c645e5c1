#!/bin/bash
# Run on the GPU box: which unit bounds each kernel of a bench-shape step?  Two --pmc passes (separate runs), then per kernel: duration,
# VALU / SALU / LDS / VMEM instruction counts per wave and the share of the kernel's SIMD-time that the VALU and the LDS unit were active.
#   bash scripts/step_pmc.sh out.txt
set -u
R="$GRAFT_REPO_ROOT"; W=/tmp/gbrl_step_pmc; rm -rf "$W"; mkdir -p "$W" "$(dirname "$R/$1")"
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-extra-legs --steps 6 --warmup 2 --large-ensemble 32"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_BUSY_CYCLES -d "$W/a" -o a -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT -d "$W/b" -o b -- $B > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d "$W/t" -o t -- $B > /dev/null 2>&1
python3 - "$W" "$R/$1" <<'PY'
import glob, re, sqlite3, sys
W, dst = sys.argv[1], sys.argv[2]
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n); n = re.sub(r"\(.*$", "", n); return n.replace("gbrl::kern::", "")[:44]
cnt = {}
for d in ("a", "b"):
    for db in glob.glob(W + "/" + d + "/**/*.db", recursive=True):
        cur = sqlite3.connect(db).cursor()
        for k, c, v, n in cur.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name"):
            cnt.setdefault(short(k), {})[c] = v / max(n, 1)
dur = {}
for db in glob.glob(W + "/t/**/*.db", recursive=True):
    cur = sqlite3.connect(db).cursor()
    for n, c, t, a, p in cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
        dur[short(n)] = (a / 1e3 if a > 1e3 else a, c, p)
out = ["# per launch (mean over the run): duration, instructions per wave, and the share of the kernel's SIMD-time in which the unit had an",
       "# instruction active (SQ_ACTIVE_INST_* are quad-cycles summed over 1024 SIMDs: x 4 / (1024 x duration x 2.4 GHz)); bench shape 2^20 x 128",
       "%-44s %8s %7s %7s %7s %6s %6s %7s %7s %7s" % ("kernel", "us", "waves", "valu/w", "salu/w", "lds/w", "vmem/w", "valu%", "lds%", "wait%")]
for k, (us, calls, pct) in sorted(dur.items(), key=lambda kv: -kv[1][2]):
    c = cnt.get(k)
    if not c or pct < 0.3: continue
    w = max(c.get("SQ_WAVES", 1), 1)
    simd_cyc = 1024 * us * 1e-6 * 2.4e9
    out.append("%-44s %8.1f %7.0f %7.0f %7.0f %6.0f %6.0f %6.0f%% %6.0f%% %6.0f%%" % (k, us, w, c.get("SQ_INSTS_VALU", 0) / w, c.get("SQ_INSTS_SALU", 0) / w, c.get("SQ_INSTS_LDS", 0) / w,
               (c.get("SQ_INSTS_VMEM_RD", 0) + c.get("SQ_INSTS_VMEM_WR", 0)) / w, 100 * 4 * c.get("SQ_ACTIVE_INST_VALU", 0) / simd_cyc, 100 * 4 * c.get("SQ_ACTIVE_INST_LDS", 0) / simd_cyc,
               100 * c.get("SQ_WAIT_INST_ANY", 0) / max(c.get("SQ_WAVE_CYCLES", 1), 1)))
open(dst, "w").write("\n".join(out) + "\n"); print("\n".join(out))
PY
