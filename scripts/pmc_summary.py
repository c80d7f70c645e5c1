"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected in SEPARATE runs, as
MI355X_MICROARCH.md prescribes: FETCH_SIZE costs 3 of the 4 TCC slots, WRITE_SIZE 2).

    python scripts/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01_pmc_traffic.txt profiles/hist_traffic.json

gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports exactly 1/2 of the bytes of a coalesced streaming
read.  Calibrated here on kernels of this code base whose byte counts are known exactly (k_transpose_keys reads N*F*4 B and
writes N*F*4 B; k_bin_cols reads N*F*4 B, writes N*F*2 B): FETCH_SIZE shows 0.500x of the read bytes for both (4 B/lane
coalesced loads), WRITE_SIZE shows 1.000x of the written bytes.  Hence traffic = 2*FETCH_SIZE + WRITE_SIZE (KiB -> bytes)."""
import glob, json, os, re, sqlite3, subprocess, sys, time

fetch_dir, write_dir, out_txt, out_json = sys.argv[1:5]
EXTRA = sys.argv[5] if len(sys.argv) > 5 else ""
try:
    COMMIT = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or "unknown"
    if subprocess.run(["git", "status", "--porcelain", "gbrl_amd"], capture_output=True, text=True).stdout.strip():
        COMMIT += "+dirty"
except Exception:
    COMMIT = "unknown"
STAMP = "commit %s, %s" % (COMMIT, time.strftime("%Y-%m-%d %H:%M"))
if EXTRA:   # on the GPU box there is no .git: the caller passes the build stamp of the shipped library
    STAMP = EXTRA + ", " + time.strftime("%Y-%m-%d %H:%M")
    COMMIT = EXTRA
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n); n = re.sub(r"\(.*$", "", n); return n[:60]
res = {}
for d, ctr in ((fetch_dir, "FETCH_SIZE"), (write_dir, "WRITE_SIZE")):
    db = sorted(glob.glob(d + "/**/*.db", recursive=True))[0]
    cur = sqlite3.connect(db).cursor()
    for k, v, c in cur.execute("select kernel_name, sum(value), count(*) from counters_collection where counter_name=? group by kernel_name", (ctr,)):
        res.setdefault(short(k), {})[ctr] = (v, c)
lines = ["# " + STAMP, "# per-kernel HBM-side traffic, rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate runs of: python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline)",
         "# corrected = 2*FETCH_SIZE + WRITE_SIZE (gfx950: FETCH_SIZE counts 128-B requests at 64 B; calibrated on k_transpose_keys / k_bin_cols, see scripts/pmc_summary.py)",
         "%-48s %6s %16s %16s %18s" % ("kernel", "calls", "FETCH_KiB/call", "WRITE_KiB/call", "corrected_MB/call")]
summary = {}
for k, d in sorted(res.items(), key=lambda kv: -sum(x[0] for x in kv[1].values())):
    f = d.get("FETCH_SIZE", (0, 1)); w = d.get("WRITE_SIZE", (0, 1))
    if f[0] + w[0] < 1000 or not k.startswith("gbrl"): continue
    fpc, wpc = f[0] / max(f[1], 1), w[0] / max(w[1], 1)
    corr = (2 * fpc + wpc) * 1024 / 1e6
    lines.append("%-48s %6d %16.1f %16.1f %18.1f" % (k, f[1], fpc, wpc, corr))
    summary[k] = {"calls": f[1], "fetch_kib_per_call": fpc, "write_kib_per_call": wpc, "corrected_bytes_per_call": corr * 1e6}
open(out_txt, "w").write("\n".join(lines) + "\n")
hb = [v for k, v in summary.items() if "k_hist_build" in k]
if hb:
    json.dump({"kernel": "k_hist_build", "bytes_per_launch": hb[0]["corrected_bytes_per_call"], "launches_per_tree": 6,
               "source": "profiles/rNN_pmc_traffic.txt (scripts/collect_evidence.sh writes it as gpurun_out/evidence/" + os.path.basename(out_txt) + "; copied into profiles/ under the round's name)", "commit": COMMIT, "taken": time.strftime("%Y-%m-%d %H:%M"), "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs; traffic = 2*FETCH + WRITE (gfx950 correction)"},
              open(out_json, "w"), indent=1)
print("\n".join(lines[:14]))
