"""Per-level HBM-side traffic table of k_hist_build from several `rocprofv3 --pmc` runs (scripts/hist_traffic_levels.sh).

    python scripts/hist_traffic_table.py out.txt out.json "<stamp>" dir1 [dir2 ...]

Dispatches of the kernel with the largest grid (the bench shape) are ordered by dispatch id; position modulo 6 = tree level.
Per level: duration, FETCH_SIZE / WRITE_SIZE as reported (KiB), the raw request counts by size where the device exposes them, and

    bytes_req  = 32 n32 + 64 n64 + 128 n128          (request-size counters; no correction factor involved)
    bytes_x2   = 2 * FETCH_SIZE + WRITE_SIZE         (the guide's correction for 16-B-per-lane coalesced streams)
    factor     = read bytes_req / FETCH_SIZE         (what FETCH_SIZE has to be multiplied by ON THIS KERNEL, level by level)
    known      = the bytes level 0 must move: N (2 F + 4 D) read (contiguous rows, 16-bit codes + int32 gradients), F 257 (D+1) 4 written

The json carries the per-level list the bench line repeats (`roofline.traffic_levels`) and the per-launch mean as `bytes_per_launch`."""
import glob, json, sqlite3, sys, time

out_txt, out_json, stamp = sys.argv[1:4]
LEVELS = 6
N, F, D = 1 << 20, 128, 8
per = {}     # counter -> [per-level mean]
dur = {}
for d in sys.argv[4:]:
    for db in sorted(glob.glob(d + "/**/*.db", recursive=True)):
        cur = sqlite3.connect(db).cursor()
        try:
            rows = cur.execute("select dispatch_id, counter_name, sum(value), max(end - start), max(grid_size) from counters_collection "
                               "where kernel_name like '%k_hist_build%' group by dispatch_id, counter_name order by dispatch_id").fetchall()
        except sqlite3.Error:
            continue
        if not rows:
            continue
        gmax = max(r[4] for r in rows)
        rows = [r for r in rows if r[4] == gmax]
        ids = sorted({r[0] for r in rows})
        pos = {i: k % LEVELS for k, i in enumerate(ids)}
        for c in sorted({r[1] for r in rows}):
            acc = [[] for _ in range(LEVELS)]
            dd = [[] for _ in range(LEVELS)]
            for i, cc, v, du, g in rows:
                if cc == c:
                    acc[pos[i]].append(v); dd[pos[i]].append(du / 1000.0)
            per[c] = [sum(a) / max(1, len(a)) for a in acc]
            dur[c] = [sum(a) / max(1, len(a)) for a in dd]
def get(name):
    for k in (name, name + "_sum"):
        if k in per:
            return per[k]
    return None
fetch, write = get("FETCH_SIZE"), get("WRITE_SIZE")
rd, r32, r64, r128, bub = get("TCC_EA0_RDREQ"), get("TCC_EA0_RDREQ_32B"), get("TCC_EA0_RDREQ_64B"), get("TCC_EA0_RDREQ_128B"), get("TCC_BUBBLE")
wr, w64 = get("TCC_EA0_WRREQ"), get("TCC_EA0_WRREQ_64B")
req, hit, miss = get("TCC_REQ"), get("TCC_HIT"), get("TCC_MISS")
us = dur.get("FETCH_SIZE") or next(iter(dur.values()))
lines = ["# " + stamp + ", " + time.strftime("%Y-%m-%d %H:%M"),
         "# k_hist_build per tree level at 2^20 x 128, D = 8 (bench shape): rocprofv3 --pmc, one run per counter set (scripts/hist_traffic_levels.sh)",
         "# counters found: " + " ".join(sorted(per))]
hdr = "%-34s" % "quantity" + "".join("%14s" % ("level %d" % l) for l in range(LEVELS)) + "%14s" % "mean/launch"
lines.append(hdr)
def row(name, vals, fmt="%14.1f"):
    if vals is None:
        lines.append("%-34s" % name + "   (counter not available on this device)")
        return
    lines.append("%-34s" % name + "".join(fmt % v for v in vals) + fmt % (sum(vals) / len(vals)))
row("duration_us (under --pmc)", us)
row("FETCH_SIZE KiB (as reported)", fetch)
row("WRITE_SIZE KiB (as reported)", write)
row("TCC_EA0_RDREQ", rd); row("TCC_EA0_RDREQ_32B", r32); row("TCC_EA0_RDREQ_64B", r64); row("TCC_EA0_RDREQ_128B", r128); row("TCC_BUBBLE", bub)
row("TCC_EA0_WRREQ", wr); row("TCC_EA0_WRREQ_64B", w64)
row("TCC_REQ", req); row("TCC_HIT", hit); row("TCC_MISS", miss)
if hit and miss:
    row("L2 hit rate", [h / max(1.0, h + m) for h, m in zip(hit, miss)], "%14.3f")
known_rd = N * (2 * F + 4 * D)
known_wr = [(1 << max(0, l - 1)) * F * 257 * (D + 1) * 4 if l else F * 257 * (D + 1) * 4 for l in range(LEVELS)]   # smaller children only (sibling subtraction)
alg = [N * (F + 4 * D + 4) + (1 << l) * F * 256 * (D + 1) * 4 for l in range(LEVELS)]                                # SURVEY 8d, per level
read_req = None
if rd and r32 is not None:
    if r128 and r64:
        read_req = [32 * a + 64 * b + 128 * c for a, b, c in zip(r32, r64, r128)]
        how = "32 n32 + 64 n64 + 128 n128"
    elif bub:
        read_req = [32 * a + 128 * c + 64 * (t - a - c) for t, a, c in zip(rd, r32, bub)]
        how = "32 n32 + 128 bubble + 64 rest (the FETCH_SIZE expression of counter_defs.yaml)"
    else:
        read_req = None
levels = []
if fetch and write:
    x2 = [(2 * f + w) * 1024 for f, w in zip(fetch, write)]
    row("bytes_x2 = 2 FETCH + WRITE (MB)", [b / 1e6 for b in x2])
    if read_req:
        row("read bytes_req (MB): " + how[:12], [b / 1e6 for b in read_req])
        row("factor = read bytes_req / FETCH", [b / max(1.0, f * 1024) for b, f in zip(read_req, fetch)], "%14.3f")
    lines.append("%-34s%14.1f   (N (2F + 4D); FETCH_SIZE x factor must reproduce it: factor0 = %.3f)" % ("level 0 known read MB", known_rd / 1e6, known_rd / max(1.0, fetch[0] * 1024)))
    f0 = known_rd / max(1.0, fetch[0] * 1024)
    row("algorithmic MB (SURVEY 8d)", [a / 1e6 for a in alg])
    best = [(r if read_req else 2 * f * 1024) + w * 1024 for r, f, w in zip(read_req or [0] * LEVELS, fetch, write)]
    row("traffic MB (read bytes_req + WRITE)" if read_req else "traffic MB (2 FETCH + WRITE)", [b / 1e6 for b in best])
    row("traffic / algorithmic", [b / a for b, a in zip(best, alg)], "%14.2f")
    row("traffic TB/s", [b / (u * 1e-6) / 1e12 for b, u in zip(best, us)], "%14.2f")
    for l in range(LEVELS):
        levels.append({"level": l, "us_under_pmc": round(us[l], 1), "fetch_kib": round(fetch[l], 1), "write_kib": round(write[l], 1),
                       "traffic_bytes": round(best[l]), "algorithmic_bytes": alg[l], "factor_on_fetch": round((read_req[l] if read_req else 2 * fetch[l] * 1024) / max(1.0, fetch[l] * 1024), 3)})
    json.dump({"kernel": "k_hist_build", "bytes_per_launch": sum(best) / LEVELS, "launches_per_tree": LEVELS, "levels": levels,
               "level0_known_read_bytes": known_rd, "level0_factor_from_known_bytes": round(f0, 3),
               "method": ("rocprofv3 --pmc, separate runs: read bytes from the request-size counters (" + how + "), written bytes = WRITE_SIZE; FETCH_SIZE factor checked on level 0's known bytes"
                          if read_req else "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs; traffic = 2*FETCH + WRITE (gfx950 correction; request-size counters unavailable)"),
               "source": "profiles/r06_hist_levels_traffic.txt (scripts/hist_traffic_levels.sh on the GPU box)", "commit": stamp, "taken": time.strftime("%Y-%m-%d %H:%M")},
              open(out_json, "w"), indent=1)
open(out_txt, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
