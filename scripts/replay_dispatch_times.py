import glob, sqlite3, sys
db = sorted(glob.glob(sys.argv[1] + "/**/*.db", recursive=True))[0]
cur = sqlite3.connect(db).cursor()
rows = cur.execute("select name, start, end from kernels order by start").fetchall()
out = []
for n, a, b in rows:
    for key in ("k_seq_stitch", "k_seq_summary", "k_near_list", "k_near_rows", "k_near_sides", "k_near_replay"):
        if key in n:
            out.append((key, (b - a) / 1000.0))
# print the first tree of each config: sequences until 'k_near_list' count 6
seq = []
lists = 0
for k, d in out:
    if k == "k_near_list":
        lists += 1
        if lists in (1, 19): seq.append("\n== tree (list #%d)" % lists)
    if lists <= 6 or 19 <= lists <= 24:
        seq.append("%s %.0f" % (k.replace("k_", ""), d))
print(" | ".join(seq))
