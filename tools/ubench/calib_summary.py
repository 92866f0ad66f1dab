"""Table of tools/ubench/fetch_calib.bin's PMC passes (tools/gpu.sh TAG calib): counter values of the TIMED launch of every
kernel (the second of each pair) beside the byte counts the program knows."""
import glob, sqlite3, sys

out = sys.argv[1]
names = ["stream16", "gather80", "gather80r", "gather64h", "gather64c1", "gather64c2"]
vals = {}
for db in sorted(glob.glob(out + "/c*/c_results.db")):
    con = sqlite3.connect(db)
    rows = con.execute("select dispatch_id, kernel_name, counter_name, sum(value) from counters_collection "
                       "group by dispatch_id, counter_name order by dispatch_id").fetchall()
    per = {}
    for did, k, c, v in rows:
        if k.startswith("k_") or "k_gather" in k:
            per.setdefault(c, []).append(v)
    for c, v in per.items():
        vals[c] = v[1::2]  # warm-up, timed, warm-up, timed, ...
print(open(out + "/plain.txt").read())
cols = ["FETCH_SIZE", "TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_HIT_sum", "TCC_MISS_sum", "TCP_TCC_READ_REQ_sum"]
print("%-11s" % "launch" + "".join("%24s" % c for c in cols) + "   FETCH_SIZE x 1024 x 2 (MB)")
for i, n in enumerate(names):
    line = "%-11s" % n
    for c in cols:
        v = vals.get(c, [])
        line += "%24.0f" % v[i] if i < len(v) else "%24s" % "-"
    f = vals.get("FETCH_SIZE", [])
    if i < len(f):
        line += "   %10.1f" % (f[i] * 1024 * 2 / 1e6)
    print(line)
