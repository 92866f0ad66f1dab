"""Range markers (roctx, csrc/satba_capi.hip `Range`) of a `rocprofv3 --marker-trace` run: count and mean host-side duration per name.
usage: python tools/rocpd_markers.py <results.db>"""
import json
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
reg = [t for t in tabs if t.startswith("rocpd_region_")][0]
ev = [t for t in tabs if t.startswith("rocpd_event_")][0]
acc = {}
for start, end, ext in db.execute("select r.start, r.end, e.extdata from {} r join {} e on r.event_id = e.id".format(reg, ev)):
    name = json.loads(ext or "{}").get("message")
    if name:
        c = acc.setdefault(name, [0, 0.0])
        c[0] += 1
        c[1] += (end - start) * 1e-3
print("{:<28} {:>7} {:>14}".format("range", "count", "mean host us"))
for name, (n, tot) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    print("{:<28} {:>7} {:>14.1f}".format(name, n, tot / n))
