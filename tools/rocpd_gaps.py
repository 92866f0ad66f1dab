"""Idle time between consecutive kernels of a rocprofv3 rocpd database: for every kernel name the average gap in front of it
(start - end of the previous dispatch), over the last `frac` of the trace (the timed loop of bench.py).  usage: rocpd_gaps.py DB [frac]"""
import re
import sqlite3
import sys
from collections import defaultdict


def main(path, frac=0.5):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    disp = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
    sym = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
    cols = [r[1] for r in c.execute("pragma table_info({})".format(disp))]
    scols = [r[1] for r in c.execute("pragma table_info({})".format(sym))]
    name_col = "display_name" if "display_name" in scols else "kernel_name"
    start, end = ("start", "end") if "start" in cols else ("start_timestamp", "end_timestamp")
    rows = c.execute("select s.{n}, d.{s}, d.{e} from {d} d join {y} s on d.kernel_id = s.id order by d.{s}".format(
        n=name_col, s=start, e=end, d=disp, y=sym)).fetchall()
    rows = rows[int(len(rows) * (1.0 - frac)):]
    gaps, durs = defaultdict(list), defaultdict(list)
    prev_end = None
    for name, s, e in rows:
        name = re.sub(r"\(.*", "", name)[:60]
        if prev_end is not None:
            gaps[name].append(max(0, s - prev_end))
        durs[name].append(e - s)
        prev_end = max(prev_end or e, e)
    span = rows[-1][2] - rows[0][1]
    busy = sum(sum(v) for v in durs.values())
    print("dispatches {}  span {:.3f} ms  kernel time {:.3f} ms  idle {:.3f} ms ({:.1f} %)".format(len(rows), span / 1e6, busy / 1e6, (span - busy) / 1e6,
                                                                                                 100.0 * (span - busy) / span))
    print("{:<62s} {:>6s} {:>9s} {:>9s} {:>10s}".format("kernel", "calls", "avg_us", "gap_us", "gap_tot_ms"))
    for name in sorted(gaps, key=lambda k: -sum(gaps[k])):
        g = gaps[name]
        print("{:<62s} {:>6d} {:>9.2f} {:>9.2f} {:>10.3f}".format(name, len(g), sum(durs[name]) / len(durs[name]) / 1e3, sum(g) / len(g) / 1e3, sum(g) / 1e6))


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.5)
