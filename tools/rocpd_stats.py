"""Per-kernel statistics (calls, total / average / min / max duration) from a rocprofv3 rocpd sqlite database."""
import re
import sqlite3
import sys


def main(path, out=sys.stdout):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    disp = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
    sym = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
    cols = [r[1] for r in c.execute("pragma table_info({})".format(disp))]
    scols = [r[1] for r in c.execute("pragma table_info({})".format(sym))]
    name_col = "display_name" if "display_name" in scols else "kernel_name"
    start, end = ("start", "end") if "start" in cols else ("start_timestamp", "end_timestamp")
    q = ("select s.{n}, count(*), sum(d.{e} - d.{s}), avg(d.{e} - d.{s}), min(d.{e} - d.{s}), max(d.{e} - d.{s}) "
         "from {d} d join {y} s on d.kernel_id = s.id group by s.{n} order by 3 desc").format(
             n=name_col, s=start, e=end, d=disp, y=sym)
    rows = c.execute(q).fetchall()
    total = sum(r[2] for r in rows) or 1
    out.write("{:<70s} {:>7s} {:>12s} {:>11s} {:>11s} {:>11s} {:>6s}\n".format(
        "kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"))
    for name, n, tot, avg, mn, mx in rows:
        name = re.sub(r"\(.*", "", name)
        out.write("{:<70s} {:>7d} {:>12.1f} {:>11.2f} {:>11.2f} {:>11.2f} {:>6.2f}\n".format(
            name[:70], n, tot / 1e3, avg / 1e3, mn / 1e3, mx / 1e3, 100.0 * tot / total))


if __name__ == "__main__":
    main(sys.argv[1])
