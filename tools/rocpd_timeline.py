"""Time line of the Schur / solve phase from a rocprofv3 rocpd database: for every launch of the tile Cholesky, where the kernels of
the phase start and end relative to k_vinv's start (medians over the launches; microseconds)."""
import re
import sqlite3
import statistics
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
    rows = c.execute("select s.{n}, d.{s}, d.{e} from {d} d join {y} s on d.kernel_id = s.id order by d.{s}".format(
        n=name_col, s=start, e=end, d=disp, y=sym)).fetchall()
    rows = [(re.sub(r"\(.*", "", n), a, b) for n, a, b in rows]
    names = ["k_vinv", "k_schur_diag<", "k_schur_finish", "k_schur_pairs<", "k_chol_tiles", "k_trsv_back_mw", "k_unscale",
             "k_backsub"]
    phases = []
    for idx, (n, a, b) in enumerate(rows):
        if "k_vinv" not in n:
            continue
        ph = {"k_vinv": (0.0, (b - a) / 1e3)}
        for n2, a2, b2 in rows[max(0, idx - 4):idx + 14]:
            for key in names[1:]:
                if key in n2 and key not in ph and b2 > a:
                    ph[key] = ((a2 - a) / 1e3, (b2 - a) / 1e3)
        # (the weighted Schur complement has no k_schur_diag launch: its diagonal items run inside the pair kernel)
        if all(k in ph for k in names if k not in ("k_unscale", "k_schur_diag<")):
            phases.append(ph)
    if not phases:
        out.write("no Schur / solve phase with the tile Cholesky in this trace\n")
        return
    out.write("{} phases\n{:<22s} {:>9s} {:>9s} {:>9s}\n".format(len(phases), "kernel", "start", "end", "length"))
    for k in names:
        if not all(k in p for p in phases):
            continue
        s = statistics.median(p[k][0] for p in phases)
        e = statistics.median(p[k][1] for p in phases)
        out.write("{:<22s} {:>9.1f} {:>9.1f} {:>9.1f}\n".format(k.rstrip("<"), s, e, statistics.median(p[k][1] - p[k][0] for p in phases)))
    tails = sorted(p["k_chol_tiles"][1] - p["k_schur_pairs<"][1] for p in phases)
    heads = sorted(p["k_chol_tiles"][0] for p in phases)
    q = lambda v, f: v[min(len(v) - 1, int(f * len(v)))]
    out.write("factorisation end - pair kernel end: min {:.1f} median {:.1f} p90 {:.1f} max {:.1f}\n".format(tails[0], q(tails, 0.5), q(tails, 0.9), tails[-1]))
    out.write("factorisation start - k_vinv start:  min {:.1f} median {:.1f} p90 {:.1f} max {:.1f}\n".format(heads[0], q(heads, 0.5), q(heads, 0.9), heads[-1]))


if __name__ == "__main__":
    main(sys.argv[1])
