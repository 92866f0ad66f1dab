"""Average PMC counter values per kernel from rocprofv3 --pmc output: counter_collection.csv files or rocpd sqlite
databases (view counters_collection; the rows of one dispatch and counter -- one per XCD / instance -- are summed)."""
import collections
import csv
import glob
import re
import sqlite3
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            name = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "")
            acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
for path in glob.glob(sys.argv[1] + "/**/*_results.db", recursive=True):
    c = sqlite3.connect(path)
    try:
        rows = c.execute("select kernel_name, counter_name, dispatch_id, sum(value) from counters_collection "
                         "group by kernel_name, counter_name, dispatch_id").fetchall()
    except sqlite3.Error:
        continue
    for kname, cname, _, v in rows:
        name = re.sub(r"\(.*", "", kname).replace("void ", "")
        acc[name][cname].append(float(v))
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for k in sorted(acc):
    if pat and not re.search(pat, k):
        continue
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print("    {:32s} avg {:16.1f}   n {}".format(c, sum(v) / len(v), len(v)))
