"""Average PMC counter values per kernel from rocprofv3 --pmc CSV output (counter_collection.csv files)."""
import collections
import csv
import glob
import re
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            name = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "")
            acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for k in sorted(acc):
    if pat and not re.search(pat, k):
        continue
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print("    {:32s} avg {:16.1f}   n {}".format(c, sum(v) / len(v), len(v)))
