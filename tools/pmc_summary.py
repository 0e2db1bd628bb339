"""Summarise a rocprofv3 --pmc run (rocpd .db): per kernel name, dispatch count and the mean / total of each collected counter.
usage: pmc_summary.py results.db out.csv"""
import collections
import csv
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
con = sqlite3.connect(db)
cur = con.cursor()
tables = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
view = "counters_collection" if "counters_collection" in tables else None
if view is None:
    print("no counters_collection view; tables:", tables)
    sys.exit(1)
cols = [r[1] for r in cur.execute("pragma table_info(%s)" % view)]
kcol = "kernel_name" if "kernel_name" in cols else [c for c in cols if "kernel" in c][0]
ncol = "counter_name" if "counter_name" in cols else [c for c in cols if "name" in c and c != kcol][0]
vcol = "value" if "value" in cols else [c for c in cols if "value" in c][0]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for k, n, v in cur.execute("select %s, %s, %s from %s" % (kcol, ncol, vcol, view)):
    a = agg[k][n]
    a[0] += 1
    a[1] += float(v)
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Kernel", "Counter", "Dispatches", "Total", "MeanPerDispatch"])
    for k in sorted(agg, key=lambda k: -max(x[1] for x in agg[k].values())):
        for n, (c, t) in agg[k].items():
            w.writerow([k, n, c, "%.3f" % t, "%.3f" % (t / c)])
print("wrote", out, len(agg), "kernels")
