"""Dump the per-kernel summary (calls, total/avg duration, %) of a rocprofv3 rocpd .db to CSV (rocprofv3 --kernel-trace --stats)."""
import csv, sqlite3, sys
db, out = sys.argv[1], sys.argv[2]
cur = sqlite3.connect(db).cursor()
rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
    for r in rows:
        w.writerow([r[0], r[1], "%.3f" % r[2], "%.3f" % r[3], "%.4f" % r[4]])
print("wrote", out, len(rows), "kernels")
