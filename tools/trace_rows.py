"""Rows of a rocprofv3 kernel trace whose kernel name matches any of the comma separated substrings: start (ms from the first
row printed), duration, grid, name."""
import csv
import sys

pats = sys.argv[2].split(",")
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        if any(p in r["Kernel_Name"] for p in pats):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Grid_Size_Y", ""), r["Kernel_Name"][:60]))
rows.sort()
t0 = rows[0][0] if rows else 0
for s, e, gx, gy, nm in rows:
    print(f"+{(s - t0) / 1e6:9.3f} ms {(e - s) / 1e3:8.1f} us grid {gx:>7} x {gy:>3} {nm}")
