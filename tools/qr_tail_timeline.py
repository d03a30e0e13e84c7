"""prints the kernels of the LAST one-pass QR factorization of a rocprofv3 kernel trace (start, end, queue, duration)"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
fh = [r for r in rows if "fh::tq_" in r["Kernel_Name"] or "fh::tu_" in r["Kernel_Name"]]
# a factorization starts with a Gram launch preceded by a gap of more than 300 us
starts = [i for i, r in enumerate(fh) if i == 0 or r["s"] - max(x["e"] for x in fh[max(0, i - 4):i]) > 300000]
last = fh[starts[-1]:]
t0 = last[0]["s"]
for r in last:
    print(f"+{(r['s'] - t0) / 1e3:8.1f} .. +{(r['e'] - t0) / 1e3:8.1f}  q{r['Queue_Id']} {(r['e'] - r['s']) / 1e3:7.1f} us  {r['Kernel_Name'].split('(')[0][-36:]}")
print(f"span {(max(r['e'] for r in last) - t0) / 1e3:.1f} us, kernels {len(last)}, sum of durations {sum(r['e'] - r['s'] for r in last) / 1e3:.1f} us")
