"""Offline analysis of a rocprofv3 kernel-trace csv of one look-ahead factorization: per-queue busy time, idle gaps,
and the largest kernels (used to see which stream bounds LLT / LU).  usage: trace_timeline.py <csv> [iteration]"""
import csv
import collections
import os
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    r["fh"] = "fh::" in r["Kernel_Name"]
rows.sort(key=lambda r: r["s"])
# split into factorizations: a gap of non-fh kernels (the reset copy) separates them
runs, cur = [], []
for r in rows:
    if r["fh"]:
        cur.append(r)
    elif cur and "copy" in r["Kernel_Name"].lower() or (cur and "elementwise" in r["Kernel_Name"]):
        runs.append(cur)
        cur = []
if cur:
    runs.append(cur)
runs = [x for x in runs if len(x) > 50]
if os.environ.get("TL_WHOLE"):  # the whole trace as one run (drivers whose steps contain copy kernels, e.g. the distributed LU);
    # TL_WHOLE=2: the trace holds two identical runs (cold + warm), analyse the second
    allk = [r for r in rows if r["fh"]]
    runs = [allk[len(allk) // 2:]] if os.environ["TL_WHOLE"] == "2" else [allk]
print("factorizations found:", len(runs), [len(x) for x in runs])
it = int(sys.argv[2]) if len(sys.argv) > 2 else len(runs) - 1
run = runs[it]
t0, t1 = min(r["s"] for r in run), max(r["e"] for r in run)
print(f"iteration {it}: {len(run)} kernels, span {(t1 - t0) / 1e6:.2f} ms")
byq = collections.defaultdict(list)
for r in run:
    byq[r["Queue_Id"]].append(r)


def short(n):
    n = n.replace("void fh::", "").split("(")[0]
    return n[:60]


for q, ks in byq.items():
    busy = sum(k["e"] - k["s"] for k in ks)
    print(f"\nqueue {q}: {len(ks)} kernels, busy {busy / 1e6:.2f} ms, first start +{(ks[0]['s'] - t0) / 1e6:.2f} ms, last end +{(ks[-1]['e'] - t0) / 1e6:.2f} ms")
    agg = collections.defaultdict(lambda: [0, 0])
    for k in ks:
        a = agg[short(k["Kernel_Name"]) + f" wg{k['Workgroup_Size_X']}"]
        a[0] += 1
        a[1] += k["e"] - k["s"]
    for name, (cnt, tot) in sorted(agg.items(), key=lambda x: -x[1][1])[:8]:
        print(f"   {name:75s} x{cnt:5d} {tot / 1e6:8.2f} ms  avg {tot / cnt / 1e3:8.1f} us")
    # idle gaps > 100 us
    gaps = [(ks[i + 1]["s"] - ks[i]["e"], ks[i]["e"] - t0, short(ks[i + 1]["Kernel_Name"])) for i in range(len(ks) - 1)]
    big = [g for g in gaps if g[0] > 100e3]
    print(f"   idle inside: total {sum(g[0] for g in gaps if g[0] > 0) / 1e6:.2f} ms; gaps > 100 us: {len(big)} totalling {sum(g[0] for g in big) / 1e6:.2f} ms")
    for g in big[:40]:
        print(f"      gap {g[0] / 1e3:8.1f} us at +{g[1] / 1e6:7.2f} ms before {g[2]}")
if len(sys.argv) > 3:  # dump the big kernels of every queue in time order
    for r in run:
        d = r["e"] - r["s"]
        if d > float(sys.argv[3]) * 1e3:
            print(f"+{(r['s'] - t0) / 1e6:8.3f} ms q{r['Queue_Id']} {d / 1e3:9.1f} us grid {r['Grid_Size_X']:>8s} {short(r['Kernel_Name'])}")
if os.environ.get("TL_WINDOW"):  # every kernel of every queue inside [a, b] ms of the run, in time order
    wa, wb = (float(x) for x in os.environ["TL_WINDOW"].split(","))
    for r in run:
        ts = (r["s"] - t0) / 1e6
        if wa <= ts <= wb:
            print(f"+{ts:8.3f} .. +{(r['e'] - t0) / 1e6:8.3f} ms q{r['Queue_Id']} {(r['e'] - r['s']) / 1e3:8.1f} us grid {r['Grid_Size_X']:>8s} {short(r['Kernel_Name'])}")
