"""Summarise a rocprofv3 kernel-trace CSV: per-kernel busy time and the idle gaps between consecutive kernels."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].split("(")[0].replace("void dpcg::", "").replace("dpcg::", "")[:60]
dur = collections.defaultdict(list); gap_after = collections.defaultdict(list)
for a, b in zip(rows, rows[1:]):
    d = int(a["End_Timestamp"]) - int(a["Start_Timestamp"])
    g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
    if d > 2000:  # ignore the no-op launches after convergence
        dur[name(a)].append(d)
        if g < 50000: gap_after[name(a) + " -> " + name(b)].append(g)
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    v.sort(); print(f"{k:62s} n={len(v):6d} median {v[len(v)//2]/1e3:8.2f} us  mean {sum(v)/len(v)/1e3:8.2f} us")
print()
for k, v in sorted(gap_after.items(), key=lambda kv: -len(kv[1]))[:8]:
    v.sort(); print(f"gap {k:100s} n={len(v):6d} median {v[len(v)//2]/1e3:6.2f} us")
