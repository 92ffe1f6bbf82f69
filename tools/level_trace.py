"""Per-launch durations of the triangular-solve kernels of ONE PCG update from a rocprofv3 kernel-trace CSV (tools/trace_run_c3.py)."""
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].split("(")[0].replace("void dpcg::", "").replace("dpcg::", "")[:44]
# the last complete update: from the last k_spmv_* launch with CTL (the loop's K1) backwards one update
idx = [i for i, r in enumerate(rows) if name(r).startswith("k_spmv_tile") or name(r).startswith("k_spmv_stream")]
lo, hi = idx[-4], idx[-3]
t0 = int(rows[lo]["Start_Timestamp"])
tot = 0
for r in rows[lo:hi]:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    tot += d
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.2f} us  +{d / 1e3:7.2f} us  grid {r.get('Grid_Size', '?'):>8s}  {name(r)}")
print(f"update: {(int(rows[hi]['Start_Timestamp']) - t0) / 1e3:.1f} us wall, {tot / 1e3:.1f} us in kernels, {hi - lo} launches")
