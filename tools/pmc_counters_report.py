"""Per-kernel means of every counter collected by tools/pmc_spmv_counters.sh (one rocprofv3 --pmc pass per counter group).

    python tools/pmc_counters_report.py gpurun_out/r03_pmc_spmv > profiles/r03_spmv_tile_counters.md

Rows: counters; columns: the kernels of the 256^3 PCG loop.  No-op launches (after convergence / below 20 % of the
kernel's median duration) are dropped.  Durations come from the kernel trace of the same pass (PMC passes serialise
dispatches, so they are close to the un-profiled ones but not identical)."""
import collections
import csv
import glob
import statistics
import sys

root = sys.argv[1]
KEEP = ("k_spmv_tile", "k_spmv_stream", "k_update_r", "k_update_xp_deferred", "k_stream_bench", "k_sptrsv", "k_lm_finish")


def short(name):
    n = name.replace("void dpcg::", "").replace("dpcg::", "")
    return n.split("(")[0]


table = collections.defaultdict(dict)     # counter -> kernel -> mean
dur = collections.defaultdict(list)
for p in sorted(glob.glob(f"{root}/pass*")):
    if not p.split("/")[-1].startswith("pass") or p.endswith(".log"):
        continue
    traces = glob.glob(f"{p}/**/*kernel_trace.csv", recursive=True)
    d_by_id = {}
    for f in traces:
        for r in csv.DictReader(open(f)):
            d_by_id[r["Dispatch_Id"]] = (short(r["Kernel_Name"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    med = collections.defaultdict(list)
    for k, us in d_by_id.values():
        med[k].append(us)
    med = {k: statistics.median(v) for k, v in med.items()}
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{p}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if not k.startswith(KEEP):
                continue
            did = r["Dispatch_Id"]
            if did in d_by_id and d_by_id[did][1] < 0.2 * med.get(k, 0):
                continue
            vals[r["Counter_Name"]][k].append(float(r["Counter_Value"]))
    for k, us in d_by_id.values():
        if k.startswith(KEEP) and us >= 0.2 * med[k]:
            dur[k].append(us)
    for c, per in vals.items():
        for k, v in per.items():
            table[c][k] = sum(v) / len(v)
kernels = sorted({k for per in table.values() for k in per})
print("| counter | " + " | ".join(kernels) + " |")
print("|---|" + "---|" * len(kernels))
print("| median duration in the PMC passes, us | " + " | ".join(f"{statistics.median(dur[k]):.1f}" if dur[k] else "" for k in kernels) + " |")
for c in table:
    print(f"| {c} | " + " | ".join(f"{table[c][k]:.4g}" if k in table[c] else "" for k in kernels) + " |")
