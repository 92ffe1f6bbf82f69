"""Memory-side traffic per launch of the kernels of ONE PCG update with IC(0) in multicolour order, from two rocprofv3 --pmc passes
(FETCH_SIZE, WRITE_SIZE) over tools/trace_run_mc.py; units as profiles/r03_pmc_summary.md calibrates them (read = 2 x FETCH_SIZE KiB,
write = WRITE_SIZE KiB).   python tools/pmc_sweeps_report.py <dir with FETCH_SIZE pass> <dir with WRITE_SIZE pass>"""
import collections, csv, glob, statistics, sys


def per_kernel(root, counter):
    out = collections.defaultdict(list)
    for f in glob.glob(f"{root}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"].replace("void dpcg::", "").replace("dpcg::", "").split("(")[0]
            out[name].append(float(r["Counter_Value"]))
    return out


rd, wr = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
print("| kernel | launches | read MB per launch (2 x FETCH_SIZE x 1024) | write MB per launch (WRITE_SIZE x 1024) |")
print("|---|---|---|---|")
for k in sorted(rd):
    if not k.startswith(("k_lm_sweep", "k_update_r", "k_update_xp", "k_spmv_tile")):
        continue
    a = [v for v in rd[k] if v > 0.2 * statistics.median(rd[k])]            # (drop the no-op launches after convergence)
    b = [v for v in wr.get(k, [0.0]) if v > 0.2 * statistics.median(wr.get(k, [1.0]))] or [0.0]
    print(f"| `{k}` | {len(a)} | {2 * statistics.mean(a) * 1024 / 1e6:.1f} | {statistics.mean(b) * 1024 / 1e6:.1f} |")
