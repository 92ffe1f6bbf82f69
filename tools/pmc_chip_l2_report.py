"""profiles/<tag>_chip_l2_counters.md + `chip_3d_100_l2_counters` in profiles/pmc_traffic.json from tools/pmc_chip_l2.sh's passes.
    python tools/pmc_chip_l2_report.py <out_dir> <tag>
Every counter is the MEAN over the launches of k_pcg_chip<8,7,true,0> but the first (cold caches), summed over the dimensions the profiler
writes separately (XCDs / shader engines); one launch = one whole solve."""
import collections
import csv
import glob
import json
import pathlib
import shutil
import sys

out_dir, tag = sys.argv[1:3]
ROOT = pathlib.Path(__file__).resolve().parent.parent


def short(name):
    return name.replace("(anonymous namespace)::", "").split("(")[0].replace("void dpcg::", "").replace("dpcg::", "")


vals = collections.defaultdict(lambda: collections.defaultdict(float))       # counter -> dispatch -> value (summed over instances)
for f in glob.glob(f"{out_dir}/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if short(r["Kernel_Name"]).startswith("k_pcg_chip"):
            vals[r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
case = [l.split() for l in open(f"{out_dir}/run.log") if l.startswith("poisson")][0]
its, n, nnz = int(case[1]), int(case[2]), int(case[3])
mean = {}
for c, d in vals.items():
    v = [d[k] for k in sorted(d)]
    v = v[1:] if len(v) > 1 else v
    mean[c] = sum(v) / len(v)
gathers = nnz * its                      # 16-byte granule gathers per launch (one per matrix entry and update)
lines = [f"# L2-level and issue counters of the whole-chip solve kernel ({tag})", "",
         "`rocprofv3 --pmc <group> --kernel-trace --output-format csv -- python3 tools/pmc_chip_l2_run.py`, one pass per group",
         f"(tools/pmc_chip_l2.sh).  Kernel `k_pcg_chip<8,7,true,0>`, system poisson3d_100 (n = {n}, nnz = {nnz}), {its} updates per launch;",
         f"per launch the kernel issues {gathers:,} sixteen-byte granule gathers ({gathers * 16 / 1e9:.2f} GB) and {n * its:,} granule stores.", "",
         "| counter | per launch | per update | per gathered granule |", "|---|---|---|---|"]
for c in sorted(mean):
    lines.append(f"| {c} | {mean[c]:,.0f} | {mean[c] / its:,.0f} | {mean[c] / gathers:.4f} |")
d = {k: round(v) for k, v in mean.items()}
d["updates"] = its
derived = []
if "TCC_HIT_sum" in mean and "TCC_MISS_sum" in mean and mean["TCC_HIT_sum"] + mean["TCC_MISS_sum"] > 0:
    hr = mean["TCC_HIT_sum"] / (mean["TCC_HIT_sum"] + mean["TCC_MISS_sum"])
    d["l2_hit_rate"] = round(hr, 4)
    derived.append(f"* L2 hit rate TCC_HIT / (TCC_HIT + TCC_MISS) = **{hr:.4f}**")
if "TCC_REQ_sum" in mean:
    derived.append(f"* TCC_REQ per update = {mean['TCC_REQ_sum'] / its:,.0f}; a wave's gather of 64 consecutive granules is 1 KiB = 8 requests of 128 B: "
                   f"{nnz / 64 * 8:,.0f} expected per update from the gathers alone")
    d["tcc_req_per_update"] = round(mean["TCC_REQ_sum"] / its)
if "SQ_BUSY_CYCLES" in mean and "SQ_WAIT_ANY" in mean and "SQ_WAVE_CYCLES" in mean and mean["SQ_WAVE_CYCLES"] > 0:
    derived.append(f"* SQ_WAIT_ANY / SQ_WAVE_CYCLES = {mean['SQ_WAIT_ANY'] / mean['SQ_WAVE_CYCLES']:.3f} of a wave's resident cycles are spent waiting")
    d["wait_fraction_of_wave_cycles"] = round(mean["SQ_WAIT_ANY"] / mean["SQ_WAVE_CYCLES"], 4)
if "SQ_INSTS_VALU" in mean:
    derived.append(f"* SQ_INSTS_VALU per update and wave = {mean['SQ_INSTS_VALU'] / its / 2048:,.0f} (2 048 waves)")
lines += [""] + derived
stats = glob.glob(f"{out_dir}/trace/**/*kernel_stats.csv", recursive=True)
if stats:
    shutil.copy(stats[0], ROOT / "profiles" / f"{tag}_chip_kernel_stats.csv")
    rows = {short(r["Name"]): r for r in csv.DictReader(open(stats[0]))}
    lines += ["", f"Kernel trace of the three variants (`--variants`: DPCG_CHIP_BENCH unset / 1 / 3), `profiles/{tag}_chip_kernel_stats.csv`:", "",
              "| kernel | calls | average ns | per update us |", "|---|---|---|---|"]
    for k, r in rows.items():
        if k.startswith("k_pcg_chip"):
            lines.append(f"| {k} | {r['Calls']} | {float(r['AverageNs']):,.0f} | {float(r['AverageNs']) / its / 1e3:.3f} |")
(ROOT / "profiles" / f"{tag}_chip_l2_counters.md").write_text("\n".join(lines) + "\n")
pj = ROOT / "profiles" / "pmc_traffic.json"
allj = json.loads(pj.read_text()) if pj.exists() else {}
allj["chip_3d_100_l2_counters"] = d
pj.write_text(json.dumps(allj, indent=1) + "\n")
print("\n".join(lines))
