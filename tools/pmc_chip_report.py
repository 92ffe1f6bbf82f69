"""profiles/<tag>_chip_traffic.md + the `chip_*` entries of profiles/pmc_traffic.json from tools/pmc_chip.sh's passes.
    python tools/pmc_chip_report.py <out_dir> <tag>
Read bytes: 128 x RDREQ_128B + 64 x RDREQ_64B + 32 x RDREQ_32B (2 x FETCH_SIZE x 1024 beside it: MI355X_MICROARCH.md's gfx950 correction,
calibrated in profiles/r01_pmc_summary.md); write bytes: WRITE_SIZE x 1024.  Per LAUNCH of k_pcg_chip = one whole solve."""
import collections
import csv
import glob
import json
import pathlib
import shutil
import sys

out_dir, tag = sys.argv[1:3]
ROOT = pathlib.Path(__file__).resolve().parent.parent


def load(d):
    """Counter values of the k_pcg_chip launches, keyed by (system index, counter): the systems of tools/pmc_chip_run.py run one
    after another, each through its own instantiation of the kernel (<8, 7, ...> for the 7-point system, <8, 5, ...> for the
    5-point one), so a system is told by the order in which the instantiations first appear."""
    out = collections.defaultdict(list)
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Dispatch_Id"]))
        order = []
        for r in rows:
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void dpcg::", "").replace("dpcg::", "")
            if name.startswith("k_pcg_chip"):
                if name not in order:
                    order.append(name)
                out[(order.index(name), r["Counter_Name"])].append(float(r["Counter_Value"]))
    return out


def mean(v):
    v = v[1:] if len(v) > 1 else v          # (the first launch of a system also pays its cold caches)
    if len(v) >= 4:                         # one sample per (launch, XCD) or per launch, whichever the profiler wrote: keep whole launches
        pass
    return sum(v) / len(v) if v else float("nan")


F, W, R = load(f"{out_dir}/pass1"), load(f"{out_dir}/pass2"), load(f"{out_dir}/pass3")
cases = [l.split() for l in open(f"{out_dir}/run.log") if l.startswith("poisson")]
lines = [f"# Memory-side traffic of the whole-chip solve ({tag})", "",
         "`rocprofv3 --pmc <counters> --kernel-trace --output-format csv -- python3 tools/pmc_chip_run.py`, three passes; per LAUNCH of",
         "`k_pcg_chip` (one whole solve) and per update.  Algorithmic bytes of an update = B_spmv + 76 n (the multi-launch update it replaces).", "",
         "| system | updates | algorithmic MB per update | read MB per launch (request counters) | read MB (2 x FETCH_SIZE) | write MB per launch | traffic MB per update | traffic / algorithmic |",
         "|---|---|---|---|---|---|---|---|"]
traffic = {}
for seg, c in enumerate(cases):
    name, its, n, nnz = c[0], int(c[1]), int(c[2]), int(c[3])
    alg = nnz * 12 + (n + 1) * 4 + 16 * n + 76 * n
    rd = 128 * mean(R[(seg, "TCC_EA0_RDREQ_128B_sum")]) + 64 * mean(R[(seg, "TCC_EA0_RDREQ_64B_sum")]) + 32 * mean(R[(seg, "TCC_EA0_RDREQ_32B_sum")])
    rd2 = 2 * 1024 * mean(F[(seg, "FETCH_SIZE")])
    wr = 1024 * mean(W[(seg, "WRITE_SIZE")])
    tot = rd + wr
    key = "chip_" + name.replace("poisson", "")
    traffic[key] = {"bytes_per_launch": round(tot), "updates": its, "bytes_per_update": round(tot / its), "algorithmic_per_update": alg,
                    "ratio": round(tot / its / alg, 3)}
    lines.append(f"| {name} | {its} | {alg / 1e6:.1f} | {rd / 1e6:.1f} | {rd2 / 1e6:.1f} | {wr / 1e6:.1f} | {tot / its / 1e6:.2f} | {tot / its / alg:.3f} |")
lines += ["", "Reading: what crosses the L2s' memory side per update is the written-through copies of the granules the neighbouring groups",
          "gather, the reduction slots and their polling, and whatever of the plainly stored granules the 4 MiB L2s evict -- not the matrix",
          "and not the vectors: they never leave the CUs."]
stats = glob.glob(f"{out_dir}/trace/**/*kernel_stats.csv", recursive=True)
if stats:
    shutil.copy(stats[0], ROOT / "profiles" / f"{tag}_chip_kernel_stats.csv")
    lines += ["", f"Kernel trace of the same workload: `profiles/{tag}_chip_kernel_stats.csv` (rocprofv3 --kernel-trace --stats)."]
(ROOT / "profiles" / f"{tag}_chip_traffic.md").write_text("\n".join(lines) + "\n")
pj = ROOT / "profiles" / "pmc_traffic.json"
allj = json.loads(pj.read_text()) if pj.exists() else {}
for k, v in traffic.items():
    allj[k] = v["bytes_per_launch"]
    allj[k + "_detail"] = v
pj.write_text(json.dumps(allj, indent=1) + "\n")
print("\n".join(lines))
