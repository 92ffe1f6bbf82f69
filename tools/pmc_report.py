"""Turn the rocprofv3 --pmc CSVs into profiles/<tag>_pmc_summary.md and profiles/pmc_traffic.json
(bytes per launch of the SpMV kernel; bench.py reports it as roofline.traffic).

    python tools/pmc_report.py <fetch_dir> <write_dir> <rdreq_dir> <tag>

Passes (each its own rocprofv3 run of tools/pmc_run.py, --kernel-trace only beside --pmc):
  FETCH_SIZE | WRITE_SIZE | TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
Correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE is in KiB and on gfx950 tallies 128-B requests at
64 B, so read bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE x 1024 is exact.  The request-size counters verify
that on kernels of known volume (every read here is a 128-B request) for 4-, 8- and 16-byte-per-lane loads.
"""
import collections
import csv
import glob
import json
import pathlib
import sys

fetch_dir, write_dir, rdreq_dir, tag = sys.argv[1:5]
source = sys.argv[5] if len(sys.argv) > 5 else f"rocprofv3 --pmc passes of tools/pmc_run.py ({tag}); corrected as profiles/{tag}_pmc_summary.md states"
ROOT = pathlib.Path(__file__).resolve().parent.parent
SYSTEMS = [("3d_100", 1000000, 6940000), ("2d_1024", 1048576, 5238784), ("3d_256", 256 ** 3, 7 * 256 ** 3 - 6 * 256 ** 2),
           ("3d_256_f32_create", 256 ** 3, 7 * 256 ** 3 - 6 * 256 ** 2),
           ("scrambled3d_100_gather", 1000000, 6940000), ("scrambled3d_100_reordered", 1000000, 6940000)]
SPMV_ROWS = [0, 1, 2, 4, 5]


def load(d):
    """{(segment, kernel, counter): [values]}; a segment starts at each k_gen_poisson dispatch (one per system)."""
    out = collections.defaultdict(list)
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Dispatch_Id"]))
        seg, last_gen = -1, None
        for r in rows:
            name = r["Kernel_Name"].split("(")[0].replace("void dpcg::", "").replace("dpcg::", "")
            if name.startswith("k_gen_poisson") and r["Dispatch_Id"] != last_gen:
                seg += 1
                last_gen = r["Dispatch_Id"]
            base = name.split("<")[0]
            if base in ("k_spmv_stream", "k_spmv_tile"):
                base = "k_spmv"          # the gather kernel (cache-resident systems) or the x-tile kernel (HBM-resident)
            out[(seg, base, r["Counter_Name"])].append(float(r["Counter_Value"]))
    return out


def mean(vals, floor=0.0):
    vals = [v for v in vals if v > floor]  # drop the no-op launches after convergence
    return sum(vals) / len(vals) if vals else float("nan")


F, W, R = load(fetch_dir), load(write_dir), load(rdreq_dir)
MB = 1e6
lines = [f"# PMC traffic summary ({tag})", "",
         "`rocprofv3 --pmc <counters> --kernel-trace --output-format csv -- python3 tools/pmc_run.py`, three separate passes.",
         "", "## Calibration on kernels of known volume (256^3 system, far beyond the 256 MiB Infinity Cache)", "",
         "| kernel (bytes per lane) | known read MB | 2 x FETCH_SIZE x 1024 MB | 128 x RDREQ_128B + 64 x RDREQ_64B + 32 x RDREQ_32B MB | known write MB | WRITE_SIZE x 1024 MB |",
         "|---|---|---|---|---|---|"]
N, NNZ = SYSTEMS[2][1], SYSTEMS[2][2]
cal = [("k_dot_partials", 2, "<b,b>, 8 B/lane loads", 8 * N, 0),
       ("k_update_r", 2, "16 B/lane: reads q, r, dinv, writes r", 24 * N, 8 * N),
       ("k_update_xp_deferred", 2, "16 B/lane: even updates read r, dinv, p, write p; odd ones also read p', x, write x (mean)", 32 * N, 12 * N),
       ("k_convert", 3, "4 B/lane loads, 8 B/lane stores", 4 * NNZ, 8 * NNZ)]
for k, seg, what, rd, wr in cal:
    floor = 1000.0
    f2 = 2 * 1024 * mean(F[(seg, k, "FETCH_SIZE")], floor)
    rq = (128 * mean(R[(seg, k, "TCC_EA0_RDREQ_128B_sum")], floor) + 64 * mean(R[(seg, k, "TCC_EA0_RDREQ_64B_sum")] or [0], -1)
          + 32 * mean(R[(seg, k, "TCC_EA0_RDREQ_32B_sum")] or [0], -1))
    w = 1024 * mean(W[(seg, k, "WRITE_SIZE")], floor)
    lines.append(f"| {k} ({what}) | {rd / MB:.1f} | {f2 / MB:.1f} | {rq / MB:.1f} | {wr / MB:.1f} | {w / MB:.1f} |")
lines += ["", "## SpMV + <p,Ap> kernel of the PCG loop (k_spmv_tile: the plan picks the x-tile kernel for all three systems): bytes per launch", "",
          "| system | algorithmic MB (nnz*12 + (n+1)*4 + 16n) | read MB = 2 x FETCH_SIZE x 1024 | read MB from RDREQ sizes | write MB | traffic MB | traffic / algorithmic |",
          "|---|---|---|---|---|---|---|"]
traffic = {"_source": source}           # bench.py reports it as roofline.traffic_source
for seg in SPMV_ROWS:
    name, n, nnz = SYSTEMS[seg]
    alg = nnz * 12 + (n + 1) * 4 + 16 * n
    floor = 0.2 * alg / 2048  # FETCH_SIZE units of KiB/2: anything below is a no-op launch
    rd = 2 * 1024 * mean(F[(seg, "k_spmv", "FETCH_SIZE")], floor)
    rq = 128 * mean(R[(seg, "k_spmv", "TCC_EA0_RDREQ_128B_sum")], floor) + 64 * mean(R[(seg, "k_spmv", "TCC_EA0_RDREQ_64B_sum")], -1)
    wr = 1024 * mean(W[(seg, "k_spmv", "WRITE_SIZE")], 0.2 * 8 * n / 1024)
    tot = rd + wr
    traffic[f"spmv_{name}"] = round(tot)
    lines.append(f"| {'poisson' if name[0].isdigit() else ''}{name} | {alg / MB:.2f} | {rd / MB:.2f} | {rq / MB:.2f} | {wr / MB:.2f} | {tot / MB:.2f} | {tot / alg:.3f} |")
lines += ["", "scrambled3d_100_gather: the config-3 stand-in on the caller's numbering (k_spmv_stream, every x[col] its own 128-B line);",
          "scrambled3d_100_reordered: the same system after dpcg_reorder (k_spmv_tile on P A P^T).",
          "", "Reading: traffic is BELOW the algorithmic CSR bytes: the x-tile kernel reads a 2-byte local index instead of the",
          "4-byte column (-2 B per non-zero: -13.9 MB at 100^3, -234 MB at 256^3) and stages x in LDS; the gather kernel",
          "measured 107.2 MB at 100^3 (ratio 1.038) and 1906 MB at 256^3 (x planes at i +- n^2 re-fetched: reuse distance 6.8 MB per XCD > the 4 MiB L2).",
          "The 1M-DoF working set (~150 MB) sits in the Infinity Cache; these counters are the L2's memory-side requests",
          "(Infinity-Cache hits included), i.e. fabric traffic rather than DRAM traffic for those two rows.", ""]
(ROOT / "profiles").mkdir(exist_ok=True)
(ROOT / "profiles" / f"{tag}_pmc_summary.md").write_text("\n".join(lines))
(ROOT / "profiles" / "pmc_traffic.json").write_text(json.dumps(traffic, indent=1) + "\n")
print("\n".join(lines))
