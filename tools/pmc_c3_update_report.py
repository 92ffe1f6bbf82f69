"""profiles/<tag>_c3_update_counters.md from tools/pmc_c3_update.sh's passes: per kernel of the caller-order IC(0) update on the config-3
stand-in, bytes read at the memory side (128/64/32-byte request counters; 2 x FETCH_SIZE x 1024 beside them), bytes written
(WRITE_SIZE x 1024), L2 requests, and the duration inside the passes.
    python tools/pmc_c3_update_report.py gpurun_out/r04_pmc_c3 r04"""
import collections
import csv
import glob
import pathlib
import statistics
import sys

root, tag = sys.argv[1], sys.argv[2]
ROOT = pathlib.Path(__file__).resolve().parent.parent
KEEP = ("k_spmv_tile", "k_update_r", "k_sptrsv_syncfree_rec", "k_sptrsv_syncfree_stream", "k_lm_finish", "k_update_xp_deferred")
# SURVEY.md 8-d3 per triangular solve at C3: nnz(L) x 12 + (n + 1) x 4 + 16 n (factor as CSR, right-hand side in, solution out)
N, NNZ_A = 1_000_000, 6_940_000
NNZ_L = (NNZ_A - N) // 2 + N
ALG = {"k_sptrsv": NNZ_L * 12 + (N + 1) * 4 + 16 * N, "k_spmv_tile": NNZ_A * 12 + (N + 1) * 4 + 16 * N, "k_update_r": 32 * N,
       "k_update_xp_deferred": 40 * N, "k_lm_finish": 32 * N}


def short(name):
    return name.replace("void dpcg::", "").replace("dpcg::", "").split("(")[0]


def collect(form):
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for p in sorted(glob.glob(f"{root}/{form}/pass*")):
        d_by_id = {}
        for f in glob.glob(f"{p}/**/*kernel_trace.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                d_by_id[r["Dispatch_Id"]] = (short(r["Kernel_Name"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        med = collections.defaultdict(list)
        for k, us in d_by_id.values():
            med[k].append(us)
        med = {k: statistics.median(v) for k, v in med.items()}
        for k, us in d_by_id.values():
            if k.startswith(KEEP) and us >= 0.2 * med[k]:
                dur[k].append(us)
        for f in glob.glob(f"{p}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                if not k.startswith(KEEP):
                    continue
                d = d_by_id.get(r["Dispatch_Id"])
                if d and d[1] < 0.2 * med[k]:
                    continue                      # no-op launches after convergence
                vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return vals, dur


def mean(v):
    return sum(v) / len(v) if v else float("nan")


lines = [f"# Memory-side traffic of the caller-order IC(0) update on the config-3 stand-in ({tag})", "",
         "`bash tools/pmc_c3_update.sh` -- four `rocprofv3 --pmc` passes (FETCH_SIZE | WRITE_SIZE | TCC_EA0_RDREQ + its size split |",
         "TCP_TCC_READ_REQ, TCC_REQ / HIT / MISS) over `tools/trace_run_c3.py`, once with the sync-free solves on fixed-width records",
         "(the default at this row width) and once in CSR-stream form (`DPCG_SF_STREAM=1`).  Per launch; read bytes = 128 x RDREQ_128B +",
         "64 x RDREQ_64B + 32 x RDREQ_32B (2 x FETCH_SIZE x 1024 beside it); write bytes = WRITE_SIZE x 1024; algorithmic bytes per",
         "SURVEY.md 8-d3 (a triangular solve: nnz(L) x 12 + (n + 1) x 4 + 16 n = 71.6 MB).", ""]
for form in ("records", "stream"):
    vals, dur = collect(form)
    lines += [f"## sync-free solves in {form} form", "",
              "| kernel | us in the passes (median) | algorithmic MB | read MB (requests) | read MB (2 x FETCH) | write MB | moved / algorithmic | "
              "memory-side read requests | of them 32 B / 64 B / 128 B | L2 requests | L2 hit rate | TCP->TCC read requests |",
              "|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for k in sorted(vals):
        c = vals[k]
        rd = 128 * mean(c["TCC_EA0_RDREQ_128B_sum"]) + 64 * mean(c["TCC_EA0_RDREQ_64B_sum"]) + 32 * mean(c["TCC_EA0_RDREQ_32B_sum"])
        rd2 = 2 * 1024 * mean(c["FETCH_SIZE"])
        wr = 1024 * mean(c["WRITE_SIZE"])
        alg = next((v for p, v in ALG.items() if k.startswith(p)), float("nan"))
        req, hit = mean(c["TCC_REQ_sum"]), mean(c["TCC_HIT_sum"])
        lines.append(f"| {k[:56]} | {statistics.median(dur[k]):.1f} | {alg / 1e6:.1f} | {rd / 1e6:.1f} | {rd2 / 1e6:.1f} | {wr / 1e6:.1f} | "
                     f"{(rd + wr) / alg:.2f} | {mean(c['TCC_EA0_RDREQ_sum']):.3g} | {mean(c['TCC_EA0_RDREQ_32B_sum']):.3g} / "
                     f"{mean(c['TCC_EA0_RDREQ_64B_sum']):.3g} / {mean(c['TCC_EA0_RDREQ_128B_sum']):.3g} | {req:.3g} | {hit / req:.2f} | "
                     f"{mean(c['TCP_TCC_READ_REQ_sum']):.3g} |")
    lines.append("")
text = "\n".join(lines) + "\n"
target = ROOT / "profiles" / f"{tag}_c3_update_counters.md"
if target.exists() and "## Reading" in target.read_text():      # the hand-written reading of the tables stays
    text += target.read_text()[target.read_text().index("## Reading"):]
target.write_text(text)
print(text)
