"""profiles/<tag>_mesh_spmv_traffic.md + the mesh entries of profiles/pmc_traffic.json from three rocprofv3 --pmc passes of
tools/pmc_mesh_run.py (FETCH_SIZE | WRITE_SIZE | TCC_EA0_RDREQ_sum + its 32B/64B/128B split).
    python tools/pmc_mesh_report.py <fetch_dir> <write_dir> <rdreq_dir> <run_log> <tag>
Read bytes: the size-split request counters (128 x RDREQ_128B + 64 x RDREQ_64B + 32 x RDREQ_32B); 2 x FETCH_SIZE x 1024 beside it (the
gfx950 correction of MI355X_MICROARCH.md's HBM section, calibrated in profiles/r01_pmc_summary.md).  Write bytes: WRITE_SIZE x 1024."""
import collections
import csv
import glob
import json
import pathlib
import sys

fetch_dir, write_dir, rdreq_dir, run_log, tag = sys.argv[1:6]
ROOT = pathlib.Path(__file__).resolve().parent.parent


def load(d):
    out = collections.defaultdict(list)
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Dispatch_Id"]))
        seg, last = -1, None
        for r in rows:
            name = r["Kernel_Name"].split("(")[0].replace("void dpcg::", "").replace("dpcg::", "")
            if name.startswith("k_gen_poisson") and r["Dispatch_Id"] != last:
                seg += 1
                last = r["Dispatch_Id"]
            base = name.split("<")[0]
            if base in ("k_spmv_stream", "k_spmv_tile"):
                out[(seg, r["Counter_Name"])].append(float(r["Counter_Value"]))
                out[(seg, "kernel")] = [name]
    return out


def mean(v):
    return sum(v) / len(v) if v else float("nan")


F, W, R = load(fetch_dir), load(write_dir), load(rdreq_dir)
cases = [l.split() for l in open(run_log) if l.split() and l.split()[0] in
         ("quadtree_foam", "quadtree_foam_rcm", "quadtree_random", "quadtree_random_gather", "delaunay", "quadtree_foam_as_is")]
lines = [f"# SpMV memory-side traffic on the config-3 mesh systems ({tag})", "",
         "`rocprofv3 --pmc <counters> --kernel-trace --output-format csv -- python3 tools/pmc_mesh_run.py`, three passes; per launch of the",
         "in-loop SpMV (+<p,Ap>) kernel.  Algorithmic bytes = nnz x 12 + (n + 1) x 4 + 16 n (SURVEY.md 8-d3).", "",
         "| system | n | nnz | kernel | reordered | algorithmic MB | read MB (request counters) | read MB (2 x FETCH_SIZE) | write MB | traffic / algorithmic |",
         "|---|---|---|---|---|---|---|---|---|---|"]
traffic = {}
for seg, c in enumerate(cases):
    name, n, nnz, kern, reordered = c[0], int(c[1]), int(c[2]), c[3], c[4]
    alg = nnz * 12 + (n + 1) * 4 + 16 * n
    rd = 128 * mean(R[(seg, "TCC_EA0_RDREQ_128B_sum")]) + 64 * mean(R[(seg, "TCC_EA0_RDREQ_64B_sum")]) + 32 * mean(R[(seg, "TCC_EA0_RDREQ_32B_sum")])
    rd2 = 2 * 1024 * mean(F[(seg, "FETCH_SIZE")])
    wr = 1024 * mean(W[(seg, "WRITE_SIZE")])
    tot = rd + wr
    traffic[f"spmv_mesh_{name}"] = {"bytes": round(tot), "algorithmic": alg, "ratio": round(tot / alg, 3), "kernel": kern,
                                    "reordered": reordered == "True"}
    lines.append(f"| {name} | {n} | {nnz} | {(F[(seg, 'kernel')] or ['?'])[0][:48]} | {reordered} | {alg / 1e6:.1f} | {rd / 1e6:.1f} | {rd2 / 1e6:.1f} | "
                 f"{wr / 1e6:.1f} | {tot / alg:.2f} |")
(ROOT / "profiles" / f"{tag}_mesh_spmv_traffic.md").write_text("\n".join(lines) + "\n")
pj = ROOT / "profiles" / "pmc_traffic.json"
allj = json.loads(pj.read_text()) if pj.exists() else {}
allj.update(traffic)
pj.write_text(json.dumps(allj, indent=1) + "\n")
print("\n".join(lines))
