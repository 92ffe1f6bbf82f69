"""bench.py -- PCG iterations/s and SpMV GB/s vs the HBM roofline on a 1M-DoF Poisson CSR system.

    python bench.py [--gpus N] [--steps K] [--warmup W]

A step = one pass of the hot path over one batch: every rank solves `--systems-per-gpu` (default 1)
1M-DoF pressure-Poisson systems (3-D 7-point, 100^3, Jacobi-preconditioned fp64 PCG, rtol 1e-8 on
<r,r>/<b,b>, max_iter 1024 -- the reference defaults, cg.py:51) that are already resident in HBM.
N > 1: one process per GPU (torch.distributed over RCCL).  Rank 0 owns the batch description and SCATTERS it over the backend
(3-integer specs of the synthetic systems -- rebuilt in the owner's HBM -- or, `--scatter arrays`, the CSR arrays and right-hand
sides as grouped point-to-point sends), system s goes to rank s mod N, every rank solves its share with no data-path collective,
the result records are GATHERED (all_gather).  The timed region (the solves) is bracketed by barrier + synchronize and the MAX
over ranks is taken; `scatter_ms` / `gather_ms` are reported beside `value`, never inside it.  `--config4`: BASELINE config 4 as
stated (64 x 256^3 over 8 GPUs, 8 per GPU).  Rank 0 prints ONE JSON line.  `value` = PCG iterations of all ranks / that time.
"""

from __future__ import annotations

import argparse
import json
import os
import pathlib
import subprocess
import sys
import time

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def spmv_bytes(n: int, nnz: int, wv: int = 8, wx: int = 8) -> int:
    """Algorithmic bytes of one CSR SpMV: nnz*(wv+4) + (n+1)*4 + 2*n*wx."""
    return nnz * (wv + 4) + (n + 1) * 4 + 2 * n * wx


def fused_update_bytes(n: int) -> int:
    """Extra algorithmic bytes of the two-kernel iteration's SpMV kernel, which also does the vector update of
    cg.py:79,83: reads z and x, writes x and the new p (the old p it reads is the SpMV's x operand): 4 x 8 B per row."""
    return 32 * n


def loop_kernel_bytes(system) -> int:
    """Algorithmic bytes of the kernel `spmv_dot_bench` times: the SpMV of the PCG loop, fused with the vector update
    when the system runs two-kernel updates."""
    return spmv_bytes(system.n, system.nnz) + (fused_update_bytes(system.n) if system.info()["two_kernel_updates"] else 0)


def sptrsv_bytes(n: int, nnz_l: int) -> int:
    """Algorithmic bytes of ONE sparse triangular solve: nnz_L*12 + (n+1)*4 + 2*n*8 + the row list 4*n."""
    return nnz_l * 12 + (n + 1) * 4 + 16 * n + 4 * n


def apply_roofline(system, apply_us: float) -> dict:
    """`roofline` object of a preconditioner apply z = L^-T (L^-1 r): two triangular solves."""
    b = 2 * sptrsv_bytes(system.n, system.info()["precond_nnz"])
    gbs = b / (apply_us * 1e-6) / 1e9
    return {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
            "algorithmic_bytes_per_apply": b, "us_per_apply": round(apply_us, 1),
            "levels": [system.info()["levels_lower"], system.info()["levels_upper"]]}


def in_loop_apply(entry: dict, jacobi_us_per_update: float) -> None:
    """The apply as the PCG loop pays for it: an update with this preconditioner minus an update with Jacobi (whose apply rides
    on K2 / K3 for nothing) -- launched from a replayed graph, next to `apply_roofline`'s host-launched stand-alone apply."""
    ar = entry.get("apply_roofline")
    if not ar or not entry.get("iterations"):
        return
    us = entry["ms"] * 1e3 / entry["iterations"] - jacobi_us_per_update
    if us <= 0:
        return
    gbs = ar["algorithmic_bytes_per_apply"] / (us * 1e-6) / 1e9
    entry["apply_in_loop"] = {"us_per_apply": round(us, 1), "achieved": round(gbs, 1), "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                              "method": "us per update minus a Jacobi update of the same system"}


def time_apply(system, b, torch, reps: int = 20) -> float:
    system.precond_apply(b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        system.precond_apply(b)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


L2_PEAK_GBS = 34500.0  # the eight 4 MiB L2s together, /opt/skills/guides/MI355X_MICROARCH.md "L2 (per XCD)" (36.9 TB/s with L1 reuse; sc1 gathers bypass the L1)


def l2_gather_ceiling(system, offsets) -> dict:
    """What the chip delivers for the access the whole-chip kernel's SpMV phase makes, with nothing else going on (dpcg_debug_l2_gather: the
    same 256 x 512 geometry and placement, a table of n / 8 sixteen-byte granules per XCD stored plainly by that XCD's workgroups, every
    thread gathering 8 x 7 granules per pass at the system's own column offsets with agent-scope loads): GB/s of gathered bytes with two rows'
    gathers in flight per lane (what the solve kernel's registers hold) and with four, and gathering the NEIGHBOURING XCD's part of a table
    that was written through (a STATIC table: after the first pass its lines sit in the reader's L2 as well -- the rate is the same; what a
    granule that another XCD rewrites every update costs is a memory-side round trip of ~2 us, a latency this probe does not see)."""
    import ctypes as C
    from deeppreconditioning_amd import _lib as L
    per_group = max(((system.n // 8) // 32) * 32, 32 * 512)
    offs = (C.c_int32 * 7)(*[int(o) for o in offsets])
    out = {"granules_per_xcd": per_group, "table_bytes_per_xcd": per_group * 16, "offsets_in_granules": [int(o) for o in offsets],
           "bytes_per_pass": 256 * 512 * 8 * 7 * 16}
    for key, depth, wt in (("plain_depth2", 2, 0), ("plain_depth4", 4, 0), ("other_xcd_written_through_depth2", 2, 2)):
        best = 0.0
        for _ in range(3):
            gbs, us, loc = C.c_double(), C.c_double(), C.c_int()
            L.check(L.lib().dpcg_debug_l2_gather(per_group, 40 if wt else 200, offs, depth, wt, None, C.byref(gbs), C.byref(us), C.byref(loc)))
            best = max(best, gbs.value)
        out[key] = round(best, 1)
        if not wt:
            out["groups_on_one_xcd"] = bool(loc.value)
    return out


def chip_roofline(system, b, check, k1: dict, traffic, counters=None, offsets=None) -> dict:
    """`roofline` object when the timed solve is the whole-chip kernel (dpcg_chip.hip: cg.py:58-90 in ONE launch, matrix and vectors resident
    in LDS / registers).  The level that serves this kernel's bytes is the L2s, not HBM: per update every matrix entry gathers one 16-byte
    granule {z, p} of its column (nnz x 16 B, agent-scope loads that hit the owner XCD's L2) and every row publishes one (n x 16 B).
    `achieved` = those bytes x updates over the kernel's duration between HIP events on its stream; `peak` = the eight L2s together
    (34.5 TB/s); `frac_of_measured_ceiling` = against the same gathers alone (`measured_l2_gather_gbs`, dpcg_debug_l2_gather), measured live.
    `phases`: the kernel against its two development variants -- the gathers issued but out of range (no memory request; what is left is
    instruction issue, DPCG_CHIP_BENCH=3) and no gathers at all (q = p; DPCG_CHIP_BENCH=1) -- all three between HIP events here and in
    the committed rocprofv3 kernel trace (profiles/r06_chip_kernel_stats.csv).  `traffic`: what crossed the L2s' memory side (PMC)."""
    n, nnz = system.n, system.nnz
    its = check.iterations
    os.environ["DPCG_CHIP_EVENTS"] = "1"
    times = {}
    try:
        for key, env in (("full", None), ("no_gathers", "1"), ("gathers_issued_out_of_range", "3")):
            if env is None:
                os.environ.pop("DPCG_CHIP_BENCH", None)
            else:
                os.environ["DPCG_CHIP_BENCH"] = env
            v = []
            for _ in range(12):
                r = system.solve(b, max_iter=(1024 if env is None else its), want_history=False)
                assert r.iterations == its
                v.append(system.chip_info()["kernel_ms"])
            times[key] = float(np.median(v[2:]))
    finally:
        os.environ.pop("DPCG_CHIP_EVENTS", None)
        os.environ.pop("DPCG_CHIP_BENCH", None)
    ms_full, ms_skip, ms_issue = times["full"], times["no_gathers"], times["gathers_issued_out_of_range"]
    gathered, published = 16 * nnz, 16 * n
    b_l2 = gathered + published
    us_upd = ms_full * 1e3 / its
    achieved = b_l2 / (us_upd * 1e-6) / 1e9
    us_spmv = max((ms_full - ms_skip) * 1e3 / its, 1e-3)
    us_bytes = max((ms_full - ms_issue) * 1e3 / its, 1e-3)
    # the system's own column offsets, in granules: what the ceiling probe gathers at
    ci = system.chip_info()
    offs = offsets if offsets is not None else [0] * 7
    ceil = l2_gather_ceiling(system, offs)
    ceil_best = max(ceil["plain_depth2"], ceil["plain_depth4"])
    b_hbm = spmv_bytes(n, nnz) + int(9.5 * 8 * n)
    out = {"bound": "l2", "kernel": f"k_pcg_chip (the whole solve, cg.py:58-90, in one launch of {ci['workgroups']} workgroups x {ci['threads']} threads; "
                                    f"{ci['rows_per_workgroup']} rows per workgroup, matrix and vectors resident in LDS / registers, operands of q = A p "
                                    f"gathered as 16-byte granules out of the XCDs' L2s)",
           "achieved": round(achieved, 1), "peak": L2_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / L2_PEAK_GBS, 4),
           "frac_of_measured_ceiling": round(achieved / ceil_best, 4), "measured_l2_gather_gbs": ceil,
           "traffic": traffic, "traffic_level": "memory side of the L2s (PMC, profiles/r06_chip_traffic.md); L2-level request / hit counters: `l2_counters`",
           "l2_counters": counters,
           "l2_bytes_per_update": b_l2, "l2_bytes_per_launch": its * b_l2, "gathered_bytes_per_update": gathered, "published_bytes_per_update": published,
           "us_per_launch": round(ms_full * 1e3, 2), "updates_per_launch": its, "us_per_update": round(us_upd, 3),
           "timing": "HIP events on the launch stream around the kernel (DPCG_CHIP_EVENTS), median of 10 launches",
           "phases": {"us_per_update": {"whole": round(us_upd, 3), "without_gathers": round(ms_skip * 1e3 / its, 3),
                                        "gathers_issued_out_of_range": round(ms_issue * 1e3 / its, 3)},
                      "spmv_phase": {"us_per_update": round(us_spmv, 3), "achieved": round(gathered / (us_spmv * 1e-6) / 1e9, 1), "unit": "GB/s",
                                     "frac": round(gathered / (us_spmv * 1e-6) / 1e9 / L2_PEAK_GBS, 4),
                                     "frac_of_measured_ceiling": round(gathered / (us_spmv * 1e-6) / 1e9 / ceil_best, 4),
                                     "method": "kernel minus its variant without the gathers (q = p), per update: issue + bytes of q = A p"},
                      "gathered_bytes_alone": {"us_per_update": round(us_bytes, 3), "achieved": round(gathered / (us_bytes * 1e-6) / 1e9, 1), "unit": "GB/s",
                                               "frac": round(gathered / (us_bytes * 1e-6) / 1e9 / L2_PEAK_GBS, 4),
                                               "method": "kernel minus its variant whose gathers are issued out of range (no memory request)"},
                      "exchanges_and_updates_us": round(ms_skip * 1e3 / its, 3),
                      "reading": "the update is latency-bound outside the gathers: two chip-wide exchanges (reduction + barrier, two hops each) and the "
                                 "register updates move almost no bytes"},
           "regime": "on_chip_resident: matrix, x, r, p, q, dinv live in LDS / registers for the whole solve; the only operands that move are the "
                     "granules, XCD-locally through the L2s.  The HBM roofline applies to the kernels that stream from memory: `streaming_spmv_kernel` "
                     "(systems beyond 1,048,576 rows, other preconditioners) and `hbm_bound_256cubed`",
           "hbm_algorithmic_equivalent": {"bytes_per_update": b_hbm, "gbs": round(b_hbm / (us_upd * 1e-6) / 1e9, 1),
                                          "note": "SURVEY.md 8-d3's bytes of the multi-launch update this kernel replaces (B_spmv + 76 n) over its time: "
                                                  "they are served on chip, so this is NOT a fraction of anything -- kept for comparison across rounds"},
           "streaming_spmv_kernel": {"bound": "hbm", **{k: k1[k] for k in ("kernel", "achieved", "peak", "frac", "traffic", "algorithmic_bytes_per_launch",
                                                                            "us_per_launch", "traffic_source", "frac_of_measured_ceiling") if k in k1}},
           "measured_stream_gbs": k1.get("measured_stream_gbs")}
    return out


def stencil_offsets(dim: int, grid: int) -> list:
    """Column offsets (col - row) of an interior row of the synthetic Poisson system, padded to 7: the pattern the L2 gather probe reads at."""
    o = [-grid * grid, -grid, -1, 0, 1, grid, grid * grid] if dim == 3 else [-grid, -1, 0, 1, grid, 0, 0]
    return o


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 20; 2 with --config4)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed warm-up steps (default 3; 1 with --config4)")
    ap.add_argument("--systems-per-gpu", type=int, default=None)
    ap.add_argument("--config4", action="store_true",
                    help="BASELINE config 4 as stated: 64 x poisson3d(256) sharded 8 per GPU over 8 GPUs (8 x --gpus systems, "
                         "distinct b per system id); same flags otherwise")
    ap.add_argument("--scatter", default="specs", choices=["specs", "arrays"],
                    help="N > 1: what rank 0 scatters -- 3-integer specs (systems rebuilt in the owner's HBM) or the CSR arrays + b "
                         "of every system as grouped point-to-point sends (the real-matrix path)")
    ap.add_argument("--dim", type=int, default=3)
    # (`--grid`, not `--n`: torch.distributed.run reads `--n` in front of the script's own arguments as an ambiguous
    # abbreviation of its --nnodes / --nproc-per-node and refuses the command line)
    ap.add_argument("--grid", "--n", dest="n", type=int, default=100, help="grid points per side (3-D 100 -> 1,000,000 DoF)")
    ap.add_argument("--precond", default="jacobi", choices=["jacobi", "none", "ic0"])
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary workloads")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed and run the scatter / gather even with ONE rank (under torch.distributed.run): "
                         "exercises the RCCL path of this script on a one-GPU box")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for the barrier / result reduction (nccl = RCCL; gloo lets the "
                         "N > 1 control flow be exercised with several ranks sharing one GPU)")
    args = ap.parse_args()
    if args.config4:
        args.dim, args.n = 3, 256
        args.systems_per_gpu = 8 if args.systems_per_gpu is None else args.systems_per_gpu
    args.systems_per_gpu = 1 if args.systems_per_gpu is None else args.systems_per_gpu
    args.steps = (2 if args.config4 else 20) if args.steps is None else args.steps
    args.warmup = (1 if args.config4 else 3) if args.warmup is None else args.warmup
    return args


def cpu_baseline(dim: int, n: int) -> dict:
    """The CPU oracle (C restatement of cg.py, oracle/pcg_oracle.c, OpenMP) timed on this box's host
    cores on the same system -- a reported baseline, not the target."""
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")   # idle OpenMP workers must not spin beside the GPU driver thread
    from oracle import c_oracle as CO
    from oracle import oracle as O
    A = O.poisson3d(n) if dim == 3 else O.poisson2d(n)
    b = O.rhs(A.shape[0], 0)
    dinv = O.jacobi_dinv(A)
    CO.pcg(A, b, "jacobi", dinv=dinv, max_iter=3)  # warm caches / thread pool
    # thread count: more is not better on a shared many-socket host (measured: 128 threads slower than 1), so a
    # short sweep picks it and the full solve runs at the best count; the 1-thread figure is reported beside it
    # 
    max_threads = CO.num_threads()
    sweep = {}
    t = 1
    while t <= max_threads:
        CO.set_num_threads(t)
        sec_t, it_t, _, _ = CO.pcg(A, b, "jacobi", dinv=dinv, max_iter=20)
        sweep[t] = round(it_t / sec_t, 2)
        t *= 2
    best = max(sweep, key=sweep.get)
    CO.set_num_threads(best)
    sec, iters, _, _ = CO.pcg(A, b, "jacobi", dinv=dinv)
    CO.set_num_threads(max_threads)
    out = {"value": round(iters / sec, 2), "unit": "iterations/s", "cores": best, "kind": "port",
           "sample": f"1 full solve of the same system ({iters} PCG iterations, {sec:.2f} s) by oracle/pcg_oracle.c "
                     f"with {best} OpenMP threads (best of a 20-iteration sweep) of {os.cpu_count()} host CPUs",
           "iterations": iters, "iterations_per_s_by_threads": sweep}
    try:    # the reference's other CPU path: scipy.sparse.linalg.cg with benchmark_cg's settings (utils.py:66-72)
        import scipy.sparse as sp
        import scipy.sparse.linalg as spla
        count = [0]
        t0 = time.perf_counter()
        spla.cg(A, b, maxiter=512, M=sp.diags(dinv), callback=lambda _: count.__setitem__(0, count[0] + 1))
        dt = time.perf_counter() - t0
        out["scipy_cg"] = {"value": round(count[0] / dt, 2), "iterations": count[0], "seconds": round(dt, 2),
                           "settings": "scipy.sparse.linalg.cg, rtol 1e-5, maxiter 512, Jacobi M (benchmark_cg)"}
    except Exception as e:  # noqa: BLE001 - a missing scipy must not cost the bench line
        out["scipy_cg"] = {"error": repr(e)}
    return out


def measured_stream_ceilings() -> dict:
    """HBM ceiling as this box delivers it, measured with the library's own streaming kernel
    (`dpcg_stream_bench`: 16-byte lane accesses, XCD-contiguous slabs, the SpMV's grid; shapes chosen with
    tools/stream_lab) on buffers far beyond the 256 MiB Infinity Cache: copy (1 read : 1 write), triad (2 : 1), the
    read:write ratio of a 7-point CSR SpMV (11 : 1) and read-only, each with plain and with non-temporal accesses.
    GB/s of reads + writes.  Any share of writes costs this memory system a fifth of its read-only rate."""
    from deeppreconditioning_amd.operators import stream_bench
    mib = 1 << 20
    out = {}
    for tag, nt in (("", False), ("_nt", True)):
        out["copy_1r1w" + tag] = round(stream_bench(1, True, 1024 * mib, 10, nt), 1)
        out["triad_2r1w" + tag] = round(stream_bench(2, True, 512 * mib, 10, nt), 1)
        out["spmv_like_11r1w" + tag] = round(stream_bench(11, True, 128 * mib, 10, nt), 1)
        out["read_only" + tag] = round(stream_bench(4, False, 384 * mib, 10, nt), 1)
        # the same streams WALKED TOGETHER by the whole grid instead of one slab per workgroup; the copy then is the shape
        # MI355X_MICROARCH.md quotes 6.29 TB/s for (one 16-byte element per thread)
        out["copy_float4" + tag] = round(stream_bench(1, True, 1024 * mib, 10, nt, walk=True), 1)
        out["spmv_like_11r1w_walk" + tag] = round(stream_bench(11, True, 128 * mib, 10, nt, walk=True), 1)
    # the same kernel on footprints INSIDE the Infinity Cache (the headline system's regime: ~100 MB per SpMV, 24-48 MB per
    # vector kernel): what the fabric delivers to a plain stream of that size, launched back to back
    out["cache_resident_spmv_like_11r1w_96MB"] = round(stream_bench(11, True, 8 * mib, 40, False), 1)
    out["cache_resident_triad_2r1w_24MB"] = round(stream_bench(2, True, 8 * mib, 40, False), 1)
    out["method"] = ("dpcg_stream_bench, 10 launches between HIP events, 1.4-2 GiB per launch; cache_resident_*: 40 launches, "
                     "8 MiB written per launch")
    return out


def main() -> None:
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world == 1:
        # convenience: launch the one-process-per-GPU job as a child (never exec after touching the GPU)
        port = os.environ.get("MASTER_PORT", "29531")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", port, str(ROOT / "bench.py")] + \
              ["--grid" if a == "--n" else a for a in sys.argv[1:]]
        sys.exit(subprocess.run(cmd).returncode)

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank % torch.cuda.device_count())
    dist = None
    if world > 1 or (args.force_dist and "RANK" in os.environ):
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", torch.cuda.current_device()))
        else:
            dist.init_process_group("gloo")
    red_dev = "cuda" if args.backend == "nccl" else "cpu"

    import deeppreconditioning_amd as D
    from deeppreconditioning_amd import poisson

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- the batch: rank 0 owns its description; scatter over the backend (RCCL when nccl); systems resident in HBM before
    # ---- the timed region.  System s of the batch has right-hand side default_rng(s) and goes to rank s mod world.
    from deeppreconditioning_amd import batch as B
    make_pc = lambda: {"jacobi": D.Jacobi(), "none": None, "ic0": D.IC0("solve")}[args.precond]   # noqa: E731
    comm = {"backend": None, "ranks": 1, "scatter_mode": None, "scatter_ms": None, "gather_ms": None}
    count = world * args.systems_per_gpu
    my_ids = list(range(count))
    received = None
    if dist is not None:
        dist.barrier()                       # the first collective creates the communicator: not part of the scatter's time
        barrier()
        t_sc = time.perf_counter()
        if args.scatter == "specs":
            specs = [B.SystemSpec(args.dim, args.n, i) for i in range(count)] if rank == 0 else None
            my_specs, my_ids, count = B.scatter_specs(specs)
            assert all((sp.dim, sp.n) == (args.dim, args.n) for sp in my_specs)
            my_seeds = [sp.seed for sp in my_specs]
        else:
            systems0 = None
            if rank == 0:                    # rank 0 holds every system of the batch (here: generated in ITS HBM)
                rp0, ci0, v0 = poisson.poisson_csr(args.dim, args.n)
                systems0 = [(rp0, ci0, v0, poisson.rhs(rp0.numel() - 1, seed=i)) for i in range(count)]
            received, my_ids, _sizes = B.scatter_systems(systems0)
            del systems0
            my_seeds = list(my_ids)
        torch.cuda.synchronize()
        comm.update(backend=dist.get_backend() + (" (RCCL over xGMI)" if args.backend == "nccl" else ""),
                    ranks=dist.get_world_size(), scatter_mode=args.scatter,
                    scatter_ms=round((time.perf_counter() - t_sc) * 1e3, 3))
    else:
        my_seeds = list(my_ids)
    if received is not None:
        # the real-matrix path: every system arrives with its own arrays (in host memory with gloo); one handle per system
        dev = torch.device("cuda", torch.cuda.current_device())
        received = [tuple(t.to(dev) for t in parts) for parts in received]
        handles = [D.CsrSystem(rp_, ci_, v_, rp_.numel() - 1) for rp_, ci_, v_, _ in received]
        rhs = [parts[3] for parts in received]
    else:
        # synthetic systems are rebuilt in the owner's HBM from their specs; systems of one rank share the ONE matrix (the
        # handles borrow the same CSR arrays, own work vectors) and differ in their right-hand sides
        rp_, ci_, v_ = poisson.poisson_csr(args.dim, args.n)
        handles = [D.CsrSystem(rp_, ci_, v_, rp_.numel() - 1) for _ in range(min(4, len(my_seeds)))]
        rhs = [poisson.rhs(handles[0].n, seed=i) for i in my_seeds]
    for h in handles:
        h.set_preconditioner(make_pc())
    system = handles[0]
    from deeppreconditioning_amd.batch import solve_batch
    last_records = np.zeros((len(rhs), 4))

    def step() -> int:
        if len(rhs) == 1:
            results = [system.solve(rhs[0], want_history=False)]
        else:
            # up to 4 systems of a rank in flight at once on separate streams: another system's kernels fill this one's
            # launch boundaries
            results = []
            for lo in range(0, len(rhs), 4):
                chunk = rhs[lo:lo + 4]
                hs = handles[lo:lo + len(chunk)] if len(handles) == len(rhs) else handles[:len(chunk)]
                results += solve_batch(hs, chunk, n_streams=len(chunk))
        for i, r in enumerate(results):
            last_records[i] = (r.iterations, r.status, r.final_res, r.seconds)
        return sum(r.iterations for r in results)

    # Device wake-up, part of the setup (not a step, not timed): while torch was imported and the systems were built the GPU sat
    # idle, and ~10 ms after load resumes its power management stalls everything once for 50-90 ms (tools/idle_probe.py: one
    # 4.5 ms solve in six took 62 ms after 2 s of host-only work).  0.3 s of SpMV launches put that transition before the
    # warm-up steps instead of somewhere inside the timed ones.
    t_wake = time.perf_counter() + 0.3
    while time.perf_counter() < t_wake:
        system.spmv_dot_bench(repeats=50)
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    iters_local = 0
    for _ in range(args.steps):
        iters_local += step()
    barrier()
    elapsed = time.perf_counter() - t0

    tot = torch.tensor([float(iters_local)], device=red_dev, dtype=torch.float64)
    tmax = torch.tensor([elapsed], device=red_dev, dtype=torch.float64)
    per_rank = None
    if dist is not None:
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        # ---- gather (outside `value`): the result records of the last step's systems, all_gather over the backend
        barrier()
        t_g = time.perf_counter()
        table = B.gather_records(last_records, count)
        torch.cuda.synchronize()
        comm["gather_ms"] = round((time.perf_counter() - t_g) * 1e3, 3)
        comm["gathered_records"] = {"systems": int(table.shape[0]), "iterations": [int(v) for v in table[:, 0]],
                                    "status": [int(v) for v in table[:, 1]],
                                    "max_final_res": float(table[:, 2].max()) if table.size else None}
        # one record per rank (its own clock and its own SpMV roofline), gathered the same way
        ms_r = system.spmv_dot_bench(repeats=100)
        gbs_r = loop_kernel_bytes(system) / (ms_r * 1e-3) / 1e9
        mine = np.array([[iters_local / elapsed, elapsed, ms_r * 1e3, gbs_r / HBM_PEAK_GBS]])
        ranks_table = B.gather_records(mine, world)
        per_rank = [{"rank": r, "iterations_per_s": round(float(ranks_table[r, 0]), 1), "seconds": round(float(ranks_table[r, 1]), 4),
                     "spmv_us_per_launch": round(float(ranks_table[r, 2]), 3), "roofline_frac": round(float(ranks_table[r, 3]), 4)}
                    for r in range(world)]
    total_iters, t = float(tot.item()), float(tmax.item())

    if rank == 0:
        n, nnz = system.n, system.nnz
        check = system.solve(poisson.rhs(n, 0))
        # ---- roofline of the dominant kernel: the SpMV+<p,Ap> launch inside the PCG iteration ----
        ms = system.spmv_dot_bench(repeats=200)     # HIP events on the launch stream, 200 launches
        info = system.info()
        b_alg = loop_kernel_bytes(system)
        achieved = b_alg / (ms * 1e-3) / 1e9
        traffic = None
        pmc = ROOT / "profiles" / "pmc_traffic.json"
        pmc_all = json.loads(pmc.read_text()) if pmc.exists() else {}
        traffic = pmc_all.get(f"spmv_{args.dim}d_{args.n}")
        ceilings = measured_stream_ceilings()
        line = {
            "metric": "pcg_iterations_per_sec", "value": round(total_iters / t, 1), "unit": "iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * t / args.steps, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "backend": comm["backend"], "ranks": comm["ranks"], "scatter_mode": comm["scatter_mode"],
            "scatter_ms": comm["scatter_ms"], "gather_ms": comm["gather_ms"],
            "config": {"workload": f"poisson{args.dim}d_{args.n}_{args.precond}_pcg_fp64", "dof": n, "nnz": nnz,
                       "rtol_sq": 1e-8, "max_iter": 1024, "iterations_per_solve": check.iterations,
                       "final_res": check.final_res, "systems_per_gpu_per_step": args.systems_per_gpu,
                       "batch": ("BASELINE config 4: 64 x poisson3d(256), 8 per GPU on 8 GPUs" if args.config4 else None),
                       "systems_in_batch": count,
                       "parallelism": f"independent systems sharded s mod {world}, no data-path collective"
                                      + (f"; {args.systems_per_gpu} systems per GPU, {min(4, len(rhs))} interleaved on streams"
                                         if args.systems_per_gpu > 1 else "")},
            "roofline": {"bound": "hbm", "kernel": (f"k_spmv_{info['spmv_kernel']}<FUSE> (p = z + beta p, x += alpha p, q = A p, <p,q>)"
                                    if info["two_kernel_updates"] else
                                    f"k_spmv_{info['spmv_kernel']}<CTL,DOT> (SpMV + <p,Ap>)"),
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "algorithmic_bytes_per_launch": b_alg, "us_per_launch": round(ms * 1e3, 3),
                         # PMC counters need rocprofv3, so `traffic` comes from a tracked file: say which run made it
                         "traffic_source": pmc_all.get("_source", "profiles/pmc_traffic.json (rocprofv3 --pmc passes of "
                                                                  "tools/pmc_run.py, corrected as profiles/*_pmc_summary.md states)"),
                         "measured_stream_gbs": ceilings,
                         # against a plain stream of the same regime: inside the Infinity Cache the 11:1 stream of the
                         # SpMV's own size, beyond it the best HBM-bound stream
                         "frac_of_measured_ceiling": round(achieved / (ceilings["cache_resident_spmv_like_11r1w_96MB"]
                                                                      if b_alg < 200e6 else
                                                                      max(v for k, v in ceilings.items() if isinstance(v, float)
                                                                          and not k.startswith("cache_resident"))), 4)},
        }
        chip = system.chip_info()
        if chip["chip_by_default"] and args.precond in ("jacobi", "none"):
            line["roofline"] = chip_roofline(system, poisson.rhs(n, 0), check, line["roofline"], pmc_all.get(f"chip_{args.dim}d_{args.n}"),
                                             pmc_all.get(f"chip_{args.dim}d_{args.n}_l2_counters"), stencil_offsets(args.dim, args.n))
            if world == 1 and args.systems_per_gpu == 1:
                # what a solve costs beyond its kernel -- the Python call, the slot pre-fill, the memset, the gather of b and the syncs around the
                # launch (all inside `value`'s wall clock, none of it inside the kernel's duration)
                line["roofline"]["per_solve_time_outside_the_kernel_us"] = round(line["ms_per_step"] * 1e3 - line["roofline"]["us_per_launch"], 1)
        if per_rank is not None:
            line["per_rank"] = per_rank
            line["gathered_records"] = comm["gathered_records"]
            line["roofline"]["per_rank_frac"] = [r["roofline_frac"] for r in per_rank]
        if args.precond == "jacobi" and not info["two_kernel_updates"] and not chip["chip_by_default"]:
            # the whole PCG update of the timed solve against the same peak: K1 + K2 (q, r, dinv read; r written) + K3 (r, dinv, p read;
            # p written; every other update also p', x read and x written) = B_spmv + 9.5 n x 8 bytes, over wall time per update
            b_upd = b_alg + int(9.5 * 8 * n)
            us_upd = t / max(total_iters, 1.0) * world * 1e6
            line["roofline"]["whole_update"] = {"algorithmic_bytes": b_upd, "us": round(us_upd, 2),
                                                "achieved": round(b_upd / (us_upd * 1e-6) / 1e9, 1),
                                                "frac": round(b_upd / (us_upd * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)}
        if "regime" not in line["roofline"]:
            line["roofline"]["regime"] = ("cache_resident_1M: the ~150 MB working set of the headline system lives in the 256 MiB "
                                          "Infinity Cache, so `achieved` is fabric, not DRAM, bandwidth; the HBM-bound figure is "
                                          "`hbm_bound_256cubed` below")
        if world == 1:
            # the same kernel on a system far beyond the Infinity Cache (BASELINE config 4's 256^3: 1.74 GB per SpMV)
            # (the time of this kernel moves 274-303 us with where the system's arrays happen to land in HBM -- tools/c4_variance_probe.py,
            # tools/c4_offset_probe.py: physical placement, nothing a virtual offset changes -- so the system is created three times
            # on fresh allocations and the MEDIAN placement is reported, all three samples beside it)
            from deeppreconditioning_amd.operators import release_cached_memory
            samples = []
            for _ in range(3):
                s4 = poisson.poisson_system(3, 256)
                s4.set_preconditioner(D.Jacobi())
                samples.append(s4.spmv_dot_bench(repeats=40))
                s4.close()
                del s4
                release_cached_memory()
                torch.cuda.empty_cache()
            s4 = poisson.poisson_system(3, 256)
            s4.set_preconditioner(D.Jacobi())
            samples.append(s4.spmv_dot_bench(repeats=40))
            samples_sorted = sorted(samples)
            ms4 = 0.5 * (samples_sorted[1] + samples_sorted[2])          # median of the four placements
            b4_alg = loop_kernel_bytes(s4)
            line["roofline"]["hbm_bound_256cubed"] = {
                "kernel": f"k_spmv_{s4.info()['spmv_kernel']}<CTL,DOT>", "achieved": round(b4_alg / (ms4 * 1e-3) / 1e9, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(b4_alg / (ms4 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "traffic": pmc_all.get("spmv_3d_256"), "algorithmic_bytes_per_launch": b4_alg,
                "us_per_launch": round(ms4 * 1e3, 2), "dof": s4.n, "nnz": s4.nnz,
                "frac_of_measured_spmv_like_stream": round(b4_alg / (ms4 * 1e-3) / 1e9 /
                                                           max(ceilings["spmv_like_11r1w"], ceilings["spmv_like_11r1w_nt"],
                                                               ceilings["spmv_like_11r1w_walk"], ceilings["spmv_like_11r1w_walk_nt"]), 4),
                "non_temporal_streams": bool(s4.info().get("spmv_nt", False)),
                "us_per_launch_by_placement": [round(v * 1e3, 2) for v in samples],
                "placement": "median of four creations of the system on fresh allocations"}
            # the same kernel INSIDE the PCG loop (K2 / K3 stream 1.5 GB through the Infinity Cache between two SpMVs, so p is not
            # resident when K1 starts): wall time per update of a 48-update solve, measured live; the in-loop K1 time itself needs the
            # profiler -- the median of the committed kernel trace of tools/trace_run_c4.py is quoted beside it
            b4 = poisson.rhs(s4.n, 0)
            s4.solve(b4, max_iter=8, want_history=False)
            r4 = s4.solve(b4, max_iter=48, want_history=False)
            in_loop = {"pcg_us_per_update": round(r4.seconds / max(r4.iterations, 1) * 1e6, 1), "updates": r4.iterations}
            tr = next((t for t in (ROOT / "profiles" / f"r0{k}_kernel_trace_256cubed_summary.txt" for k in (5, 4)) if t.exists()), None)
            if tr is not None:
                for ln in tr.read_text().splitlines():
                    if ln.startswith("k_spmv_tile") and "median" in ln:
                        in_loop["k1_us_median_in_loop_rocprof"] = float(ln.split("median")[1].split("us")[0])
                        in_loop["k1_frac_in_loop_rocprof"] = round(b4_alg / (in_loop["k1_us_median_in_loop_rocprof"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
                        in_loop["source"] = f"profiles/{tr.name} (rocprofv3 --kernel-trace of tools/trace_run_c4.py)"
                        break
            line["roofline"]["hbm_bound_256cubed"]["in_loop"] = in_loop
            s4.close()
            del s4
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.dim, args.n)
        if world == 1 and not args.no_extra:
            line["extra"] = extra_workloads(D, poisson, torch)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def mesh_workloads(D, poisson, torch, solve_twice, pmc_all) -> dict:
    """BASELINE config 3 on GENUINELY unstructured matrices (deeppreconditioning_amd/meshes.py): a quadtree-refined finite-volume
    Laplacian with hanging nodes and holes -- in OpenFOAM's own numbering (hexRef8: children appended) and in a random one -- and the
    Laplacian of a Delaunay triangulation of 1M random points.  Per system through the PLAIN call: kernel chosen, gather ratio of the
    caller's numbering, SpMV roofline and PMC traffic, Jacobi / IC(0) in multicolour order to the solution (reference defaults: they
    stop at max_iter = 1024 on these 2-D meshes unless they converge), IC(0) in the caller's order per update."""
    from deeppreconditioning_amd import meshes
    out = {}
    for name, make in (("quadtree_foam", lambda: meshes.quadtree_fv_laplacian(1000, 0)),
                       ("quadtree_random", lambda: meshes.quadtree_fv_laplacian(1000, 0, numbering="random")),
                       ("delaunay", lambda: meshes.delaunay_laplacian(1000000, 0))):
        A = make()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        S = D.CsrSystem.from_any(A)
        torch.cuda.synchronize()
        create_ms = (time.perf_counter() - t0) * 1e3
        info = S.info()
        b = poisson.rhs(S.n, 0)
        deg = np.diff(A.indptr)
        S.set_preconditioner(D.Jacobi())
        ms = S.spmv_dot_bench(200)
        b_alg = loop_kernel_bytes(S)
        gbs = b_alg / (ms * 1e-3) / 1e9
        tr = pmc_all.get(f"spmv_mesh_{name}")
        e = {"dof": S.n, "nnz": S.nnz, "row_lengths": [int(deg.min()), int(deg.max())], "create_incl_upload_and_reordering_ms": round(create_ms, 1),
             "reordered": info["reordered"], "gather_ratio_before": round(info["gather_ratio"], 2), "spmv_kernel": info["spmv_kernel"],
             "spmv_roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                               "algorithmic_bytes_per_launch": b_alg, "us_per_launch": round(ms * 1e3, 2),
                               "traffic": tr["bytes"] if tr else None, "traffic_over_algorithmic": tr["ratio"] if tr else None,
                               "traffic_source": "profiles/r05_mesh_spmv_traffic.md (rocprofv3 --pmc passes of tools/pmc_mesh_run.py)"}}
        r = solve_twice(S, b)
        e["jacobi"] = {"iterations": r.iterations, "status": r.status, "final_res": r.final_res, "ms": round(r.seconds * 1e3, 3),
                       "us_per_update": round(r.seconds / max(r.iterations, 1) * 1e6, 1), "iterations_per_s": round(r.iterations / r.seconds, 1)}
        # config 5 on the mesh: fp32-stored SpMV operands (4-byte value slots: rows of 9 entries resident at 1M rows when the numbering has a band)
        rm = solve_twice(S, b, flags=D._lib.SPMV_F32)
        e["jacobi_mixed_precision"] = {"iterations": rm.iterations, "us_per_update": round(rm.seconds / max(rm.iterations, 1) * 1e6, 1),
                                       "iterations_per_s": round(rm.iterations / rm.seconds, 1), "final_res_recurrence": rm.final_res}
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        S.set_preconditioner(D.IC0("solve", ordering="multicolor"))
        torch.cuda.synchronize()
        setup_ms = (time.perf_counter() - t0) * 1e3
        r = solve_twice(S, b)
        e["ic0_multicolor_solve"] = {"colors": S.precond_ordering()[0], "levels": S.info()["levels_lower"], "setup_new_pattern_ms": round(setup_ms, 2),
                                     "iterations": r.iterations, "status": r.status, "final_res": r.final_res, "ms": round(r.seconds * 1e3, 3),
                                     "us_per_update": round(r.seconds / max(r.iterations, 1) * 1e6, 1)}
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        S.set_preconditioner(D.IC0("solve"))
        torch.cuda.synchronize()
        setup_ms = (time.perf_counter() - t0) * 1e3
        S.solve(b, max_iter=20, want_history=False)
        r = S.solve(b, max_iter=40, want_history=False)      # (per update only: 2 000 levels in OpenFOAM's numbering cost milliseconds each)
        e["ic0_solve"] = {"levels": S.info()["levels_lower"], "setup_ms": round(setup_ms, 2), "updates_timed": r.iterations,
                          "us_per_update": round(r.seconds / max(r.iterations, 1) * 1e6, 1)}
        out[f"c3_mesh_{name}"] = e
        S.close()
        del S, A
    return out


def extra_workloads(D, poisson, torch) -> dict:
    """Secondary numbers (not the headline), one entry per BASELINE.json config that fits one GPU."""
    out = {}
    # the GPU has been idle for the ~20 s of the CPU baseline: the first kernels after that run at ramping clocks (one run
    # of this file timed the 4.4 ms config-2 solve below at 79 ms).  ~0.2 s of solves bring it back before anything is timed.
    warm = poisson.poisson_system(3, 64)
    warm.set_preconditioner(D.Jacobi())
    bw = poisson.rhs(warm.n, 0)
    t_end = time.perf_counter() + 0.2
    while time.perf_counter() < t_end:
        warm.solve(bw, want_history=False)
    warm.close()

    def solve_twice(system, b, **kw):
        system.solve(b, want_history=False, **kw)
        return system.solve(b, want_history=False, **kw)

    # config 2: 256^2 2-D system, CNN-like factor applied as L (L^T r), IC(0) by triangular solves, Jacobi
    s = poisson.poisson_system(2, 256)
    b = poisson.rhs(s.n, 0)
    c2 = {}
    for name, pc in (("jacobi", D.Jacobi()), ("ic0_solve", D.IC0("solve")),
                     ("ic0_multicolor_solve", D.IC0("solve", ordering="multicolor")),
                     ("ict_multiply_reference_default", D.ICT("multiply", 1, 0.1))):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s.set_preconditioner(pc)            # first attach: module load etc.; the second one is timed (test.py:130-135 `setups`)
        torch.cuda.synchronize()
        first_ms = (time.perf_counter() - t0) * 1e3
        t0 = time.perf_counter()
        s.set_preconditioner(pc)
        torch.cuda.synchronize()
        setup_ms = (time.perf_counter() - t0) * 1e3
        r = solve_twice(s, b)
        c2[name] = {"iterations": r.iterations, "status": r.status, "ms": round(r.seconds * 1e3, 3),
                    "iterations_per_s": round(r.iterations / r.seconds, 1), "setup_ms": round(setup_ms, 2)}
        if name == "ic0_multicolor_solve":  # the colouring belongs to the pattern: the handle keeps it
            c2[name]["setup_new_pattern_ms"] = round(first_ms, 2)
            v2 = poisson.poisson_csr(2, 256)[2]                 # the same values again, as the next time step's would arrive
            for _ in range(2):              # update_values parks the factor; the setup then only computes values (first time: + maps)
                s.update_values(v2)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                s.set_preconditioner(pc)
                torch.cuda.synchronize()
                c2[name]["setup_new_values_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
            r2 = solve_twice(s, b)
            c2[name]["ms_after_new_values_setup"] = round(r2.seconds * 1e3, 3)
        if name in ("ic0_solve", "ic0_multicolor_solve"):
            c2[name]["apply_roofline"] = apply_roofline(s, time_apply(s, b, torch))
        if name == "ic0_solve":
            c2["levels"] = s.info()["levels_lower"]
        if name in ("ic0_solve", "ic0_multicolor_solve"):
            in_loop_apply(c2[name], c2["jacobi"]["ms"] * 1e3 / c2["jacobi"]["iterations"])
    # the CNN-emitted factor (seeded random weights: no checkpoint ships), applied as z = L (L^T r) without densifying
    from deeppreconditioning_amd import model as mdl
    import scipy.sparse as sp
    torch.manual_seed(69)
    net = mdl.PreconditionerNet([1, 16, 32, 64, 32, 16, 1]).cuda()
    n2 = 256
    idx = np.arange(n2 * n2)
    A2 = sp.diags([4.0], [0], shape=(n2 * n2, n2 * n2), format="lil")
    A2 = (sp.diags([np.full(n2 * n2, 4.0), np.where((idx[:-1] + 1) % n2 != 0, -1.0, 0.0), np.full(n2 * n2 - n2, -1.0)],
                   [0, -1, -n2], format="csr"))                       # tril of the 5-point matrix
    inp, sizes = mdl.tril_batch_from_csr([A2], device="cuda")
    for _ in range(3):                                                    # last pass counts: library warm-up excluded
        # a NEW sparsity pattern per matrix (a fresh indices tensor misses the plan cache): building the plan is part of it
        new_inp = mdl.SparseBatch(inp.features, inp.indices.clone(), inp.spatial_shape, inp.batch_size)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.no_grad():
            outL = net(new_inp)
        Lparts = mdl.lower_factor_csr(outL, 0, sizes[0])
        torch.cuda.synchronize()
        fwd_ms = (time.perf_counter() - t0) * 1e3
        s.set_preconditioner(D.LLtMultiply(Lparts))
        torch.cuda.synchronize()
        setup_ms = (time.perf_counter() - t0) * 1e3 - fwd_ms
        r = s.solve(b, want_history=False)
        total_ms = (time.perf_counter() - t0) * 1e3
    with torch.no_grad():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            net(inp)
        torch.cuda.synchronize()
        fwd_cached_ms = (time.perf_counter() - t0) / 5 * 1e3
    # the forward against ITS roofline: fp32 matrix cores (v_mfma_f32_16x16x4_f32: 155 TFLOP/s dense, MI355X_MICROARCH.md) and, beside
    # it, the minimum bytes through HBM; timed with events on the stream the forward is enqueued on (torch's current stream)
    with torch.no_grad():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            net(inp)
        e1.record()
        torch.cuda.synchronize()
    fwd_ev_ms = e0.elapsed_time(e1) / 20
    cost = mdl.forward_cost(net, inp)
    forward_roofline = {"bound": "mfma", "achieved": round(cost["flops"] / fwd_ev_ms / 1e9, 2), "peak": 155.0, "unit": "TFLOP/s",
                        "frac": round(cost["flops"] / fwd_ev_ms / 1e9 / 155.0, 4), "dtype": "f32 (exact fp32 MFMA, no xf32 on gfx950)",
                        "ms_per_forward": round(fwd_ev_ms, 4), "flops_per_forward": cost["flops"],
                        "min_hbm_bytes_per_forward": cost["min_hbm_bytes"],
                        "hbm_frac_at_min_bytes": round(cost["min_hbm_bytes"] / (fwd_ev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                        "layers": cost["layers"],
                        "kernel_trace": "profiles/r04_cnn_kernel_stats.csv (rocprofv3 --kernel-trace --stats of tools/trace_run_cnn.py)"}
    c2["learned_random_weights_llt_multiply"] = {"forward_roofline": forward_roofline, "iterations": r.iterations, "status": r.status, "ms": round(r.seconds * 1e3, 3),
                                                 "iterations_per_s": round(r.iterations / r.seconds, 1),
                                                 "nnz_L": int(Lparts[1].numel()),
                                                 "cnn_forward_ms": round(fwd_ms, 3), "cnn_forward_plan_cached_ms": round(fwd_cached_ms, 3),
                                                 "cnn_path": "hip" if getattr(outL, "lower_csr", None) is not None else "torch",
                                                 "llt_setup_ms": round(setup_ms, 3),
                                                 "end_to_end_forward_setup_solve_ms": round(total_ms, 3)}
    out["c2_poisson2d_256"] = c2
    del s, net, outL, inp
    # IC(0) applied by triangular solves on natural-order 3-D grids: setup (factorisation + level / strip schedules, all
    # on the device), one apply z = L^-T (L^-1 r), and the PCG with it
    trsv = {}
    for n3 in (64, 100):
        s_t = poisson.poisson_system(3, n3)
        b_t = poisson.rhs(s_t.n, 0)
        s_t.set_preconditioner(D.IC0("solve"))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s_t.set_preconditioner(D.IC0("solve"))
        torch.cuda.synchronize()
        setup_ms = (time.perf_counter() - t0) * 1e3
        s_t.precond_apply(b_t)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            s_t.precond_apply(b_t)
        torch.cuda.synchronize()
        apply_us = (time.perf_counter() - t0) / 20 * 1e6
        r = solve_twice(s_t, b_t)
        trsv[f"poisson3d_{n3}"] = {"rows": s_t.n, "levels": s_t.info()["levels_lower"], "setup_ms": round(setup_ms, 2),
                                   "apply_us": round(apply_us, 1), "pcg_iterations": r.iterations,
                                   "pcg_us_per_update": round(r.seconds / r.iterations * 1e6, 1), "pcg_ms": round(r.seconds * 1e3, 3),
                                   "apply_roofline": apply_roofline(s_t, apply_us)}
        # the same factorisation in multicolour (here: red-black) order: 2 levels instead of hundreds
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s_t.set_preconditioner(D.IC0("solve", ordering="multicolor"))
        torch.cuda.synchronize()
        first_ms = (time.perf_counter() - t0) * 1e3
        t0 = time.perf_counter()
        s_t.set_preconditioner(D.IC0("solve", ordering="multicolor"))
        torch.cuda.synchronize()
        setup_ms = (time.perf_counter() - t0) * 1e3
        apply_us = time_apply(s_t, b_t, torch)
        n_colors = s_t.precond_ordering()[0]
        r = solve_twice(s_t, b_t)
        s_t.set_preconditioner(D.Jacobi())
        rj = solve_twice(s_t, b_t)
        trsv[f"poisson3d_{n3}_multicolor"] = {"colors": n_colors, "setup_ms": round(setup_ms, 2), "setup_new_pattern_ms": round(first_ms, 2),
                                              "apply_us": round(apply_us, 1), "pcg_iterations": r.iterations,
                                              "pcg_us_per_update": round(r.seconds / r.iterations * 1e6, 1),
                                              "pcg_ms": round(r.seconds * 1e3, 3), "jacobi_pcg_ms": round(rj.seconds * 1e3, 3),
                                              "jacobi_iterations": rj.iterations}
        s_t.close()
    out["ic0_triangular_solves_natural_order"] = trsv
    # 1M-DoF 2-D system: hits max_iter = 1024 like the reference (fixed-work throughput)
    s2 = poisson.poisson_system(2, 1024)
    s2.set_preconditioner(D.Jacobi())
    b2 = poisson.rhs(s2.n, 0)
    r = solve_twice(s2, b2)
    ms = s2.spmv_dot_bench(200)
    out["poisson2d_1024_jacobi"] = {"iterations": r.iterations, "iterations_per_s": round(r.iterations / r.seconds, 1),
                                    "spmv_gbs": round(loop_kernel_bytes(s2) / (ms * 1e-3) / 1e9, 1)}
    del s2
    # config 3: ~1M-DoF unstructured stand-in (random symmetric permutation + SPD scaling of the 3-D matrix) through the
    # PLAIN call: the library measures the x-gather traffic, reorders (reverse Cuthill-McKee on the device) and takes
    # the x-tile SpMV; b / x / the IC(0) factor stay in the caller's numbering
    A = poisson.unstructured_like_csr(3, 100, 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s3 = D.CsrSystem.from_any(A)
    torch.cuda.synchronize()
    create_s = time.perf_counter() - t0
    b3 = poisson.rhs(s3.n, 0)
    c3 = {"create_incl_upload_and_reordering_ms": round(create_s * 1e3, 1), "reordered": s3.info()["reordered"],
          "gather_ratio_before": round(s3.info()["gather_ratio"], 2), "spmv_kernel": s3.info()["spmv_kernel"]}
    # the NEXT pressure system of the same mesh: new values on the same pattern keep the plan and the reordering
    vals_dev = torch.from_numpy(A.data).cuda()
    for tag, v in (("update_values_from_host_ms", A.data), ("update_values_from_device_ms", vals_dev)):
        s3.update_values(v)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s3.update_values(v)
        torch.cuda.synchronize()
        c3[tag] = round((time.perf_counter() - t0) * 1e3, 2)
    for name, pc in (("jacobi", D.Jacobi()), ("ic0_multicolor_solve", D.IC0("solve", ordering="multicolor")),
                     ("ic0_solve", D.IC0("solve"))):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s3.set_preconditioner(pc)
        torch.cuda.synchronize()
        setup_ms = (time.perf_counter() - t0) * 1e3
        r = solve_twice(s3, b3)
        c3[name] = {"iterations": r.iterations, "ms": round(r.seconds * 1e3, 3), "iterations_per_s": round(r.iterations / r.seconds, 1),
                    "us_per_update": round(r.seconds / r.iterations * 1e6, 1), "setup_ms": round(setup_ms, 2)}
        if name != "jacobi":
            c3[name]["apply_roofline"] = apply_roofline(s3, time_apply(s3, b3, torch))
            c3[name]["levels"] = s3.info()["levels_lower"]
            in_loop_apply(c3[name], c3["jacobi"]["us_per_update"])
        if name == "ic0_multicolor_solve":   # that was the first attach on this pattern (colouring included); again, colouring kept:
            c3[name]["setup_new_pattern_ms"] = c3[name]["setup_ms"]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            s3.set_preconditioner(pc)
            torch.cuda.synchronize()
            c3[name]["setup_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
            # the next system of the same mesh: update_values parks the factor, the setup only computes values again
            for _ in range(2):                    # (the first such setup builds the entry maps)
                s3.update_values(vals_dev)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                s3.set_preconditioner(pc)
                torch.cuda.synchronize()
                c3[name]["setup_new_values_ms"] = round((time.perf_counter() - t0) * 1e3, 2)
            r = solve_twice(s3, b3)
            c3[name]["ms_after_new_values_setup"] = round(r.seconds * 1e3, 3)
    c3["levels"] = s3.info()["levels_lower"]
    # config 5 AS BASELINE STATES IT: mixed fp32-SpMV / fp64 PCG on this 1M-DoF unstructured system (values not
    # fp32-representable; the fp32 copy is made from the reordered matrix), residual-matched to the fp64 run
    s3.set_preconditioner(D.Jacobi())
    r64 = solve_twice(s3, b3)
    r32 = solve_twice(s3, b3, flags=D._lib.SPMV_F32)
    rt = b3 - s3 @ r32.x
    out["c5_mixed_precision_unstructured3d_100"] = {
        "iterations_fp64": r64.iterations, "iterations_mixed": r32.iterations, "final_res_fp64": r64.final_res,
        "final_res_mixed_recurrence": r32.final_res, "final_res_mixed_true_fp64_residual": D.dot(rt, rt) / D.dot(b3, b3),
        "iterations_per_s_fp64": round(r64.iterations / r64.seconds, 1),
        "iterations_per_s_mixed": round(r32.iterations / r32.seconds, 1),
        "spmv_algorithmic_bytes_mixed": spmv_bytes(s3.n, s3.nnz, wv=4, wx=4) + 4 * s3.n}   # y stays fp64: + 4 n
    ms = s3.spmv_dot_bench(100)
    c3["spmv_gbs"] = round(loop_kernel_bytes(s3) / (ms * 1e-3) / 1e9, 1)
    c3["spmv_frac_of_hbm_peak"] = round(c3["spmv_gbs"] / HBM_PEAK_GBS, 4)
    pmc = ROOT / "profiles" / "pmc_traffic.json"
    pmc_all = json.loads(pmc.read_text()) if pmc.exists() else {}
    c3["spmv_traffic_bytes_per_launch"] = pmc_all.get("spmv_scrambled3d_100_reordered")
    c3["spmv_algorithmic_bytes_per_launch"] = loop_kernel_bytes(s3)
    # the same system WITHOUT reordering (reorder=None): the gather SpMV on the scrambled numbering
    s3n = D.CsrSystem.from_any(A, reorder=None)
    s3n.set_preconditioner(D.Jacobi())
    r = solve_twice(s3n, b3)
    ms = s3n.spmv_dot_bench(100)
    c3["not_reordered"] = {"jacobi_iterations": r.iterations, "jacobi_iterations_per_s": round(r.iterations / r.seconds, 1),
                           "spmv_gbs": round(loop_kernel_bytes(s3n) / (ms * 1e-3) / 1e9, 1),
                           "spmv_traffic_bytes_per_launch": pmc_all.get("spmv_scrambled3d_100_gather")}
    s3n.set_preconditioner(D.IC0("solve"))
    r = solve_twice(s3n, b3)
    c3["not_reordered"]["ic0_solve_us_per_update"] = round(r.seconds / r.iterations * 1e6, 1)
    c3["time_to_solution_ms"] = {k: c3[k]["ms"] for k in ("jacobi", "ic0_multicolor_solve", "ic0_solve")}
    c3["triangular_solve_form"] = ("launches: at 1M rows (8 rows a thread) the factor does not fit on the chip beside the matrix, and the streamed "
                                   "one-launch form loses to the launches (66 against 58 us per update) -- see c3_unstructured3d_80 for the resident form")
    out["c3_unstructured3d_100"] = c3
    del s3, s3n, A
    # config 3 at the capacity of the one-launch kernel WITH the triangular solves inside (dpcg_chip_trsv.hip: <= 524 288 rows, matrix in
    # registers, L / L^T in LDS): the same stand-in at 80^3 = 512 000 rows, and the 216 000-row one; plain calls, the launches beside them
    for tag, grid in (("c3_unstructured3d_80", 80), ("c3_unstructured3d_60", 60)):
        Ah = poisson.unstructured_like_csr(3, grid, 0)
        sh = D.CsrSystem.from_any(Ah)
        bh = poisson.rhs(sh.n, 0)
        ch = {"dof": sh.n, "nnz": sh.nnz, "reordered": sh.info()["reordered"]}
        for name, pc in (("jacobi", D.Jacobi()), ("ic0_multicolor_solve", D.IC0("solve", ordering="multicolor")), ("ic0_solve", D.IC0("solve"))):
            sh.set_preconditioner(pc)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            sh.set_preconditioner(pc)
            one_launch = sh.chip_info()["chip_by_default"]            # (for the triangular solves this also builds the kernel's plan: part of the setup)
            torch.cuda.synchronize()
            setup_ms = (time.perf_counter() - t0) * 1e3
            r = solve_twice(sh, bh)
            rl = solve_twice(sh, bh, flags=D._lib.NO_SMALL)
            ch[name] = {"iterations": r.iterations, "one_launch": bool(one_launch), "ms": round(r.seconds * 1e3, 3),
                        "us_per_update": round(r.seconds / r.iterations * 1e6, 2), "iterations_per_s": round(r.iterations / r.seconds, 1),
                        "setup_ms": round(setup_ms, 2), "levels": sh.info()["levels_lower"] if name != "jacobi" else None,
                        "launches_us_per_update": round(rl.seconds / rl.iterations * 1e6, 2), "launches_ms": round(rl.seconds * 1e3, 3)}
        ch["time_to_solution_ms"] = {k: ch[k]["ms"] for k in ("jacobi", "ic0_multicolor_solve", "ic0_solve")}
        ch["time_to_solution_launches_ms"] = {k: ch[k]["launches_ms"] for k in ("jacobi", "ic0_multicolor_solve", "ic0_solve")}
        out[tag] = ch
        sh.close()
        del sh, Ah
    out.update(mesh_workloads(D, poisson, torch, solve_twice, pmc_all))
    # config 5: mixed fp32 SpMV / fp64 everything else on the headline system
    s5 = poisson.poisson_system(3, 100)
    s5.set_preconditioner(D.Jacobi())
    b5 = poisson.rhs(s5.n, 0)
    r64 = solve_twice(s5, b5)
    r32 = solve_twice(s5, b5, flags=D._lib.SPMV_F32)
    out["c5_mixed_precision_poisson3d_100"] = {"iterations_fp64": r64.iterations, "iterations_mixed": r32.iterations,
                                               "final_res_fp64": r64.final_res, "final_res_mixed": r32.final_res,
                                               "iterations_per_s_mixed": round(r32.iterations / r32.seconds, 1)}
    # opt-in, for the STREAMING kernels (systems the one-launch kernel does not take: it keeps the matrix on chip): matrix values streamed
    # as fp32 because they are fp32-representable -- bit-identical to the same kernels on fp64 values
    rc = solve_twice(s5, b5, flags=D._lib.VAL32_IF_LOSSLESS)
    rl = solve_twice(s5, b5, flags=D._lib.NO_SMALL)
    out["lossless_fp32_value_storage_poisson3d_100"] = {
        "path": "multi-launch (streaming SpMV)", "iterations": rc.iterations, "final_res": rc.final_res,
        "bitwise_equal_to_fp64_values": bool(rc.final_res == rl.final_res), "iterations_per_s": round(rc.iterations / rc.seconds, 1),
        "iterations_per_s_fp64_values_same_path": round(rl.iterations / rl.seconds, 1), "spmv_bytes_per_launch": spmv_bytes(s5.n, s5.nnz, wv=4)}
    del s5
    # the reference's real size class (2.4k-5.5k rows): a batch of 256 independent systems,
    # one launch, one workgroup per system
    from deeppreconditioning_amd.batch import solve_batch
    systems = [poisson.poisson_system(2, 49 + (i % 4)) for i in range(256)]
    for sy in systems:
        sy.set_preconditioner(D.Jacobi())
    rhs_b = [poisson.rhs(sy.n, i) for i, sy in enumerate(systems)]
    solve_batch(systems, rhs_b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res_b = solve_batch(systems, rhs_b)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    one = systems[0].solve(rhs_b[0], want_history=False)
    one = systems[0].solve(rhs_b[0], want_history=False)
    out["batch_256_systems_2401_to_2704_rows"] = {"batch_ms": round(dt * 1e3, 3), "iterations_total": int(sum(r.iterations for r in res_b)),
                                                 "iterations_per_s_aggregate": round(sum(r.iterations for r in res_b) / dt, 1),
                                                 "systems_per_s": round(256 / dt, 1),
                                                 "single_system_us_per_update": round(one.seconds / one.iterations * 1e6, 2)}
    del systems, rhs_b
    # independent systems interleaved on several streams (dpcg_solve_batch, general path): another system's kernels
    # fill the launch boundaries and ramps of this one's -- aggregate rate vs one after another
    conc = {}
    for label, dim_c, n_c, count in (("4x_poisson3d_100", 3, 100, 4), ("8x_poisson2d_256", 2, 256, 8)):
        group = [poisson.poisson_system(dim_c, n_c) for _ in range(count)]
        for sy in group:
            sy.set_preconditioner(D.Jacobi())
        rhs_c = [poisson.rhs(sy.n, i) for i, sy in enumerate(group)]
        entry = {}
        # multi-launch path interleaved on 4 streams; for the mid-size systems also the default of a batch: one launch, one
        # team of 32 workgroups per system (dpcg_team.hip)
        for tag, fl in (("interleaved_4_streams", D._lib.NO_SMALL | D._lib.NO_TEAM), ("default_batch", 0)):
            solve_batch(group, rhs_c, n_streams=4, flags=fl)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            res_c = solve_batch(group, rhs_c, n_streams=4, flags=fl)
            torch.cuda.synchronize()
            dt_c = time.perf_counter() - t0
            entry[f"iterations_per_s_{tag}"] = round(sum(r.iterations for r in res_c) / dt_c, 1)
        entry["default_batch_is"] = ("team kernel (one launch, one team per system)" if group[0].n <= 65536 else
                                     ("whole-chip kernel, one system after another" if group[0].chip_info()["chip_by_default"]
                                      else "multi-launch path, interleaved"))
        t0 = time.perf_counter()
        seq_its = sum(sy.solve(b_c, want_history=False, flags=D._lib.NO_SMALL).iterations for sy, b_c in zip(group, rhs_c))
        torch.cuda.synchronize()
        dt_s = time.perf_counter() - t0
        entry["iterations_per_s_one_after_another"] = round(seq_its / dt_s, 1)          # (the multi-launch path, DPCG_NO_SMALL)
        conc[label] = entry
        for sy in group:
            sy.close()
        del group, rhs_c
    out["independent_systems_interleaved"] = conc
    # config 4 (one GPU's share): 256^3 systems, 1.74 GB per SpMV, beyond the Infinity Cache -> HBM-bound
    s4 = poisson.poisson_system(3, 256)
    s4.set_preconditioner(D.Jacobi())
    b4 = poisson.rhs(s4.n, 0)
    s4.solve(b4, max_iter=16, want_history=False)
    r = s4.solve(b4, max_iter=64, want_history=False)
    ms = s4.spmv_dot_bench(50)
    gbs = loop_kernel_bytes(s4) / (ms * 1e-3) / 1e9
    out["c4_poisson3d_256_jacobi"] = {"iterations": r.iterations, "iterations_per_s": round(r.iterations / r.seconds, 2),
                                      "spmv_gbs": round(gbs, 1), "spmv_frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 4),
                                      "spmv_us_per_launch": round(ms * 1e3, 1)}
    # the same system to the solution (reference defaults): Jacobi against IC(0) in multicolour order, every stream from HBM
    to_solution = {}
    for name, pc in (("jacobi", D.Jacobi()), ("ic0_multicolor_solve", D.IC0("solve", ordering="multicolor"))):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s4.set_preconditioner(pc)
        torch.cuda.synchronize()
        setup_ms = (time.perf_counter() - t0) * 1e3
        s4.solve(b4, want_history=False)
        r = s4.solve(b4, want_history=False)
        to_solution[name] = {"iterations": r.iterations, "status": r.status, "ms": round(r.seconds * 1e3, 2),
                             "us_per_update": round(r.seconds / r.iterations * 1e6, 1), "setup_new_pattern_ms": round(setup_ms, 2)}
    to_solution["ic0_multicolor_solve"]["apply_roofline"] = {
        "bound": "hbm", "algorithmic_bytes_per_apply": 2 * sptrsv_bytes(s4.n, s4.info()["precond_nnz"]),
        "levels": [s4.info()["levels_lower"], s4.info()["levels_upper"]]}
    in_loop_apply(to_solution["ic0_multicolor_solve"], to_solution["jacobi"]["us_per_update"])
    out["c4_poisson3d_256_to_solution"] = to_solution
    s4.close()
    del s4, b4
    # config 4 as ONE GPU sees it: its 8 of the 64 systems (s mod 8 == rank), full solves at the reference defaults,
    # through the batch path the multi-GPU driver uses -- one after another and four in flight
    from deeppreconditioning_amd.batch import SystemSpec, shard, solve_specs_local
    specs = [SystemSpec(3, 256, sid) for sid in shard(64, 0, 8)]
    share = {}
    for conc in (1, 4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        recs = solve_specs_local(specs, concurrent=conc)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        share[f"concurrent_{conc}"] = {"iterations_total": int(recs[:, 0].sum()), "all_converged": bool((recs[:, 1] == 0).all()),
                                       "wall_s_incl_generation_and_setup": round(dt, 3),
                                       "iterations_per_s_solve_time_only": round(float(recs[:, 0].sum() / (recs[:, 3].sum() / conc)), 1),
                                       "iterations_per_s_wall": round(float(recs[:, 0].sum() / dt), 1)}
    share["iterations_per_system"] = [int(v) for v in recs[:, 0]]
    out["c4_one_gpu_share_8x_poisson3d_256"] = share
    return out


if __name__ == "__main__":
    main()
