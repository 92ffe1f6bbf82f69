"""Config 3 on genuinely unstructured matrices (deeppreconditioning_amd/meshes.py): what the library does with them.
Per system: create time, SpMV kernel chosen, gather ratio before / after reordering, SpMV time and roofline, bit-exactness of
`A @ x` against scipy on the iterated numbering, Jacobi / IC(0) caller / IC(0) multicolour: setup, levels, colours, per update,
to the solution.    python tools/mesh_probe.py [quadtree_foam quadtree_random delaunay] [--m 1000] [--n 1000000]"""
import argparse
import time

import numpy as np
import torch

import deeppreconditioning_amd as D
from deeppreconditioning_amd import meshes, poisson

ap = argparse.ArgumentParser()
ap.add_argument("cases", nargs="*")
ap.add_argument("--m", type=int, default=1000)
ap.add_argument("--n", type=int, default=1000000)
ap.add_argument("--no-ic0", action="store_true")
args = ap.parse_args()

makers = {"quadtree_foam": lambda: meshes.quadtree_fv_laplacian(args.m, 0, numbering="foam"),
          "quadtree_random": lambda: meshes.quadtree_fv_laplacian(args.m, 0, numbering="random"),
          "delaunay": lambda: meshes.delaunay_laplacian(args.n, 0)}
for name in (args.cases or list(makers)):
    t0 = time.perf_counter()
    A = makers[name]()
    gen_s = time.perf_counter() - t0
    n, nnz = A.shape[0], A.nnz
    deg = np.diff(A.indptr)
    print(f"== {name}: n {n} nnz {nnz} ({nnz / n:.2f}/row, rows of {deg.min()}..{deg.max()}), generated in {gen_s:.1f} s", flush=True)
    algo = nnz * 12 + (n + 1) * 4 + 16 * n
    for reorder in (None, "auto", "rcm"):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s = D.CsrSystem.from_any(A, reorder=reorder)
        torch.cuda.synchronize()
        create_ms = (time.perf_counter() - t0) * 1e3
        info = s.info()
        s.spmv_dot_bench(20)
        us = s.spmv_dot_bench(200) * 1e3
        x = torch.from_numpy(np.random.default_rng(1).uniform(-1, 1, n)).cuda()
        y = (s @ x).cpu().numpy()
        perm = s.permutation() if s.reordered else np.arange(n)
        Bp = A[perm][:, perm].tocsr()
        Bp.sort_indices()
        exact = bool(np.array_equal(y[perm], Bp @ x.cpu().numpy()[perm]))
        print(f"  reorder={reorder}: create {create_ms:.1f} ms, kernel {info['spmv_kernel']}, reordered {info['reordered']}, "
              f"gather_ratio {info['gather_ratio']:.2f}, SpMV {us:.1f} us = {algo / us / 1e3:.0f} GB/s = {algo / us / 1e3 / 8000:.3f} "
              f"of 8 TB/s, bits equal scipy's on the iterated matrix: {exact}", flush=True)
        if reorder != "auto":
            s.close()
            continue
        b = poisson.rhs(n, 0)
        todo = [("jacobi", lambda: D.Jacobi())]
        if not args.no_ic0:
            todo += [("ic0 caller", lambda: D.IC0("solve")), ("ic0 multicolour", lambda: D.IC0("solve", ordering="multicolor"))]
        for label, pc in todo:
            s.set_preconditioner(pc())
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            s.set_preconditioner(pc())
            torch.cuda.synchronize()
            setup_ms = (time.perf_counter() - t0) * 1e3
            inf = s.info()
            nc = s.precond_ordering()[0] if "multicolour" in label else 0
            s.solve(b, want_history=False)
            r = s.solve(b, want_history=False)
            print(f"    {label}: setup {setup_ms:.2f} ms, levels {inf['levels_lower']}, colours {nc}, {r.iterations} its, status {r.status}, "
                  f"{r.seconds * 1e3:.2f} ms, {r.seconds / max(r.iterations, 1) * 1e6:.1f} us/update, res {r.final_res:.3e}", flush=True)
        s.close()
