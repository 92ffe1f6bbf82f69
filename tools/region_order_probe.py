"""The region-by-region numbering (dpcg_reorder.hip::region_order) beside reverse Cuthill-McKee and no reordering, on the 1M-row
meshes: time of the create (reordering included), SpMV kernel chosen and its rate against the 8 TB/s roofline, chunks of x a 256-row
block touches, bits of `A @ x` against scipy on the iterated numbering, Jacobi updates to the solution.
    python tools/region_order_probe.py [quadtree_foam quadtree_random delaunay scrambled3d] [--m 1000] [--n 1000000]
(DPCG_SETUP_TRACE=1 prints the phases of the reordering)"""
import argparse
import time

import numpy as np
import scipy.sparse as sp
import torch

import deeppreconditioning_amd as D
from deeppreconditioning_amd import meshes, poisson

ap = argparse.ArgumentParser()
ap.add_argument("cases", nargs="*")
ap.add_argument("--m", type=int, default=1000)
ap.add_argument("--n", type=int, default=1000000)
ap.add_argument("--modes", default="none,auto,regions,rcm")
args = ap.parse_args()




def chunks(A):
    n = A.shape[0]
    blk = np.repeat(np.arange(n), np.diff(A.indptr)) // 256
    u = np.unique(blk.astype(np.int64) * (n // 64 + 2) + A.indices // 64)
    per = np.bincount(u // (n // 64 + 2), minlength=(n + 255) // 256)
    return per.mean(), per.max(), int((per > 40).sum())


makers = {"quadtree_foam": lambda: meshes.quadtree_fv_laplacian(args.m, 0, numbering="foam"),
          "quadtree_random": lambda: meshes.quadtree_fv_laplacian(args.m, 0, numbering="random"),
          "delaunay": lambda: meshes.delaunay_laplacian(args.n, 0),
          "scrambled3d_100": lambda: poisson.unstructured_like_csr(3, 100), "scrambled3d_160": lambda: poisson.unstructured_like_csr(3, 160),
          "scrambled2d_1024": lambda: poisson.unstructured_like_csr(2, 1024), "scrambled2d_700": lambda: poisson.unstructured_like_csr(2, 700)}
for name in (args.cases or list(makers)[:3]):
    A = sp.csr_matrix(makers[name]())
    A.sort_indices()
    n, nnz = A.shape[0], A.nnz
    print(f"== {name}: n {n} nnz {nnz} ({nnz / n:.2f}/row)", flush=True)
    algo = nnz * 12 + (n + 1) * 4 + 16 * n
    b = poisson.rhs(n, 0)
    for mode in args.modes.split(","):
        reorder = None if mode == "none" else mode
        D.CsrSystem.from_any(A, reorder=reorder).close()              # (allocations of the first create)
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
        mean, mx, over = chunks(Bp)
        s.set_preconditioner(D.Jacobi())
        r = s.solve(b, want_history=False)
        print(f"  reorder={mode}: create {create_ms:.1f} ms, kernel {info['spmv_kernel']}, reordered {info['reordered']}, gather_ratio "
              f"{info['gather_ratio']:.2f}, chunks/block {mean:.1f} (max {mx}, {over} blocks over 40), SpMV {us:.1f} us = "
              f"{algo / us / 1e3 / 8000:.3f} of 8 TB/s, bits equal scipy's: {exact}; jacobi {r.iterations} updates "
              f"{r.seconds * 1e3:.1f} ms ({r.seconds * 1e6 / max(r.iterations, 1):.1f} us/update)", flush=True)
        s.close()
