"""RCM with narrow levels walked by one workgroup (DPCG_RCM_NARROW=1, default) against a launch per level (=0): the permutation must be
the same, the time is not.   for k in 0 1; do DPCG_RCM_NARROW=$k python tools/rcm_narrow_ab.py; done"""
import hashlib
import os
import time

import numpy as np
import scipy.sparse as sp
import torch

import deeppreconditioning_amd as D
from deeppreconditioning_amd import meshes, poisson


def scrambled(A, seed):
    rng = np.random.default_rng(seed)
    q = rng.permutation(A.shape[0])
    B = A[q][:, q].tocsr()
    B.sort_indices()
    return B


def grid2d(m):
    T = sp.diags([-1, 2, -1], [-1, 0, 1], shape=(m, m))
    return (sp.kron(sp.eye(m), T) + sp.kron(T, sp.eye(m))).tocsr()


cases = [("quadtree_random 1M", lambda: meshes.quadtree_fv_laplacian(1000, 0, numbering="random")),
         ("delaunay 1M", lambda: meshes.delaunay_laplacian(1000000, 0)),
         ("scrambled 100^3", lambda: poisson.unstructured_like_csr(3, 100, 0)),
         ("scrambled 600^2", lambda: scrambled(grid2d(600), 3)),
         ("quadtree_random 60K", lambda: meshes.quadtree_fv_laplacian(240, 5, numbering="random")),
         ("two components", lambda: scrambled(sp.block_diag([grid2d(150), grid2d(90)]).tocsr(), 7)),
         ("path 5000 + grid", lambda: scrambled(sp.block_diag([sp.diags([-1, 2, -1], [-1, 0, 1], shape=(5000, 5000)), grid2d(64)]).tocsr(), 9))]
print("DPCG_RCM_NARROW =", os.environ.get("DPCG_RCM_NARROW", "(default: 1)"), " DPCG_RCM_SWEEPS =", os.environ.get("DPCG_RCM_SWEEPS", "(default: 1)"))
for name, make in cases:
    A = make()
    s = D.CsrSystem.from_any(A, reorder="rcm")
    s.close()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s = D.CsrSystem.from_any(A, reorder="rcm")
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    s.spmv_dot_bench(20)
    spmv_us = s.spmv_dot_bench(200) * 1e3
    perm = s.permutation()
    assert np.array_equal(np.sort(perm), np.arange(A.shape[0]))
    print(f"{name}: n {A.shape[0]}, create + RCM {ms:.1f} ms, permutation sha1 {hashlib.sha1(np.ascontiguousarray(perm).tobytes()).hexdigest()[:16]}, "
          f"kernel {s.info()['spmv_kernel']}, SpMV {spmv_us:.2f} us", flush=True)
    s.close()
