"""Device memory after repeated setups through round 4's new paths (icholt, peel colouring + fold, the CSR-stream sync-free form,
MIX tile plans): must stay flat (the library's block cache is released before looking).   python tools/leak_check_r04.py"""
import gc
import os

os.environ.setdefault("DPCG_TILE_MIX_MAX", "30")
import torch

import deeppreconditioning_amd as D
from deeppreconditioning_amd import meshes, poisson
from deeppreconditioning_amd.operators import release_cached_memory


def used():
    torch.cuda.synchronize()
    release_cached_memory()
    f, t = torch.cuda.mem_get_info()
    return (t - f) / 1e6


mats = [meshes.quadtree_fv_laplacian(260, 1), meshes.quadtree_fv_laplacian(260, 2, numbering="random"), meshes.delaunay_laplacian(90000, 3),
        meshes.quadtree_fv_laplacian(60, 4)]
base = None
for rep in range(4):
    for i in range(12):
        A = mats[i % len(mats)]
        S = D.CsrSystem.from_any(A, reorder=("auto", "rcm", None)[i % 3])
        b = poisson.rhs(S.n, i)
        pcs = [D.Jacobi(), D.IC0("solve"), D.IC0("solve", ordering="multicolor"), D.ICT("solve", 1, 0.01)]
        if S.n < 40000:
            pcs.append(D.ICholT("solve", 1, 0.1))
        for pc in pcs:
            S.set_preconditioner(pc)
            S.solve(b, max_iter=15, want_history=False, flags=D._lib.NO_SMALL)
        S.close()
        del S, b
    gc.collect()
    torch.cuda.empty_cache()
    u = used()
    base = u if base is None else base
    print(f"after round {rep}: {u - base:+.1f} MB against the first round", flush=True)
