"""Config 5 (fp32-stored SpMV operands, fp64 everything else) on the 1M-row quadtree meshes: the whole-chip kernel with 4-byte value slots
(resident) against fp64 (streamed form) and against the launches:  python tools/c5_mesh_probe.py"""
import pathlib
import sys

import numpy as np
import torch

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import deeppreconditioning_amd as D  # noqa: E402
from oracle import oracle as O  # noqa: E402

for name, A in (("quadtree_random_1M", O.quadtree_fv_laplacian(1000, 0, numbering="random")), ("quadtree_foam_1M", O.quadtree_fv_laplacian(1000, 0))):
    n = A.shape[0]
    S = D.CsrSystem.from_any(A)
    S.set_preconditioner(D.Jacobi())
    b = torch.from_numpy(O.rhs(n, 0)).cuda()
    ci = S.chip_info()
    out = []
    for tag, flags in (("fp64", 0), ("mixed", D._lib.SPMV_F32), ("mixed, launches", D._lib.SPMV_F32 | D._lib.NO_SMALL)):
        S.solve(b, max_iter=300, flags=flags, want_history=False)
        r = S.solve(b, max_iter=300, flags=flags, want_history=False)
        out.append(f"{tag}: {r.seconds / r.iterations * 1e6:.2f} us per update ({r.iterations / r.seconds:.0f} it/s)")
    print(f"{name}: n {n}, rows of up to {ci['max_row_len']}, band {ci['max_band']}: " + "; ".join(out), flush=True)
    S.close()
