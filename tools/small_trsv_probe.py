import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import deeppreconditioning_amd as D
from oracle import oracle as O
for name, A in (("quadtree_40 (2.2K)", O.quadtree_fv_laplacian(40, 1)), ("quadtree_64 (5.5K)", O.quadtree_fv_laplacian(64, 1)), ("poisson2d_100 (10K)", O.poisson2d(100)), ("poisson3d_24 (13.8K)", O.poisson3d(24))):
    n = A.shape[0]
    S = D.CsrSystem.from_any(A)
    b = torch.from_numpy(O.rhs(n, 0)).cuda()
    S.set_preconditioner(D.IC0("solve", ordering="multicolor"))
    ci = S.chip_info()
    r = S.solve(b); r = S.solve(b)
    m = S.solve(b, flags=D._lib.NO_SMALL); m = S.solve(b, flags=D._lib.NO_SMALL)
    print(f"{name}: n {n} chip_by_default {ci['chip_by_default']}: plain {r.seconds / r.iterations * 1e6:.2f} us per update ({r.iterations}), launches {m.seconds / m.iterations * 1e6:.2f}", flush=True)
    S.close()
