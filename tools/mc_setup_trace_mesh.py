"""Where the multicolour IC(0) setup goes on the 1M-row mesh systems:  DPCG_SETUP_TRACE=1 python tools/mc_setup_trace_mesh.py"""
import sys
import time

import torch

import deeppreconditioning_amd as D
from deeppreconditioning_amd import meshes

for name, make in (("delaunay", lambda: meshes.delaunay_laplacian(1000000, 0)),
                   ("quadtree_random", lambda: meshes.quadtree_fv_laplacian(1000, 0, numbering="random"))):
    if sys.argv[1:] and name not in sys.argv[1:]:
        continue
    A = make()
    s = D.CsrSystem.from_any(A)
    s.set_preconditioner(D.IC0("solve", ordering="multicolor"))      # (warm-up: allocator, kernels)
    s.close()
    s = D.CsrSystem.from_any(A)                                       # a new handle: nothing of the pattern is known
    torch.cuda.synchronize()
    print(f"== {name}", file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    s.set_preconditioner(D.IC0("solve", ordering="multicolor"))
    torch.cuda.synchronize()
    print(f"== {name}: {1e3 * (time.perf_counter() - t0):.2f} ms, {s.info()}", file=sys.stderr, flush=True)
    s.close()
