"""Setup time of ICholT (ilupp.icholt as ILU++ defines it: one wave walks the columns) by size, beside IC(0) and the level-1 ICT.
python tools/icholt_probe.py"""
import time

import torch

import deeppreconditioning_amd as D
from deeppreconditioning_amd import meshes, poisson

for name, make in (("poisson2d_49 (2.4K, the reference's size)", lambda: poisson.poisson_system(2, 49)),
                   ("quadtree 2.2K", lambda: D.CsrSystem.from_any(meshes.quadtree_fv_laplacian(45, 3))),
                   ("poisson2d_150 (22K)", lambda: poisson.poisson_system(2, 150)),
                   ("poisson2d_256 (65K)", lambda: poisson.poisson_system(2, 256)),
                   ("poisson3d_40 (64K)", lambda: poisson.poisson_system(3, 40))):
    S = make()
    b = poisson.rhs(S.n, 0)
    for label, pc in (("icholt(1, 0.1) solve", lambda: D.ICholT("solve", 1, 0.1)), ("ic0 solve", lambda: D.IC0("solve")),
                      ("ict level-1 (1, 0.1) solve", lambda: D.ICT("solve", 1, 0.1))):
        S.set_preconditioner(pc())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        S.set_preconditioner(pc())
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3
        r = S.solve(b, want_history=False)
        print(f"{name}: {label}: setup {ms:.2f} ms ({ms * 1e3 / S.n:.2f} us per row), nnz(L) {S.info()['precond_nnz']}, {r.iterations} its, {r.seconds * 1e3:.2f} ms", flush=True)
    S.close()
