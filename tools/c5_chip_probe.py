"""BASELINE config 5 (fp32-stored A and p in `A @ p`, fp64 everything else) through the one-launch kernel beside fp64 and the launches.
    python tools/c5_chip_probe.py"""
import numpy as np
import torch

import deeppreconditioning_amd as D
from deeppreconditioning_amd import poisson

F32, NO_SMALL = D._lib.SPMV_F32, D._lib.NO_SMALL
for name, make in (("poisson3d_100", lambda: poisson.poisson_system(3, 100)),
                   ("unstructured3d_100", lambda: D.CsrSystem.from_any(poisson.unstructured_like_csr(3, 100))),
                   ("poisson2d_1024", lambda: poisson.poisson_system(2, 1024))):
    S = make()
    b = poisson.rhs(S.n, 0)
    S.set_preconditioner(D.Jacobi())
    for label, flags in (("fp64 one launch", 0), ("mixed one launch", F32), ("mixed launches", F32 | NO_SMALL), ("fp64 launches", NO_SMALL)):
        best = None
        for _ in range(5):
            r = S.solve(b, flags=flags, want_history=False)
            best = r if best is None or r.seconds < best.seconds else best
        print(f"{name}: {label}: {best.iterations} updates, {best.seconds * 1e3:.3f} ms, {best.seconds * 1e6 / best.iterations:.2f} us per update, "
              f"{best.iterations / best.seconds:.0f} it/s, final {best.final_res:.6e}", flush=True)
    S.close()
