"""The one-launch kernel with the matrix streamed (k_pcg_chip MODE 5) below 524 289 rows: Delaunay graphs (rows of up to ~20 entries), default call
beside the launches.    python tools/stream_small_probe.py"""
import os, numpy as np, torch
import deeppreconditioning_amd as D
from deeppreconditioning_amd import meshes, poisson
for n in (100000, 250000, 500000):
    A = meshes.delaunay_laplacian(n, 1)
    S = D.CsrSystem.from_any(A)
    b = poisson.rhs(S.n, 0)
    S.set_preconditioner(D.Jacobi())
    for flags, label in ((0, "default"), (D._lib.NO_SMALL, "launches")):
        best = None
        for _ in range(3):
            r = S.solve(b, flags=flags, want_history=False, max_iter=400)
            best = r if best is None or r.seconds < best.seconds else best
        print(f"delaunay n={n} DPCG_CHIP_STREAM={os.environ.get('DPCG_CHIP_STREAM','unset')} {label}: {best.iterations} updates {best.seconds*1e6/best.iterations:.2f} us/update chip_by_default={S.chip_info()['chip_by_default']}", flush=True)
    S.close()
