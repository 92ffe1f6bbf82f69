"""Per-workgroup phase trace of the whole-chip solve: DPCG_CHIP_TRACE=1 DPCG_CHIP_TRACE_PRINT=1 [DPCG_CHIP_TRACE_ALL=1] python tools/chip_probe2.py name..."""
import os, sys, time, pathlib
import numpy as np
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import torch
import deeppreconditioning_amd as D
from oracle import oracle as O
dev = torch.device("cuda")
def _dev(a): return torch.as_tensor(np.ascontiguousarray(a), device=dev)
makers = {"p3d80": lambda: O.poisson3d(80), "p2d720": lambda: O.poisson2d(720), "p3d100": lambda: O.poisson3d(100),
          "p2d1024": lambda: O.poisson2d(1024), "p3d64": lambda: O.poisson3d(64), "p2d512": lambda: O.poisson2d(512)}
for name in sys.argv[1:]:
    A = makers[name]()
    n = A.shape[0]
    S = D.CsrSystem.from_any(A, reorder=None)
    b = O.rhs(n, 0)
    for pc in (D.Jacobi(), None):
        S.set_preconditioner(pc)
        S.solve(_dev(b), max_iter=200, want_history=False)
        ts = [S.solve(_dev(b), max_iter=200, want_history=False) for _ in range(5)]
        r = ts[-1]
        print(f"{name} {'jacobi' if pc else 'none'} n={n}: it={r.iterations} median {np.median([t.seconds for t in ts])/r.iterations*1e6:.2f} us/update  info={S.chip_info()['chip_by_default']}", flush=True)
    S.close()
