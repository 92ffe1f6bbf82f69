"""ICT in solve mode (level-1 fill: rows longer than a record) on natural and scrambled grids: setup, levels, apply, PCG."""
import sys, time
import torch
import deeppreconditioning_amd as D
from deeppreconditioning_amd import poisson

cases = [("poisson3d_64", lambda: poisson.poisson_system(3, 64)), ("poisson2d_512", lambda: poisson.poisson_system(2, 512)),
         ("scrambled3d_64", lambda: D.CsrSystem.from_any(poisson.unstructured_like_csr(3, 64, 0))),
         ("scrambled3d_100", lambda: D.CsrSystem.from_any(poisson.unstructured_like_csr(3, 100, 0)))]
for name, make in cases:
    if sys.argv[1:] and name not in sys.argv[1:]:
        continue
    s = make()
    for thr in (0.1, 0.001):
        s.set_preconditioner(D.ICT("solve", 1, thr))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s.set_preconditioner(D.ICT("solve", 1, thr))
        torch.cuda.synchronize()
        setup_ms = (time.perf_counter() - t0) * 1e3
        info = s.info()
        r = poisson.rhs(s.n, 0)
        res = s.solve(r, want_history=False)
        res = s.solve(r, want_history=False)
        print(f"{name:16s} thr {thr:6.3f} nnz(L) {info['precond_nnz']:9d} levels {info['levels_lower']:5d}/{info['levels_upper']:5d} setup {setup_ms:8.2f} ms  "
              f"PCG {res.iterations:4d} its {res.seconds * 1e3:8.2f} ms = {res.seconds / max(res.iterations, 1) * 1e6:8.1f} us/update status {res.status}", flush=True)
    s.close()
