"""IC(0) in solve mode on the BASELINE systems: setup time (the harness's `setups` column, test.py:130-135), levels, the
cost of one apply z = L^-T (L^-1 r) and of a PCG update, and the PCG itself.  DPCG_SYNCFREE=0 gives the one-launch-per-
wide-level schedule for an A/B."""
import os
import sys
import time

import torch

import deeppreconditioning_amd as D
from deeppreconditioning_amd import poisson

cases = [("poisson2d_256", lambda: poisson.poisson_system(2, 256)), ("poisson2d_1024", lambda: poisson.poisson_system(2, 1024)),
         ("poisson3d_64", lambda: poisson.poisson_system(3, 64)), ("poisson3d_100", lambda: poisson.poisson_system(3, 100)),
         ("scrambled3d_100", lambda: D.CsrSystem.from_any(poisson.unstructured_like_csr(3, 100, 0))),
         ("scrambled3d_64", lambda: D.CsrSystem.from_any(poisson.unstructured_like_csr(3, 64, 0))),
         ("scrambled3d_40", lambda: D.CsrSystem.from_any(poisson.unstructured_like_csr(3, 40, 0))),
         ("scrambled2d_1024", lambda: D.CsrSystem.from_any(poisson.unstructured_like_csr(2, 1024, 0))),
         ("scrambled2d_256", lambda: D.CsrSystem.from_any(poisson.unstructured_like_csr(2, 256, 0)))]
only = sys.argv[1:] or None
print(f"DPCG_SYNCFREE={os.environ.get('DPCG_SYNCFREE', '1')}")
for name, make in cases:
    if only and name not in only:
        continue
    s = make()
    s.set_preconditioner(D.IC0("solve"))          # first call: library warm-up (module load) not timed
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s.set_preconditioner(D.IC0("solve"))
    torch.cuda.synchronize()
    setup_ms = (time.perf_counter() - t0) * 1e3
    info = s.info()
    r = poisson.rhs(s.n, 0)
    z = s.precond_apply(r)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        z = s.precond_apply(r)
    torch.cuda.synchronize()
    apply_us = (time.perf_counter() - t0) / 20 * 1e6
    res = s.solve(r, want_history=False)
    res = s.solve(r, want_history=False)
    print(f"{name:16s} rows {s.n:8d} levels {info['levels_lower']:5d}/{info['levels_upper']:5d}  setup {setup_ms:8.2f} ms  "
          f"apply {apply_us:9.1f} us  PCG {res.iterations:4d} its {res.seconds * 1e3:8.2f} ms = "
          f"{res.seconds / max(res.iterations, 1) * 1e6:8.1f} us/update  status {res.status}  checksum {float(z.sum()):.15e}",
          flush=True)
    s.close()
