import time, torch
import deeppreconditioning_amd as D
from deeppreconditioning_amd import poisson
for name, mk in (("scrambled3d_100", lambda: D.CsrSystem.from_any(poisson.unstructured_like_csr(3, 100, 0))), ("scrambled3d_64", lambda: D.CsrSystem.from_any(poisson.unstructured_like_csr(3, 64, 0)))):
    s = mk(); s.set_preconditioner(D.IC0("solve"))
    r = poisson.rhs(s.n, 0)
    for up in (False, True):
        s.sptrsv(r, upper=up); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): s.sptrsv(r, upper=up)
        torch.cuda.synchronize(); print(name, "upper" if up else "lower", f"{(time.perf_counter() - t0) / 50 * 1e6:.1f} us per standalone solve (incl. way-in / way-out passes)", flush=True)
    z = s.precond_apply(r); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): s.precond_apply(r)
    torch.cuda.synchronize(); print(name, f"apply {(time.perf_counter() - t0) / 50 * 1e6:.1f} us", flush=True)
    s.close()
