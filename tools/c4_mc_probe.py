"""IC(0) in multicolour order at config 4's size (256^3, 16.8M rows: every stream from HBM): setup, iterations, time to the
solution against Jacobi, and the true residual of the result.   python tools/c4_mc_probe.py [n]"""
import sys, time, torch
import deeppreconditioning_amd as D
from deeppreconditioning_amd import poisson
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
s = poisson.poisson_system(3, n)
b = poisson.rhs(s.n, 0)
for label, pc in (("jacobi", D.Jacobi()), ("ic0 multicolour", D.IC0("solve", ordering="multicolor"))):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    s.set_preconditioner(pc); torch.cuda.synchronize()
    setup = (time.perf_counter() - t0) * 1e3
    s.solve(b, want_history=False)
    r = s.solve(b, want_history=False)
    res = b - s @ r.x
    print(f"{n}^3 {label:16s} setup {setup:8.2f} ms  {r.iterations:4d} its  {r.seconds * 1e3:9.3f} ms = {r.seconds / r.iterations * 1e6:8.1f} us/update  "
          f"status {r.status}  true |r|^2/|b|^2 {float(D.dot(res, res) / D.dot(b, b)):.3e}  levels {s.info()['levels_lower']}", flush=True)
