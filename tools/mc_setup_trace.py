"""Phases of the IC(0)-in-multicolour-order setup (DPCG_SETUP_TRACE=1 prints them), and the setup of the NEXT system of the same
mesh (update_values parks the factor; only values are computed again):  python tools/mc_setup_trace.py [c2|c3|natural]"""
import sys, time, torch
import deeppreconditioning_amd as D
from deeppreconditioning_amd import poisson
which = sys.argv[1] if len(sys.argv) > 1 else "c2"
if which == "c3":
    A = poisson.unstructured_like_csr(3, 100, 0)
    s = D.CsrSystem.from_any(A)
    vals = torch.from_numpy(A.data.astype("float64")).cuda()
else:
    from oracle import oracle as O
    A = O.poisson2d(256) if which == "c2" else O.poisson3d(100)
    s = D.CsrSystem.from_any(A)
    vals = torch.from_numpy(A.data.astype("float64")).cuda()
pc = lambda: D.IC0("solve", ordering="multicolor")
s.set_preconditioner(pc()); torch.cuda.synchronize()
print("---- second call (colouring kept)", file=sys.stderr, flush=True)
t0 = time.perf_counter()
s.set_preconditioner(pc()); torch.cuda.synchronize()
print(f"---- total {(time.perf_counter() - t0) * 1e3:.2f} ms", file=sys.stderr)
for rnd in range(3):      # the first refresh builds the entry maps, the later ones are the steady state of a time-stepping loop
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s.update_values(vals); torch.cuda.synchronize()
    t1 = time.perf_counter()
    s.set_preconditioner(pc()); torch.cuda.synchronize()
    print(f"---- new values {rnd}: update_values {(t1 - t0) * 1e3:.2f} ms, setup {(time.perf_counter() - t1) * 1e3:.2f} ms", file=sys.stderr)
b = poisson.rhs(s.n, 0)
r = s.solve(b, want_history=False)
print(f"solve: {r.iterations} iterations, status {r.status}", file=sys.stderr)
