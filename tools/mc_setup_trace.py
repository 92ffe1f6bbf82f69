"""Phases of the IC(0)-in-multicolour-order setup (DPCG_SETUP_TRACE=1 prints them):  python tools/mc_setup_trace.py [c2|c3|natural]"""
import sys, time, torch
import deeppreconditioning_amd as D
from deeppreconditioning_amd import poisson
which = sys.argv[1] if len(sys.argv) > 1 else "c2"
s = (poisson.poisson_system(2, 256) if which == "c2" else poisson.poisson_system(3, 100) if which == "natural"
     else D.CsrSystem.from_any(poisson.unstructured_like_csr(3, 100, 0)))
s.set_preconditioner(D.IC0("solve", ordering="multicolor")); torch.cuda.synchronize()
print("---- second call", file=sys.stderr, flush=True)
t0 = time.perf_counter()
s.set_preconditioner(D.IC0("solve", ordering="multicolor")); torch.cuda.synchronize()
print(f"---- total {(time.perf_counter() - t0) * 1e3:.2f} ms (with the trace's synchronisations)", file=sys.stderr)
