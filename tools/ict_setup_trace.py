"""DPCG_SETUP_TRACE=1 python tools/ict_setup_trace.py [n]: phases of the ICT(1, 0.1) setup (the harness's default technique) at n x n."""
import sys
import torch
import deeppreconditioning_amd as D
from deeppreconditioning_amd import poisson
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for mode in ("multiply", "solve"):
    s = poisson.poisson_system(2, n)
    s.set_preconditioner(D.ICT(mode))
    torch.cuda.synchronize()
    print(f"---- ICT {mode} {n}^2, second call", file=sys.stderr, flush=True)
    s.set_preconditioner(D.ICT(mode))
    torch.cuda.synchronize()
    info = s.info()
    print(f"     nnz(L) {info['precond_nnz']} levels {info['levels_lower']}", file=sys.stderr, flush=True)
    s.close()
