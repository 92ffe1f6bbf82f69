"""DPCG_SETUP_TRACE=1 python tools/create_trace.py: the phases of creating (and reordering) the config-3 stand-in and a scrambled 2-D grid."""
import sys, time
import torch
import deeppreconditioning_amd as D
from deeppreconditioning_amd import poisson
for dim, n in ((3, 100), (2, 1024)):
    A = poisson.unstructured_like_csr(dim, n, 0)
    S = D.CsrSystem.from_any(A); S.close()
    print(f"---- scrambled {dim}-D {n}", file=sys.stderr, flush=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    S = D.CsrSystem.from_any(A)
    torch.cuda.synchronize(); print(f"create {1e3 * (time.perf_counter() - t0):.1f} ms", file=sys.stderr, flush=True)
    S.close()
