import sys, torch
import deeppreconditioning_amd as D
from deeppreconditioning_amd import poisson
A = poisson.unstructured_like_csr(3, 100, 0)
s = D.CsrSystem.from_any(A)
s.set_preconditioner(D.IC0("solve")); torch.cuda.synchronize()
print("---- second call", file=sys.stderr, flush=True)
s.set_preconditioner(D.IC0("solve")); torch.cuda.synchronize()
