"""Setup-phase breakdown (DPCG_SETUP_TRACE=1 prints the phases on stderr): IC(0) in solve mode on one BASELINE system."""
import sys
import torch
import deeppreconditioning_amd as D
from deeppreconditioning_amd import poisson

dim, size = int(sys.argv[1]), int(sys.argv[2])
s = poisson.poisson_system(dim, size)
s.set_preconditioner(D.IC0("solve"))
torch.cuda.synchronize()
print("---- second call", file=sys.stderr, flush=True)
s.set_preconditioner(D.IC0("solve"))
torch.cuda.synchronize()
