"""Where a NEW-pattern multicolour IC(0) setup goes at 256^3 with a warm allocator (a second handle of the same matrix):
DPCG_SETUP_TRACE=1 python tools/c4_mc_setup_trace.py [n]"""
import sys
import time

import torch

import deeppreconditioning_amd as D
from deeppreconditioning_amd import poisson

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for round_ in range(3):
    s = poisson.poisson_system(3, n)
    torch.cuda.synchronize()
    print(f"== round {round_}", file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    s.set_preconditioner(D.IC0("solve", ordering="multicolor"))
    torch.cuda.synchronize()
    print(f"== round {round_}: {1e3 * (time.perf_counter() - t0):.2f} ms", file=sys.stderr, flush=True)
    s.close()
