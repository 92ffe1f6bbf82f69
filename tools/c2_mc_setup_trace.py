import sys, time, torch
import deeppreconditioning_amd as D
from deeppreconditioning_amd import poisson
for r in range(3):
    s = poisson.poisson_system(2, 256)
    torch.cuda.synchronize()
    print(f"== round {r}", file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    s.set_preconditioner(D.IC0("solve", ordering="multicolor"))
    torch.cuda.synchronize()
    print(f"== round {r}: {1e3 * (time.perf_counter() - t0):.2f} ms", file=sys.stderr, flush=True)
    s.close()
