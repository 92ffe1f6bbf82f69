"""One full-size parity point for BASELINE config 4: a 256^3 system (16.8M DoF, 117M non-zeros) solved by the HIP path and
by the C oracle (oracle/pcg_oracle.c, OpenMP) on the box's host cores -- iteration count and residual history compared at
north_star's 1e-10.  Takes a minute or two of CPU time, hence a tool (its output is kept under profiles/), not a unit test."""
import os
import sys
import time

import numpy as np
import torch

os.environ.setdefault("OMP_WAIT_POLICY", "passive")
import deeppreconditioning_amd as D  # noqa: E402
from deeppreconditioning_amd import poisson  # noqa: E402
from oracle import c_oracle as CO  # noqa: E402
from oracle import oracle as O  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 3
S = poisson.poisson_system(3, 256)
S.set_preconditioner(D.Jacobi())
b = poisson.rhs(S.n, seed)
res = S.solve(b)
print(f"HIP: seed {seed}: {res.iterations} iterations, final res {res.final_res:.17g}, {res.seconds * 1e3:.1f} ms", flush=True)
A = O.poisson3d(256)
bh = O.rhs(A.shape[0], seed)
CO.set_num_threads(min(32, CO.num_threads()))
t0 = time.perf_counter()
_, it, hist, x = CO.pcg(A, bh, "jacobi", dinv=O.jacobi_dinv(A))
print(f"oracle: {it} iterations, final res {hist[-1]:.17g}, {time.perf_counter() - t0:.1f} s on {min(32, CO.num_threads())} threads", flush=True)
m = min(len(hist), len(res.res_history))
rel = np.abs(res.res_history[:m] - hist[:m]) / hist[:m]
print(f"iterations equal: {res.iterations == it}; max relative history difference {rel.max():.3e}; "
      f"max |x - x_oracle| / max|x| = {np.abs(res.x.cpu().numpy() - x).max() / np.abs(x).max():.3e}")
