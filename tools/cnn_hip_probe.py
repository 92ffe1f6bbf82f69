"""PreconditionerNet forward at BASELINE config 2's size (256^2 5-point system, seeded random weights): the HIP path
(dpcg_convnet_*: plan per pattern + matrix-core forward) against the torch-ops path, and the whole learned pipeline
forward -> LLtMultiply setup -> PCG solve.   python tools/cnn_hip_probe.py [n]"""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp
import torch

import deeppreconditioning_amd as D
from deeppreconditioning_amd import model as mdl
from deeppreconditioning_amd import poisson

n2 = int(sys.argv[1]) if len(sys.argv) > 1 else 256
torch.manual_seed(69)
net = mdl.PreconditionerNet([1, 16, 32, 64, 32, 16, 1]).cuda()
idx = np.arange(n2 * n2)
A2 = sp.diags([np.full(n2 * n2, 4.0), np.where((idx[:-1] + 1) % n2 != 0, -1.0, 0.0), np.full(n2 * n2 - n2, -1.0)],
              [0, -1, -n2], format="csr")
inp, sizes = mdl.tril_batch_from_csr([A2], device="cuda")


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, out


with torch.no_grad():
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = net(inp)
    torch.cuda.synchronize()
    print(f"HIP forward, first call (module load + plan + forward): {(time.perf_counter() - t0) * 1e3:.2f} ms, sites {out.features.shape[0]}")
    ms, out = timed(lambda: net(inp))
    print(f"HIP forward, plan cached: {ms:.3f} ms")

    from deeppreconditioning_amd.utils import SparseBatch

    def fresh():     # a new indices tensor = a new sparsity pattern as far as the plan cache knows: the plan is rebuilt
        return net(SparseBatch(inp.features, inp.indices.clone(), inp.spatial_shape, inp.batch_size))
    ms, _ = timed(fresh)
    print(f"HIP forward incl. a new plan every call (a new sparsity pattern per matrix): {ms:.3f} ms")
    ms, Lparts = timed(lambda: mdl.lower_factor_csr(out, 0, sizes[0]))
    print(f"lower_factor_csr (slices of the emitted CSR): {ms:.3f} ms, nnz_L {Lparts[1].numel()}")
    os.environ["DPCG_CNN_TORCH"] = "1"
    ms, ref = timed(lambda: net(inp), reps=3)
    print(f"torch-ops forward: {ms:.2f} ms; max |HIP - torch| = {float((out.features - ref.features).abs().max()):.2e}")
    ms, _ = timed(lambda: mdl.lower_factor_csr(ref, 0, sizes[0]), reps=3)
    print(f"lower_factor_csr of the torch output (mask + sort): {ms:.2f} ms")
    del os.environ["DPCG_CNN_TORCH"]

s = poisson.poisson_system(2, n2)
b = poisson.rhs(s.n, 0)
with torch.no_grad():
    for rep in range(3):
        new_inp = SparseBatch(inp.features, inp.indices.clone(), inp.spatial_shape, inp.batch_size)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        o = net(new_inp)
        Lp = mdl.lower_factor_csr(o, 0, sizes[0])
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        s.set_preconditioner(D.LLtMultiply(Lp))
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        r = s.solve(b, want_history=False)
        t3 = time.perf_counter()
        print(f"end to end #{rep}: forward + CSR {1e3 * (t1 - t0):.2f} ms, LLtMultiply setup {1e3 * (t2 - t1):.2f} ms, "
              f"solve {1e3 * (t3 - t2):.2f} ms ({r.iterations} iterations, loop {r.seconds * 1e3:.2f} ms), total {1e3 * (t3 - t0):.2f} ms")
