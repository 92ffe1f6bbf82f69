"""Where the spconv-free PreconditionerNet forward spends its time (256^2 5-point system, random weights)."""
import time
import numpy as np
import scipy.sparse as sp
import torch
from deeppreconditioning_amd import model as mdl

torch.manual_seed(69)
net = mdl.PreconditionerNet([1, 16, 32, 64, 32, 16, 1]).cuda()
n2 = 256
idx = np.arange(n2 * n2)
A2 = sp.diags([np.full(n2 * n2, 4.0), np.where((idx[:-1] + 1) % n2 != 0, -1.0, 0.0), np.full(n2 * n2 - n2, -1.0)],
              [0, -1, -n2], format="csr")
inp, sizes = mdl.tril_batch_from_csr([A2], device="cuda")


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, out


with torch.no_grad():
    ms, outL = timed(lambda: net(inp))
    print(f"forward {ms:.2f} ms, nnz out {outL.features.shape[0]}")
    ms2, _ = timed(lambda: mdl.lower_factor_csr(outL, 0, sizes[0]))
    print(f"lower_factor_csr {ms2:.2f} ms")
    t = inp
    for i, m in enumerate(net.layers):
        if isinstance(m, mdl.SparseConv2d):
            ms, t2 = timed(lambda: m(t))
            print(f"layer {i} conv k={m.kernel_size} pad={m.padding} cin={m.in_channels} cout={m.out_channels}: {ms:.2f} ms, nnz {t.features.shape[0]} -> {t2.features.shape[0]}")
            t = t2
        else:
            t = t.replace_feature(m(t.features))

# the rest of the learned pipeline at this size: handing L to the solver (transpose, plans) and the solve
import deeppreconditioning_amd as D
from deeppreconditioning_amd import poisson
s = poisson.poisson_system(2, n2, device="cuda:0")
Lparts = mdl.lower_factor_csr(outL, 0, sizes[0])
for mode, cls in (("multiply", D.LLtMultiply), ("solve", D.LLtSolve)):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s.set_preconditioner(cls(Lparts))
    torch.cuda.synchronize()
    print(f"set_preconditioner(LLt{mode.capitalize()}) nnz_L {Lparts[1].numel()}: {(time.perf_counter() - t0) * 1e3:.2f} ms")
    t0 = time.perf_counter()
    s.set_preconditioner(cls(Lparts))
    torch.cuda.synchronize()
    print(f"   again: {(time.perf_counter() - t0) * 1e3:.2f} ms")
