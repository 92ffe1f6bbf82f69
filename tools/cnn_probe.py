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
