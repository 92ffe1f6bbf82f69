"""Workload for the rocprofv3 PMC passes (run as `rocprofv3 --pmc <C> --kernel-trace ... -- python3 tools/pmc_run.py`).

Launches the SpMV (+<p,Ap>) kernel of the PCG loop on the 1M-DoF and the 256^3 systems plus kernels of
KNOWN byte counts at each access width (8-B loads: k_dot_partials; 16-B loads/stores: k_update_r /
k_update_xp inside a short PCG on the 256^3 system) that calibrate FETCH_SIZE / WRITE_SIZE on gfx950.
"""
import pathlib
import sys

import torch

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import deeppreconditioning_amd as D  # noqa: E402
from deeppreconditioning_amd import poisson  # noqa: E402

torch.cuda.set_device(0)
for dim, n in ((3, 100), (2, 1024)):
    s = poisson.poisson_system(dim, n)
    s.set_preconditioner(D.Jacobi())
    s.spmv_dot_bench(20)
    s.solve(poisson.rhs(s.n, 0), max_iter=32, want_history=False)
    del s
s = poisson.poisson_system(3, 256)
s.set_preconditioner(D.Jacobi())
s.spmv_dot_bench(10)
b = poisson.rhs(s.n, 0)
for _ in range(5):
    D.dot(b, b)  # 8-byte loads, 2 x 134 MB per launch
s.solve(b, max_iter=16, want_history=False)
del s
# 4-byte loads of known volume: creating a system from fp32 values converts them (k_convert<float,double>:
# reads 4 B x nnz = 468 MB, writes 8 B x nnz = 936 MB)
rp, ci, v32 = poisson.poisson_csr(3, 256, dtype=torch.float32)
s32 = D.CsrSystem(rp, ci, v32, rp.numel() - 1)
torch.cuda.synchronize()
del s32, rp, ci, v32
# BASELINE config 3 stand-in (scrambled numbering of the 100^3 system): the gather SpMV on the caller's numbering, then the
# same system as the library iterates on it after dpcg_reorder.  A k_gen_poisson dispatch marks each segment.
A = poisson.unstructured_like_csr(3, 100, 0)
for mode in (None, "auto"):
    poisson.poisson_csr(2, 8)                      # segment marker
    s = D.CsrSystem.from_any(A, reorder=mode)
    s.set_preconditioner(D.Jacobi())
    s.spmv_dot_bench(20)
    s.close()
torch.cuda.synchronize()
