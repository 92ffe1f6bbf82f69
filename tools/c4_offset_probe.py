"""256^3 x-tile SpMV with the matrix VALUE array shifted by a chosen byte offset inside one big buffer (everything else stays
where it is: the other arrays are borrowed tensors / reused cached blocks): does the 274-311 us spread between runs come from the
relative placement of the streams?"""
import torch
import deeppreconditioning_amd as D
from deeppreconditioning_amd import poisson

rp, ci, val0 = poisson.poisson_csr(3, 256)
nnz = val0.numel()
big = torch.empty(nnz + (64 << 20) // 8, dtype=torch.float64, device="cuda")
for off in (0, 256, 1024, 4096, 16384, 65536, 1 << 18, 1 << 20, 2 << 20, 3 << 20, 5 << 20, 8 << 20, 16 << 20, 33 << 20):
    v = big[off // 8: off // 8 + nnz]
    v.copy_(val0)
    s = D.CsrSystem(rp, ci, v, rp.numel() - 1)
    s.set_preconditioner(D.Jacobi())
    ms = sorted(s.spmv_dot_bench(40) for _ in range(3))
    print(f"val offset {off:9d} B (addr mod 2 MiB = {v.data_ptr() % (2 << 20):8d}): {ms[0] * 1e3:7.1f} {ms[1] * 1e3:7.1f} {ms[2] * 1e3:7.1f} us", flush=True)
    s.close()
    del s
