"""256^3 x-tile SpMV, back to back, on systems created several times in one process (and with the device arrays shifted by
a few KiB through dummy allocations in between): how much of the 274-311 us spread between runs is placement?"""
import sys
import torch
import deeppreconditioning_amd as D
from deeppreconditioning_amd import poisson
from deeppreconditioning_amd.operators import release_cached_memory

pads = [0, 0, 4096, 65536, 1 << 20, 3 << 20, 0]
keep = []
for i, pad in enumerate(pads):
    if pad:
        keep.append(torch.empty(pad, dtype=torch.uint8, device="cuda"))
    s = poisson.poisson_system(3, 256)
    s.set_preconditioner(D.Jacobi())
    ms = [s.spmv_dot_bench(40) for _ in range(3)]
    print(f"create {i} pad {pad:8d}: " + "  ".join(f"{m * 1e3:7.1f} us" for m in ms), flush=True)
    s.close()
    del s
    if len(sys.argv) > 1:
        release_cached_memory()
        torch.cuda.empty_cache()
