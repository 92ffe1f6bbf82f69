"""Does the persistent SpMV grid lose time to the remainder of row blocks per workgroup?  In-loop SpMV time for 2-D systems
whose number of 256-row blocks sits just below / at / just above multiples of the 1536-workgroup grid."""
import torch

import deeppreconditioning_amd as D
from deeppreconditioning_amd import poisson

for n in (886, 887, 950, 1000, 1024, 1085, 1086, 1087, 1150, 1254, 1255):
    s = poisson.poisson_system(2, n)
    s.set_preconditioner(D.Jacobi())
    ms = s.spmv_dot_bench(200)
    nrb = (s.n + 255) // 256
    info = s.info()
    print(f"n={n:5d} rows {s.n:8d} blocks {nrb:5d} = {nrb / 1536:5.2f} per workgroup  kernel {info['spmv_kernel']:6s} {ms * 1e3:7.2f} us  "
          f"{ms * 1e6 / nrb:6.2f} ns/block", flush=True)
    s.close()
