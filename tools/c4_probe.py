"""Times the in-PCG SpMV kernel of a Poisson system through the library (spmv_dot_bench), several repeats."""
import sys
import torch
from deeppreconditioning_amd import poisson

dim, n = int(sys.argv[1]), int(sys.argv[2])
s = poisson.poisson_system(dim, n, device="cuda:0")
print("info", s.info())
for rep in (20, 50, 200):
    for _ in range(3):
        ms = s.spmv_dot_bench(rep)
        print(f"reps {rep:4d}: {ms * 1e3:8.2f} us")
