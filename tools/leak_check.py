import sys, torch, gc
sys.path.insert(0, "/root/repo")
import deeppreconditioning_amd as D
from deeppreconditioning_amd import poisson
def used(): 
    torch.cuda.synchronize(); f, t = torch.cuda.mem_get_info(); return (t - f) / 1e6
base = used()
for rep in range(3):
    for i in range(40):
        S = poisson.poisson_system(3, 60 if i % 8 else 76); b = poisson.rhs(S.n, i)      # (76^3: tile plans for the factors too)
        for pc in (D.Jacobi(), D.IC0("solve"), D.IC0("multiply"), D.IC0("solve", ordering="multicolor"), None):
            S.set_preconditioner(pc); S.solve(b, max_iter=20, want_history=False)
        S.solve(b, max_iter=10, flags=D._lib.SPMV_F32 | D._lib.NO_SMALL)
        S.close(); del S, b
    gc.collect(); torch.cuda.empty_cache()
    print("after round", rep, "used MB above baseline:", round(used() - base, 1))
