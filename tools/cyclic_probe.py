"""x-tile SpMV: row blocks in slabs vs dealt out cyclically (DPCG_SPMV_CYCLIC), by workgroups per CU, on the HBM-bound 256^3 system
and on the cache-resident 1M-DoF systems.  One child process per setting (the knobs are read once).   python tools/cyclic_probe.py"""
import os
import subprocess
import sys

CHILD = r'''
import sys, torch
import deeppreconditioning_amd as D
from deeppreconditioning_amd import poisson
from deeppreconditioning_amd.operators import release_cached_memory
what = sys.argv[1]
if what == "c4":
    out = []
    for _ in range(3):
        s = poisson.poisson_system(3, 256); s.set_preconditioner(D.Jacobi())
        out.append(min(s.spmv_dot_bench(40) for _ in range(2)) * 1e3)
        r = s.solve(poisson.rhs(s.n, 0), max_iter=48, want_history=False)
        upd = r.seconds / r.iterations * 1e6
        s.close(); del s; release_cached_memory(); torch.cuda.empty_cache()
    print("256^3 SpMV us by placement:", " ".join(f"{v:.1f}" for v in out), f"| PCG update {upd:.1f} us")
else:
    for dim, n in ((3, 100), (2, 1024), (3, 128)):
        s = poisson.poisson_system(dim, n); s.set_preconditioner(D.Jacobi())
        us = min(s.spmv_dot_bench(200) for _ in range(3)) * 1e3
        b = poisson.rhs(s.n, 0); s.solve(b, want_history=False); r = s.solve(b, want_history=False)
        print(f"{dim}d {n}: SpMV {us:.2f} us, Jacobi {r.iterations / r.seconds:.0f} it/s ({r.iterations})")
'''
quick = len(sys.argv) > 1 and sys.argv[1] == "per_cu"
xcd = len(sys.argv) > 1 and sys.argv[1] == "xcd"
todo = ([("c4", [{"DPCG_SPMV_CYCLIC": c, "DPCG_SPMV_WG_PER_CU": w} for c in ("1", "2") for w in ("3", "4")])] if xcd else
        [("c4", [{"DPCG_SPMV_CYCLIC": "1", "DPCG_SPMV_WG_PER_CU": w} for w in ("2", "3", "4", "5")])] if quick else
        [("c4", [{"DPCG_SPMV_CYCLIC": c, "DPCG_SPMV_WG_PER_CU": w} for c in "01" for w in ("4", "6", "8")]),
         ("c3", [{"DPCG_SPMV_CYCLIC": c} for c in "01"])])
for what, settings in todo:
    for env in settings:
        r = subprocess.run([sys.executable, "-c", CHILD, what], env={**os.environ, **env}, capture_output=True, text=True)
        print(env, "->", r.stdout.strip().replace("\n", " ; "), r.stderr.strip()[-300:] if r.returncode else "", flush=True)
