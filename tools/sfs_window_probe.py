"""Window (workgroups in flight) of the CSR-stream sync-free solve: DPCG_SF_FACTOR x the widest level.   python tools/sfs_window_probe.py
(The `gate` column drove DPCG_SF_GATE, a level-gate experiment that was measured -- profiles/r04_sfs_window.txt -- and removed from
the library: the variable is ignored now.)"""
import os
import subprocess
import sys

CHILD = r'''
import torch
import deeppreconditioning_amd as D
from deeppreconditioning_amd import poisson, meshes
for name, make in (("delaunay_1M", lambda: meshes.delaunay_laplacian(1000000, 0)), ("scrambled3d_100", lambda: poisson.unstructured_like_csr(3, 100, 0))):
    A = make()
    s = D.CsrSystem.from_any(A)
    b = poisson.rhs(s.n, 0)
    s.set_preconditioner(D.IC0("solve"))
    s.solve(b, max_iter=30, want_history=False)
    r = s.solve(b, max_iter=150, want_history=False)
    print(f"{name} {r.seconds / r.iterations * 1e6:.1f} us/update", end="; ", flush=True)
    s.close()
'''
import itertools
gates = sys.argv[1].split(",") if len(sys.argv) > 1 else ["0"]
for stream, gate in itertools.product(("1", "0") if gates == ["0"] else ("1",), gates):
    for f in (("0.35", "0.5", "0.85", "1.2", "1.7", "2.5") if gates == ["0"] else ("0.85", "1.7", "2.5", "4", "6")):
        r = subprocess.run([sys.executable, "-c", CHILD], env={**os.environ, "DPCG_SF_STREAM": stream, "DPCG_SF_FACTOR": f, "DPCG_SF_GATE": gate},
                           capture_output=True, text=True)
        print(f"stream={stream} gate={gate} factor={f}: " + r.stdout.strip() + (r.stderr[-300:] if r.returncode else ""), flush=True)
