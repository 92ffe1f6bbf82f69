import sys, time, torch, numpy as np
import deeppreconditioning_amd as D
from deeppreconditioning_amd import meshes
for name, make in (("quadtree_random", lambda: meshes.quadtree_fv_laplacian(1000, 0, numbering="random")), ("delaunay", lambda: meshes.delaunay_laplacian(1000000, 0))):
    A = make()
    s = D.CsrSystem.from_any(A, reorder="rcm"); s.close()
    print("==", name, file=sys.stderr, flush=True)
    s = D.CsrSystem.from_any(A, reorder="rcm"); s.close()
