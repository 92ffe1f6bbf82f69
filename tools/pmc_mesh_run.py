"""Workload for the PMC passes over the config-3 mesh systems (rocprofv3 --pmc <C> --kernel-trace -- python3 tools/pmc_mesh_run.py):
per system a marker dispatch (k_gen_poisson), then the in-loop SpMV (+<p,Ap>) kernel 20 times on the handle as the plain call builds
it (reorder="auto": region by region for the quadtree mesh in OpenFOAM's numbering, reverse Cuthill-McKee for the scattered ones) and,
for the OpenFOAM numbering, also in RCM order and as it comes."""
import pathlib
import sys

import torch

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import deeppreconditioning_amd as D  # noqa: E402
from deeppreconditioning_amd import meshes, poisson  # noqa: E402

torch.cuda.set_device(0)
CASES = [("quadtree_foam", lambda: meshes.quadtree_fv_laplacian(1000, 0), "auto"),
         ("quadtree_foam_rcm", lambda: meshes.quadtree_fv_laplacian(1000, 0), "rcm"),
         ("quadtree_random", lambda: meshes.quadtree_fv_laplacian(1000, 0, numbering="random"), "auto"),
         ("quadtree_random_gather", lambda: meshes.quadtree_fv_laplacian(1000, 0, numbering="random"), None),
         ("delaunay", lambda: meshes.delaunay_laplacian(1000000, 0), "auto"),
         ("quadtree_foam_as_is", lambda: meshes.quadtree_fv_laplacian(1000, 0), None)]      # (round 5: "auto" numbers it region by region)
if __name__ == "__main__":
    for name, make, mode in CASES:
        A = make()
        poisson.poisson_csr(2, 8)                      # segment marker
        s = D.CsrSystem.from_any(A, reorder=mode)
        s.set_preconditioner(D.Jacobi())
        s.spmv_dot_bench(20)
        print(name, A.shape[0], A.nnz, s.info()["spmv_kernel"], s.reordered, flush=True)
        s.close()
    torch.cuda.synchronize()
