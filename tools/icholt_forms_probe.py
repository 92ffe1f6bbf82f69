"""ICholT setup by form: DPCG_ICHOLT_LDS=1 (default: the pipeline of waves, factor in LDS where it fits, else in a workspace in memory up to 8192 rows),\n2 (the workspace form everywhere), 0 (the one-wave kernel).    python tools/icholt_forms_probe.py"""
import time, torch, os
import deeppreconditioning_amd as D
from deeppreconditioning_amd import meshes, poisson
for name, make in (("poisson2d_49 (2.4K)", lambda: poisson.poisson_system(2, 49)), ("poisson2d_74 (5.5K)", lambda: poisson.poisson_system(2, 74)),
                   ("quadtree 5.3K", lambda: D.CsrSystem.from_any(meshes.quadtree_fv_laplacian(70, 3))), ("poisson2d_90 (8.1K)", lambda: poisson.poisson_system(2, 90)),
                   ("poisson3d_20 (8K)", lambda: poisson.poisson_system(3, 20))):
    S = make()
    for env in ("1", "2", "0"):
        os.environ["DPCG_ICHOLT_LDS"] = env
        S.set_preconditioner(D.ICholT("multiply", 1, 0.1)); torch.cuda.synchronize()
        t0 = time.perf_counter(); S.set_preconditioner(D.ICholT("multiply", 1, 0.1)); torch.cuda.synchronize()
        print(f"{name} n={S.n}: DPCG_ICHOLT_LDS={env}: ICholT('multiply') setup {(time.perf_counter() - t0) * 1e3:.3f} ms, nnz(L) {S.info()['precond_nnz']}", flush=True)
    S.close()
