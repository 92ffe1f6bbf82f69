"""ICholT / IC(0) setup at the reference's sizes, phase by phase (DPCG_SETUP_TRACE=1), and -- DPCG_ICHOLT_TRACE=1 -- the cycles a wave of
k_icholt_lds spends per column in each phase.    DPCG_SETUP_TRACE=1 [DPCG_ICHOLT_TRACE=1] [DPCG_ICHOLT_WAVES=4|8|16] python tools/icholt_trace.py"""
import time, torch
import deeppreconditioning_amd as D
from deeppreconditioning_amd import meshes, poisson
for name, make in (("poisson2d_49", lambda: poisson.poisson_system(2, 49)), ("quadtree 2.2K", lambda: D.CsrSystem.from_any(meshes.quadtree_fv_laplacian(45, 3)))):
    S = make()
    for label, pc in (("icholt", lambda: D.ICholT("solve", 1, 0.1)), ("ic0", lambda: D.IC0("solve"))):
        S.set_preconditioner(pc()); torch.cuda.synchronize()
        print("==", name, label, flush=True)
        t0 = time.perf_counter(); S.set_preconditioner(pc()); torch.cuda.synchronize()
        print("   total %.3f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
    S.close()
