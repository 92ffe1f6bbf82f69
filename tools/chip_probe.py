"""Whole-chip solve (dpcg_chip.hip) against the multi-launch path and the device-tree oracle; times and phase trace.
    DPCG_CHIP_TRACE=1 python tools/chip_probe.py [quick]"""
import os, sys, time, pathlib
import numpy as np
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import torch
import deeppreconditioning_amd as D
from oracle import oracle as O, c_oracle as CO

quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
dev = torch.device("cuda")
def _dev(a): return torch.as_tensor(np.ascontiguousarray(a), device=dev)

cases = [("poisson3d_100", lambda: O.poisson3d(100), 1024), ("poisson2d_1024", lambda: O.poisson2d(1024), 300),
         ("poisson2d_512", lambda: O.poisson2d(512), 400), ("poisson3d_50", lambda: O.poisson3d(50), 1024)]
if quick:
    cases = cases[:1]
for name, make, max_iter in cases:
    A = make()
    n = A.shape[0]
    S = D.CsrSystem.from_any(A, reorder=None)
    b = O.rhs(n, 0)
    for kind, pc, okw in (("jacobi", D.Jacobi(), dict(dinv=O.jacobi_dinv(A))), ("none", None, {})):
        S.set_preconditioner(pc)
        ci = S.chip_info()
        geo = S.reduction_geometry()
        t0 = time.perf_counter()
        res = S.solve(_dev(b), max_iter=max_iter)
        torch.cuda.synchronize()
        t_first = time.perf_counter() - t0
        multi = S.solve(_dev(b), max_iter=max_iter, flags=D._lib.NO_SMALL)
        print(f"{name} {kind}: chip_info={ci['chip_eligible']},{ci['chip_by_default']} per={ci['rows_per_workgroup']} band={ci['max_band']} len={ci['max_row_len']} "
              f"chip it={res.iterations} st={res.status} res={res.final_res:.6e} s={res.seconds*1e3:.3f} ms ({res.seconds/max(res.iterations,1)*1e6:.2f} us/update; first call {t_first*1e3:.1f} ms) | "
              f"multi it={multi.iterations} res={multi.final_res:.6e} s={multi.seconds*1e3:.3f} ms ({multi.seconds/max(multi.iterations,1)*1e6:.2f} us/update)", flush=True)
        hm = np.asarray(multi.res_history); hc = np.asarray(res.res_history)
        m = min(len(hm), len(hc))
        print("   max rel diff of histories (chip vs multi):", float(np.max(np.abs(hm[:m] - hc[:m]) / np.maximum(np.abs(hm[:m]), 1e-300))), flush=True)
        if not quick or kind == "jacobi":
            _, it, hist, x = CO.pcg(A, b, kind, max_iter=max_iter, device_tree={**geo, "form": "chip", "rows_per_workgroup": ci["rows_per_workgroup"]}, **okw)
            eq = res.iterations == it and np.array_equal(hc, hist)
            print("   oracle(chip tree): it", it, "history equal:", eq, "x equal:", bool(np.array_equal(res.x.cpu().numpy(), x)),
                  "" if eq else f"first diff at {int(np.argmax(hc[:len(hist)] != hist[:len(hc)]))}", flush=True)
        # repeat timing
        ts = []
        for _ in range(10):
            r2 = S.solve(_dev(b), max_iter=max_iter, want_history=False)
            ts.append(r2.seconds)
        print(f"   10 solves: median {np.median(ts)*1e3:.3f} ms = {np.median(ts)/max(r2.iterations,1)*1e6:.2f} us/update; min {min(ts)*1e3:.3f}; trace {S.chip_info()['trace_us']}", flush=True)
        again = S.solve(_dev(b), max_iter=max_iter)
        print("   reproducible:", bool(np.array_equal(again.res_history, res.res_history) and torch.equal(again.x, res.x)), flush=True)
    # x0 path
    x0 = O.rhs(n, 7)
    S.set_preconditioner(D.Jacobi())
    r = S.solve(_dev(b), _dev(x0), max_iter=25)
    m2 = S.solve(_dev(b), _dev(x0), max_iter=25, flags=D._lib.NO_SMALL)
    print("   x0: it", r.iterations, m2.iterations, "status", r.status, "max rel hist diff",
          float(np.max(np.abs(np.asarray(r.res_history) - np.asarray(m2.res_history)) / np.abs(np.asarray(m2.res_history)))), flush=True)
    S.close()
