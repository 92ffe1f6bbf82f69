"""Config 2 through the plain call: M = L L^T multiplied on the whole chip (dpcg_chip_llt.hip) vs the multi-launch path."""
import sys, pathlib
import numpy as np
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import torch
import deeppreconditioning_amd as D
from oracle import oracle as O, c_oracle as CO
def _dev(a): return torch.from_numpy(np.ascontiguousarray(a)).cuda()
for name, A, L in (("poisson2d_256 learned-like (15 entries a row)", O.poisson2d(256), None), ("poisson2d_256 ic0", O.poisson2d(256), "ic0"),
                   ("poisson3d_40 ic0", O.poisson3d(40), "ic0"), ("poisson2d_400 ic0", O.poisson2d(400), "ic0")):
    n = A.shape[0]
    Lf = O.learned_like_factor_preconditioning(A) if L is None else CO.ic0(A)
    S = D.CsrSystem.from_any(A)
    S.set_preconditioner(D.LLtMultiply(Lf))
    b = _dev(O.rhs(n, 0))
    S.solve(b, want_history=False); S.solve(b, flags=D._lib.NO_SMALL, want_history=False)
    chip = [S.solve(b, want_history=False) for _ in range(7)]
    multi = [S.solve(b, flags=D._lib.NO_SMALL, want_history=False) for _ in range(7)]
    c, m = chip[-1], multi[-1]
    print(f"{name}: n={n} chip_by_default={S.chip_info()['chip_by_default']} chip it={c.iterations} {np.median([r.seconds for r in chip])/c.iterations*1e6:.2f} us/update | "
          f"multi it={m.iterations} {np.median([r.seconds for r in multi])/m.iterations*1e6:.2f} us/update", flush=True)
    S.close()
