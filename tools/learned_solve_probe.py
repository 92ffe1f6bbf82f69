"""The CNN-emitted factor (15 entries per row at 256^2; seeded random weights) applied by TRIANGULAR SOLVES (north_star's "true IC
apply" of a learned L): levels, apply, PCG."""
import time
import numpy as np, scipy.sparse as sp, torch
import deeppreconditioning_amd as D
from deeppreconditioning_amd import model as mdl, poisson

for n2 in (128, 256):
    torch.manual_seed(69)
    net = mdl.PreconditionerNet([1, 16, 32, 64, 32, 16, 1]).cuda()
    idx = np.arange(n2 * n2)
    A2 = sp.diags([np.full(n2 * n2, 4.0), np.where((idx[:-1] + 1) % n2 != 0, -1.0, 0.0), np.full(n2 * n2 - n2, -1.0)], [0, -1, -n2], format="csr")
    inp, sizes = mdl.tril_batch_from_csr([A2], device="cuda")
    with torch.no_grad():
        out = net(inp)
    rp, ci, v = [t.cpu().numpy() for t in mdl.lower_factor_csr(out, 0, sizes[0])]
    L = sp.csr_matrix((v, ci, rp), shape=(n2 * n2, n2 * n2))
    s = poisson.poisson_system(2, n2)
    b = poisson.rhs(s.n, 0)
    t0 = time.perf_counter(); s.set_preconditioner(D.LLtSolve(L)); torch.cuda.synchronize(); setup = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter(); s.set_preconditioner(D.LLtSolve(L)); torch.cuda.synchronize(); setup = (time.perf_counter() - t0) * 1e3
    r = s.solve(b, want_history=False); r = s.solve(b, want_history=False)
    info = s.info()
    print(f"{n2}^2: nnz(L)/row {L.nnz / L.shape[0]:.1f} levels {info['levels_lower']}/{info['levels_upper']} setup {setup:.1f} ms  PCG {r.iterations} its "
          f"{r.seconds / max(r.iterations, 1) * 1e6:.1f} us/update status {r.status}", flush=True)
    s.close()
