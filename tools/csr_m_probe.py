"""The reference's own call -- M handed over as an explicit sparse matrix (test.py:88,105,138) -- at config-2 size: PCG with
M = L L^T (IC(0) factor and a 15-per-row factor like the CNN's) as ONE CSR matrix, per launch form."""
import numpy as np, scipy.sparse as sp, torch
import deeppreconditioning_amd as D
from deeppreconditioning_amd import poisson
from oracle import c_oracle as CO, oracle as O

def cnn_like_factor(n2):      # the sparsity of the CNN-emitted factor (15 entries per row at 256^2), seeded random weights
    from deeppreconditioning_amd import model as mdl
    torch.manual_seed(69)
    net = mdl.PreconditionerNet([1, 16, 32, 64, 32, 16, 1]).cuda()
    idx = np.arange(n2 * n2)
    A2 = sp.diags([np.full(n2 * n2, 4.0), np.where((idx[:-1] + 1) % n2 != 0, -1.0, 0.0), np.full(n2 * n2 - n2, -1.0)], [0, -1, -n2], format="csr")
    inp, sizes = mdl.tril_batch_from_csr([A2], device="cuda")
    with torch.no_grad():
        out = net(inp)
    rp, ci, v = [t.cpu().numpy() for t in mdl.lower_factor_csr(out, 0, sizes[0])]
    return sp.csr_matrix((v, ci, rp), shape=(n2 * n2, n2 * n2))


for name, A in (("poisson2d_256", O.poisson2d(256)), ("poisson3d_40", O.poisson3d(40)), ("poisson2d_256_cnn", O.poisson2d(256)),
                ("poisson2d_128_cnn", O.poisson2d(128))):
    n = A.shape[0]
    b = O.rhs(n, 0)
    L = cnn_like_factor(int(round(n ** 0.5))) if name.endswith("cnn") else CO.ic0(A)
    M = (L @ L.T).tocsr(); M.sort_indices()
    S = D.CsrSystem.from_any(A)
    S.set_preconditioner(D.CsrPreconditioner(M))
    _, it, hist, xs = CO.pcg(A, b, "csr", M=M)
    for flags, tag in ((0, "default"), (D._lib.NO_SMALL, "no_small"), (D._lib.NO_SMALL | D._lib.NO_FUSE, "three-kernel")):
        r = S.solve(torch.from_numpy(b).cuda(), flags=flags); r = S.solve(torch.from_numpy(b).cuda(), flags=flags)
        m = min(len(hist), len(r.res_history), 8)
        print(f"{name:18s} {S.info()['spmv_kernel']:6s} nnz(M)/row {M.nnz / n:5.1f} {tag:13s} its {r.iterations:4d} (oracle {it}) {r.seconds / max(r.iterations, 1) * 1e6:7.2f} us/update  "
              f"head rel err {float(np.max(np.abs(r.res_history[:m] - hist[:m]) / hist[:m])):.1e}", flush=True)
    S.close()
