"""Generate tests/golden/*.npz by running the REFERENCE itself (imported from /root/reference).

Run once in the build container (the reference never travels to the GPU box):

    python tests/golden/make_golden.py            # all cases  (~3 min on 8 cores)
    python tests/golden/make_golden.py --quick    # skip the >= 1M-DoF cases

Only outputs are stored (iteration counts, residual histories, small result vectors); inputs are
re-created by the deterministic generators in oracle/oracle.py (closed-form Poisson matrices,
`np.random.default_rng(seed)` right-hand sides).  The residual history is captured by wrapping the
module-level `cg.stopping_criterion`, which the reference loop looks up by name every iteration
(cg.py:66,86), so the reference source is executed unmodified.
"""

from __future__ import annotations

import argparse
import pathlib
import sys

import numpy as np
import scipy.sparse as sp
import torch

HERE = pathlib.Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, "/root/reference")

from oracle import c_oracle as CO  # noqa: E402
from oracle import oracle as O  # noqa: E402
from uibk.deep_preconditioning import cg as ref_cg  # noqa: E402
from uibk.deep_preconditioning import utils as ref_utils  # noqa: E402


def to_torch_csr(A: sp.csr_matrix) -> torch.Tensor:
    A = A.tocsr()
    return torch.sparse_csr_tensor(
        torch.from_numpy(A.indptr.astype(np.int64)), torch.from_numpy(A.indices.astype(np.int64)),
        torch.from_numpy(A.data.astype(np.float64)), size=A.shape, dtype=torch.float64)


class _HistoryTap:
    """Records every value the reference's stopping criterion returns."""

    def __init__(self):
        self.values = []
        self._orig = ref_cg.stopping_criterion

    def __enter__(self):
        def tapped(A, rk, b):
            v = self._orig(A, rk, b)
            self.values.append(float(v))
            return v

        ref_cg.stopping_criterion = tapped
        return self

    def __exit__(self, *exc):
        ref_cg.stopping_criterion = self._orig


class TriSolveOperator:
    """Duck-typed `M` (cg.py only needs `M @ r`): z = L^-T L^-1 r via sequential substitution."""

    def __init__(self, L: sp.csr_matrix):
        self.L = L
        self.U = CO.transpose_csr(L)

    def __matmul__(self, r: torch.Tensor) -> torch.Tensor:
        y = CO.sptrsv_lower(self.L, r.numpy())
        return torch.from_numpy(CO.sptrsv_upper(self.U, y))


class _StubSparseConvTensor:
    """The four attributes utils.sparse_matvec_mul touches (utils.py:26-35); spconv is absent."""

    def __init__(self, features, indices, batch_size):
        self.features, self.indices, self.batch_size = features, indices, batch_size

    def replace_feature(self, f):
        return _StubSparseConvTensor(f, self.indices, self.batch_size)


def run_ref_pcg(A, b, M, x0=None, max_iter=1024):
    with _HistoryTap() as tap:
        _, iters, info = ref_cg.preconditioned_conjugate_gradient(
            A, torch.from_numpy(b), M, x0=None if x0 is None else torch.from_numpy(x0), max_iter=max_iter)
    assert len(tap.values) == iters + 1 and info == 0
    return iters, np.array(tap.values)


def ground_truth_cases(out: dict) -> None:
    """generate_data.py:107, the live call site of scipy's cg: `scipy.sparse.linalg.cg(matrix, rhs, rtol=0, atol=1e-6)`
    on a COO matrix (generate_data.py:78 builds one).  `generate_data.py` itself cannot be imported (dvc, triangle, stl
    are absent), so the call is made verbatim here; a callback only counts the iterations."""
    import scipy.sparse.linalg
    cases = {"unstructured2d_49_seed1": O.unstructured_like(O.poisson2d(49), seed=1), "poisson3d_20": O.poisson3d(20)}
    for name, A in cases.items():
        matrix = sp.coo_matrix(A)
        right_hand_side = O.rhs(A.shape[0], 69)
        count = [0]
        solution, info = scipy.sparse.linalg.cg(matrix, right_hand_side, rtol=0, atol=1e-6,
                                                callback=lambda _: count.__setitem__(0, count[0] + 1))
        out[f"ground_truth/{name}/iters_info"] = np.array([count[0], info], dtype=np.int64)
        out[f"ground_truth/{name}/x"] = np.asarray(solution, dtype=np.float64)
        r = right_hand_side - A @ solution
        print(f"ground truth {name}: {count[0]} iterations, info {info}, ||r|| = {np.linalg.norm(r):.3e}", flush=True)


def sparse_loss_cases(out: dict) -> None:
    """metrics.py:34-55 (`inverse_loss`) on a batch with the sparsity of real inputs: tril of 2-D Poisson matrices (fp32)
    and lower factors on the same pattern; the reference densifies, so only `.dense()` of the batch is needed."""
    from uibk.deep_preconditioning import metrics as ref_metrics

    class DenseStub:
        def __init__(self, dense4):
            self._d = dense4

        def dense(self):
            return self._d.clone()

    gen = torch.Generator().manual_seed(321)
    mats = [O.poisson2d(5), O.poisson2d(6)]
    dof = max(m.shape[0] for m in mats)
    sys_low = torch.zeros(len(mats), 1, dof, dof)
    pre_low = torch.zeros(len(mats), 1, dof, dof)
    for bi, m in enumerate(mats):
        n = m.shape[0]
        dense = torch.from_numpy(sp.tril(m).toarray()).float()
        sys_low[bi, 0, :n, :n] = dense
        pattern = (dense != 0).float()
        Lr = pattern * (torch.rand(n, n, generator=gen) * 0.3 - 0.15)
        Lr = torch.tril(Lr, -1) + torch.diag(0.4 + torch.rand(n, generator=gen))
        pre_low[bi, 0, :n, :n] = Lr
        for k in range(n, dof):                      # identity padding, as the data sets pad (data_set.py:94-97)
            sys_low[bi, 0, k, k] = 1.0
            pre_low[bi, 0, k, k] = 1.0
    out["metrics_sparse/systems_tril"] = sys_low.numpy().copy()
    out["metrics_sparse/preconditioners_tril"] = pre_low.numpy().copy()
    out["metrics_sparse/inverse_loss"] = np.float64(ref_metrics.inverse_loss(DenseStub(sys_low), DenseStub(pre_low)))
    print("metrics_sparse/inverse_loss:", float(out["metrics_sparse/inverse_loss"]), flush=True)


class MixedTorchOperator:
    """BASELINE config 5 as a duck-typed `A` for the REFERENCE's loop (cg.py only needs `A @ v`): matrix values and the
    vector stored in fp32, products and row sums in fp64 -- y = fp64(fp32(A)) @ fp64(fp32(v)) by torch's own sparse-CSR
    matvec.  With x0 = None every other use of A in cg.py (lines 60, 67, 87) multiplies zeros."""

    def __init__(self, A: sp.csr_matrix):
        A = A.tocsr()
        self.A32 = to_torch_csr(sp.csr_matrix((A.data.astype(np.float32).astype(np.float64), A.indices, A.indptr),
                                              shape=A.shape))

    def __matmul__(self, v: torch.Tensor) -> torch.Tensor:
        return self.A32 @ v.to(torch.float32).to(torch.float64)


def round3_cases(out: dict) -> None:
    """Round 3: (1) mixed-precision PCG (config 5) through the reference's own loop on systems whose values are NOT
    fp32-representable (D A D scaling); (2) BASELINE config 2 with a well-conditioned learned-like factor at 256^2,
    M = L L^T materialised as CSR and multiplied (test.py:100-105), so that the count is reproducible."""
    def put(name, iters, hist):
        out[f"{name}/iters"] = np.int64(iters)
        out[f"{name}/hist"] = np.asarray(hist, dtype=np.float64)
        print(f"{name}: iters={iters} res[0]={hist[0]:.17g} res[-2]={hist[-2]:.17g} res[-1]={hist[-1]:.17g}", flush=True)

    for name, A in (("unstructured3d_16", O.unstructured_like(O.poisson3d(16), seed=0)),
                    ("unstructured2d_64_seed2", O.unstructured_like(O.poisson2d(64), seed=2))):
        b = O.rhs(A.shape[0], 0)
        assert np.any(A.data.astype(np.float32).astype(np.float64) != A.data)          # lossy in fp32
        iters, hist = run_ref_pcg(MixedTorchOperator(A), b, to_torch_csr(sp.diags(O.jacobi_dinv(A)).tocsr()))
        put(f"mixed/pcg_{name}_jacobi", iters, hist)
        iters, hist = run_ref_pcg(MixedTorchOperator(A), b, TriSolveOperator(CO.ic0(A)))
        put(f"mixed/pcg_{name}_ic0_solve", iters, hist)
    A = O.poisson2d(256)
    b = O.rhs(A.shape[0], 0)
    Lw = O.learned_like_factor_preconditioning(A, seed=1, noise=0.005)   # 15 entries per row, the CNN's output pattern
    iters, hist = run_ref_pcg(to_torch_csr(A), b, to_torch_csr((Lw @ Lw.T).tocsr()))
    put("pcg_poisson2d_256_learnedlike_preconditioning_multiply", iters, hist)
    out.pop("pcg_poisson2d_256_learnedlike_wellcond_multiply/iters", None)   # first try of this round: still chaotic
    out.pop("pcg_poisson2d_256_learnedlike_wellcond_multiply/hist", None)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--add-round3", action="store_true",
                    help="only add the mixed-precision and config-2 well-conditioned fixtures to reference_outputs.npz")
    ap.add_argument("--add-round2", action="store_true",
                    help="only add the ground-truth-solve and sparse-loss fixtures to the existing reference_outputs.npz")
    args = ap.parse_args()
    out: dict[str, np.ndarray] = {}
    if args.add_round3:
        with np.load(HERE / "reference_outputs.npz") as old:
            out = {k: old[k] for k in old.files}
        round3_cases(out)
        np.savez_compressed(HERE / "reference_outputs.npz", **out)
        print("updated", HERE / "reference_outputs.npz", (HERE / "reference_outputs.npz").stat().st_size, "bytes")
        return
    if args.add_round2:
        with np.load(HERE / "reference_outputs.npz") as old:
            out = {k: old[k] for k in old.files}
        ground_truth_cases(out)
        sparse_loss_cases(out)
        np.savez_compressed(HERE / "reference_outputs.npz", **out)
        print("updated", HERE / "reference_outputs.npz", (HERE / "reference_outputs.npz").stat().st_size, "bytes")
        return

    def put(name, iters, hist):
        out[f"{name}/iters"] = np.int64(iters)
        out[f"{name}/hist"] = np.asarray(hist, dtype=np.float64)
        print(f"{name}: iters={iters} res[0]={hist[0]:.17g} res[-2]={hist[-2]:.17g} res[-1]={hist[-1]:.17g}", flush=True)

    # --- PCG with Jacobi M = diag(1/a_ii) as torch sparse CSR (test.py:74-79), SURVEY 8-c3 -------
    cases = [("poisson2d", 64), ("poisson2d", 256), ("poisson3d", 32), ("poisson3d", 64)]
    if not args.quick:
        cases += [("poisson3d", 100), ("poisson2d", 1024), ("poisson3d", 128)]
    for kind, n in cases:
        A = getattr(O, kind)(n)
        b = O.rhs(A.shape[0], 0)
        M = to_torch_csr(sp.diags(O.jacobi_dinv(A)).tocsr())
        iters, hist = run_ref_pcg(to_torch_csr(A), b, M)
        put(f"pcg_{kind}_{n}_jacobi", iters, hist)

    # --- identity M (test.py:70-72), nonzero x0, dense A as the reference's callers pass it --------
    A = O.poisson2d(64)
    b = O.rhs(A.shape[0], 0)
    iters, hist = run_ref_pcg(to_torch_csr(A), b, to_torch_csr(sp.eye(A.shape[0]).tocsr()))
    put("pcg_poisson2d_64_identity", iters, hist)
    x0 = np.random.default_rng(7).uniform(-1, 1, A.shape[0])
    iters, hist = run_ref_pcg(to_torch_csr(A), b, to_torch_csr(sp.diags(O.jacobi_dinv(A)).tocsr()), x0=x0)
    put("pcg_poisson2d_64_jacobi_x0seed7", iters, hist)
    A32 = O.poisson2d(32)
    b32 = O.rhs(A32.shape[0], 3)
    iters, hist = run_ref_pcg(torch.from_numpy(A32.toarray()), b32,
                              torch.from_numpy(np.diag(O.jacobi_dinv(A32))))  # dense A, dense M (train.py:93-100)
    put("pcg_poisson2d_32_dense_jacobi_bseed3", iters, hist)
    iters, hist = run_ref_pcg(to_torch_csr(A), b, to_torch_csr(sp.diags(O.jacobi_dinv(A)).tocsr()), max_iter=20)
    put("pcg_poisson2d_64_jacobi_maxiter20", iters, hist)

    # --- M = L L^T materialised as CSR and MULTIPLIED (test.py:81-88,100-105) ----------------------
    L = CO.ic0(A)
    iters, hist = run_ref_pcg(to_torch_csr(A), b, to_torch_csr((L @ L.T).tocsr()))
    put("pcg_poisson2d_64_ic0_multiply", iters, hist)
    Ll = O.learned_like_factor(A32, seed=0)
    iters, hist = run_ref_pcg(to_torch_csr(A32), b32, to_torch_csr((Ll @ Ll.T).tocsr()))
    put("pcg_poisson2d_32_learnedlike_multiply_bseed3", iters, hist)  # ill-conditioned M A: chaotic late history
    Lw = O.learned_like_factor(A, seed=1, scale=0.02, diag_sigma=0.1)  # well-conditioned stand-in
    iters, hist = run_ref_pcg(to_torch_csr(A), b, to_torch_csr((Lw @ Lw.T).tocsr()))
    put("pcg_poisson2d_64_learnedlike_wellcond_multiply", iters, hist)

    # --- IC(0) applied by triangular solves through the reference loop (duck-typed M) --------------
    iters, hist = run_ref_pcg(to_torch_csr(A), b, TriSolveOperator(L))
    put("pcg_poisson2d_64_ic0_solve", iters, hist)
    Au = O.unstructured_like(O.poisson3d(16), seed=0)
    bu = O.rhs(Au.shape[0], 0)
    iters, hist = run_ref_pcg(to_torch_csr(Au), bu, to_torch_csr(sp.diags(O.jacobi_dinv(Au)).tocsr()))
    put("pcg_unstructured3d_16_jacobi", iters, hist)
    iters, hist = run_ref_pcg(to_torch_csr(Au), bu, TriSolveOperator(CO.ic0(Au)))
    put("pcg_unstructured3d_16_ic0_solve", iters, hist)

    # --- conjugate_gradient (cg.py:20-47): errors = [(A-norm error, res)], x_hat --------------------
    x_true = np.random.default_rng(11).uniform(-1, 1, A32.shape[0])
    b_cg = A32 @ x_true
    errors, x_hat = ref_cg.conjugate_gradient(to_torch_csr(A32), torch.from_numpy(b_cg), x_true=torch.from_numpy(x_true))
    out["cg_poisson2d_32_xtrue11/err"] = np.array([float(e) for e, _ in errors])
    out["cg_poisson2d_32_xtrue11/hist"] = np.array([float(r) for _, r in errors])
    out["cg_poisson2d_32_xtrue11/x"] = x_hat.numpy().copy()
    print("cg_poisson2d_32_xtrue11: iters", len(errors) - 1, flush=True)
    errors, x_hat = ref_cg.conjugate_gradient(to_torch_csr(A), torch.from_numpy(b))
    out["cg_poisson2d_64/hist"] = np.array([float(r) for _, r in errors])
    out["cg_poisson2d_64/x"] = x_hat.numpy().copy()
    print("cg_poisson2d_64: iters", len(errors) - 1, flush=True)

    # --- stopping_criterion (cg.py:15-17) -----------------------------------------------------------
    r_sc = np.random.default_rng(5).uniform(-1, 1, 1000)
    b_sc = np.random.default_rng(6).uniform(-1, 1, 1000)
    out["stopping_criterion_seed5_6/value"] = np.float64(
        ref_cg.stopping_criterion(None, torch.from_numpy(r_sc), torch.from_numpy(b_sc)).item())

    # --- sparse_matvec_mul (utils.py:15-43): the reference's own KAT (tests/test_utils.py:11-41) ----
    idx = torch.tensor([[0, 0, 0], [0, 0, 1], [0, 1, 0], [0, 1, 1], [0, 2, 2],
                        [1, 0, 1], [1, 0, 2], [1, 1, 0], [1, 1, 1], [1, 2, 1]]).int()
    feat = torch.tensor([[1, 2, 3, 4, 5, 2, 3, 1, 4, 5]]).T.float()
    vec = torch.tensor([[1, 2, 3], [1, -1, 1]]).float()
    stub = _StubSparseConvTensor(feat, idx, 2)
    out["spmm_kat/indices"] = idx.numpy()
    out["spmm_kat/features"] = feat.numpy()
    out["spmm_kat/vectors"] = vec.numpy()
    out["spmm_kat/y"] = ref_utils.sparse_matvec_mul(stub, vec, transpose=False).numpy()
    out["spmm_kat/yt"] = ref_utils.sparse_matvec_mul(stub, vec, transpose=True).numpy()
    assert np.array_equal(out["spmm_kat/y"], np.array([[5, 11, 15], [1, -3, -5]], dtype=np.float32))
    # a seeded ragged batch: 3 systems, dof 40, unsorted COO with duplicates
    rng = np.random.default_rng(21)
    nnz, B, dof = 500, 3, 40
    idx_r = np.stack([rng.integers(0, B, nnz), rng.integers(0, dof, nnz), rng.integers(0, dof, nnz)], 1).astype(np.int32)
    feat_r = rng.standard_normal((nnz, 1)).astype(np.float32)
    vec_r = rng.standard_normal((B, dof)).astype(np.float32)
    stub = _StubSparseConvTensor(torch.from_numpy(feat_r), torch.from_numpy(idx_r), B)
    out["spmm_rand21/indices"], out["spmm_rand21/features"], out["spmm_rand21/vectors"] = idx_r, feat_r, vec_r
    out["spmm_rand21/y"] = ref_utils.sparse_matvec_mul(stub, torch.from_numpy(vec_r), transpose=False).numpy()
    out["spmm_rand21/yt"] = ref_utils.sparse_matvec_mul(stub, torch.from_numpy(vec_r), transpose=True).numpy()

    # --- benchmark_cg (utils.py:46-76): scipy cg, maxiter 512, rtol 1e-5 ----------------------------
    for n in (64, 256):
        An = O.poisson2d(n)
        bn = O.rhs(An.shape[0], 0)
        _, it0, info0 = ref_utils.benchmark_cg(An, bn)
        _, it1, info1 = ref_utils.benchmark_cg(An, bn, sp.diags(O.jacobi_dinv(An)).tocsr())
        out[f"benchmark_cg_poisson2d_{n}/none"] = np.array([it0, info0], dtype=np.int64)
        out[f"benchmark_cg_poisson2d_{n}/jacobi"] = np.array([it1, info1], dtype=np.int64)
        print(f"benchmark_cg {n}: none {(it0, info0)} jacobi {(it1, info1)}", flush=True)

    # --- edge cases of the drop-in signatures: what the reference RETURNS on degenerate arguments ----
    A8 = O.poisson2d(8)
    A8t, I8 = to_torch_csr(A8), to_torch_csr(sp.identity(64, format="csr"))
    b8 = torch.from_numpy(O.rhs(64, 0))
    b_nan = b8.clone()
    b_nan[3] = float("nan")
    edge = {
        "rtol_1": ref_cg.preconditioned_conjugate_gradient(A8t, b8, I8, rtol=1.0)[1:],          # 1.0 < 1.0 is false
        "rtol_1e9": ref_cg.preconditioned_conjugate_gradient(A8t, b8, I8, rtol=1e9)[1:],
        "max_iter_0": ref_cg.preconditioned_conjugate_gradient(A8t, b8, I8, max_iter=0)[1:],
        "max_iter_5": ref_cg.preconditioned_conjugate_gradient(A8t, b8, I8, max_iter=5)[1:],
        "b_zero_max30": ref_cg.preconditioned_conjugate_gradient(A8t, torch.zeros_like(b8), I8, max_iter=30)[1:],
        "b_nan_max30": ref_cg.preconditioned_conjugate_gradient(A8t, b_nan, I8, max_iter=30)[1:],
    }
    for k, v in edge.items():
        out[f"edge_pcg/{k}"] = np.array(v, dtype=np.int64)
    e0, x0_ = ref_cg.conjugate_gradient(A8t, b8, max_iter=0)
    e1, _ = ref_cg.conjugate_gradient(A8t, b8, rtol=1.0)
    out["edge_cg/max_iter_0_len"] = np.int64(len(e0))
    out["edge_cg/max_iter_0_x"] = x0_.numpy().copy()
    out["edge_cg/rtol_1_hist"] = np.array([float(r) for _, r in e1])
    print("edge cases:", {k: tuple(int(t) for t in v) for k, v in edge.items()}, len(e0), out["edge_cg/rtol_1_hist"], flush=True)

    # --- dense losses of metrics.py:34-100 on a duck-typed batch (only `.dense()` is used) --------------
    from uibk.deep_preconditioning import metrics as ref_metrics

    class DenseStub:
        def __init__(self, dense4):
            self._d = dense4

        def dense(self):
            return self._d.clone()

    gen = torch.Generator().manual_seed(123)
    nb, nn_ = 2, 12
    sys_low = torch.zeros(nb, 1, nn_, nn_)
    pre_low = torch.zeros(nb, 1, nn_, nn_)
    for bi in range(nb):
        Ad = torch.from_numpy(sp.diags([-1.0, 2.5 + bi, -1.0], [-1, 0, 1], shape=(nn_, nn_)).toarray()).float()
        sys_low[bi, 0] = torch.tril(Ad)
        Lr = torch.tril(torch.rand(nn_, nn_, generator=gen) * 0.2) + torch.diag(0.5 + torch.rand(nn_, generator=gen))
        pre_low[bi, 0] = Lr
    out["metrics/systems_tril"] = sys_low.numpy().copy()
    out["metrics/preconditioners_tril"] = pre_low.numpy().copy()
    out["metrics/inverse_loss"] = np.float64(ref_metrics.inverse_loss(DenseStub(sys_low), DenseStub(pre_low)))
    out["metrics/condition_loss"] = np.float64(ref_metrics.condition_loss(DenseStub(sys_low), DenseStub(pre_low)))
    torch.manual_seed(7)
    out["metrics/hutchinson_trace_seed7_cpu"] = np.float64(
        ref_metrics.hutchinson_trace(DenseStub(sys_low), DenseStub(pre_low)))
    print("metrics:", float(out["metrics/inverse_loss"]), float(out["metrics/condition_loss"]),
          float(out["metrics/hutchinson_trace_seed7_cpu"]), flush=True)

    ground_truth_cases(out)
    sparse_loss_cases(out)
    round3_cases(out)
    name = "reference_outputs_quick.npz" if args.quick else "reference_outputs.npz"
    np.savez_compressed(HERE / name, **out)
    print("wrote", HERE / name, (HERE / name).stat().st_size, "bytes")


if __name__ == "__main__":
    main()
