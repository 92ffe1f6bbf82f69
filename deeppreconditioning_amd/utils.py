"""GPU drop-ins for `uibk/deep_preconditioning/utils.py` (same names and signatures)."""

from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L
from .operators import CsrSystem, _dev_ptr, _stream


def _coo_spmv(idx, feat, vec, transpose: bool) -> torch.Tensor:
    batch, dof = vec.shape
    out = torch.empty_like(vec)
    with torch.cuda.device(vec.device):
        L.check(L.lib().dpcg_batched_coo_spmv(idx.shape[0], _dev_ptr(idx), _dev_ptr(feat), batch, dof, _dev_ptr(vec),
                                              _dev_ptr(out), 1 if transpose else 0, _stream()))
    return out


class _SparseMatvec(torch.autograd.Function):
    """y[b] = A[b] v[b] (or A[b]^T v[b]) on COO triples, with gradients for training through it
    (metrics.frobenius_loss, metrics.py:28-29): dL/dv = A^T g (the same kernel with `transpose` flipped),
    dL/dfeature_k = g[b,row_k] * v[b,col_k] (dpcg_batched_coo_edge)."""

    @staticmethod
    def forward(ctx, feat, vec, idx, transpose):
        ctx.save_for_backward(feat, vec, idx)
        ctx.transpose = bool(transpose)
        return _coo_spmv(idx, feat, vec, ctx.transpose)

    @staticmethod
    def backward(ctx, g):
        feat, vec, idx = ctx.saved_tensors
        g = g.contiguous()
        g_feat = g_vec = None
        if ctx.needs_input_grad[0]:
            g_feat = torch.empty_like(feat)
            batch, dof = vec.shape
            with torch.cuda.device(vec.device):
                L.check(L.lib().dpcg_batched_coo_edge(idx.shape[0], _dev_ptr(idx), batch, dof, _dev_ptr(g), _dev_ptr(vec),
                                                      _dev_ptr(g_feat), 1 if ctx.transpose else 0, _stream()))
        if ctx.needs_input_grad[1]:
            g_vec = _coo_spmv(idx, feat, g, not ctx.transpose)
        return g_feat, g_vec, None, None


def _coo_spmm(idx, feat, panel, transpose: bool) -> torch.Tensor:
    batch, dof, ncols = panel.shape
    out = torch.empty_like(panel)
    with torch.cuda.device(panel.device):
        L.check(L.lib().dpcg_batched_coo_spmm(idx.shape[0], _dev_ptr(idx), _dev_ptr(feat), batch, dof, ncols, _dev_ptr(panel),
                                              _dev_ptr(out), 1 if transpose else 0, _stream()))
    return out


class _SparseMatmat(torch.autograd.Function):
    """Y[b] = A[b] P[b] (or A[b]^T P[b]) for a panel P of `ncols` columns per sample -- `_SparseMatvec` with several
    right-hand sides; gradients: dL/dP = A^T G (same kernel, `transpose` flipped), dL/dfeature_k = <G[b,row_k,:],
    P[b,col_k,:]> (dpcg_batched_coo_sddmm).  Used by `metrics.inverse_loss`."""

    @staticmethod
    def forward(ctx, feat, panel, idx, transpose):
        ctx.save_for_backward(feat, panel, idx)
        ctx.transpose = bool(transpose)
        return _coo_spmm(idx, feat, panel, ctx.transpose)

    @staticmethod
    def backward(ctx, g):
        feat, panel, idx = ctx.saved_tensors
        g = g.contiguous()
        g_feat = g_panel = None
        if ctx.needs_input_grad[0]:
            g_feat = torch.empty_like(feat)
            batch, dof, ncols = panel.shape
            with torch.cuda.device(panel.device):
                L.check(L.lib().dpcg_batched_coo_sddmm(idx.shape[0], _dev_ptr(idx), batch, dof, ncols, _dev_ptr(g),
                                                       _dev_ptr(panel), _dev_ptr(g_feat), 1 if ctx.transpose else 0, _stream()))
        if ctx.needs_input_grad[1]:
            g_panel = _coo_spmm(idx, feat, g, not ctx.transpose)
        return g_feat, g_panel, None, None


def sparse_matmat_mul(spconv_batch, panel: torch.Tensor, transpose: bool) -> torch.Tensor:
    """`sparse_matvec_mul` for a panel (batch, dof, ncols) of right-hand sides, differentiable."""
    if not panel.is_cuda:
        raise ValueError("sparse_matmat_mul runs on the GPU: the panel must be a CUDA tensor")
    dev = panel.device
    idx = spconv_batch.indices.to(device=dev, dtype=torch.int32).contiguous()
    feat = spconv_batch.features.to(device=dev, dtype=torch.float32).reshape(-1).contiguous()
    if panel.shape[0] != spconv_batch.batch_size:
        raise ValueError("batch size mismatch")
    return _SparseMatmat.apply(feat, panel.to(torch.float32).contiguous(), idx, bool(transpose))


def sparse_matvec_mul(spconv_batch, vector_batch: torch.Tensor, transpose: bool) -> torch.Tensor:
    """Batched sparse matrix-vector product on COO triples (utils.py:15-43), differentiable.

    `spconv_batch` needs `.indices` (nnz,3) int32 `(batch,row,col)`, `.features` (nnz,1) and
    `.batch_size` -- an spconv `SparseConvTensor` or this package's `SparseBatch`.
    """
    dev = vector_batch.device
    if not vector_batch.is_cuda:
        raise ValueError("sparse_matvec_mul runs on the GPU: vector_batch must be a CUDA tensor")
    idx = spconv_batch.indices.to(device=dev, dtype=torch.int32).contiguous()
    feat = spconv_batch.features.to(device=dev, dtype=torch.float32).reshape(-1).contiguous()
    vec = vector_batch.to(torch.float32).contiguous()
    if vec.shape[0] != spconv_batch.batch_size:
        raise ValueError("batch size mismatch")
    return _SparseMatvec.apply(feat, vec, idx, bool(transpose)).to(vector_batch.dtype)


def benchmark_cg(matrix, right_hand_side, preconditioner=None) -> tuple[float, int, int]:
    """`scipy.sparse.linalg.cg(matrix, rhs, maxiter=512, M=preconditioner)` semantics (utils.py:46-76)
    on the GPU: stop when ||r|| < 1e-5 ||b|| (scipy's default rtol, tested on r before each update),
    at most 512 updates; returns `(duration, iterations, info)` with scipy's info (0 or 512).
    """
    system = CsrSystem.from_any(matrix)
    system.set_preconditioner(preconditioner)
    rtol = 1e-5
    result = system.solve(right_hand_side, None, rtol_sq=rtol * rtol, max_iter=512, flags=L.INIT_CHECK_R,
                          want_history=False)
    info = 0 if result.status == L.OK else 512
    return result.seconds, result.iterations, info


class SparseBatch:
    """Minimal stand-in for `spconv.pytorch.SparseConvTensor` (absent on ROCm): the attributes the
    reference's hot path touches (utils.py:26-35, data_set.py:122-125)."""

    def __init__(self, features: torch.Tensor, indices: torch.Tensor, spatial_shape, batch_size: int,
                 indice_dict: dict | None = None):
        self.features = features
        self.indices = indices
        self.spatial_shape = list(spatial_shape)
        self.batch_size = int(batch_size)
        self.indice_dict = {} if indice_dict is None else indice_dict   # rulebooks by `indice_key`, as spconv keeps them

    @classmethod
    def from_dense(cls, x: torch.Tensor) -> "SparseBatch":
        """As `SparseConvTensor.from_dense`: x is (batch, H, W, channels); a site is active where any channel is
        non-zero; indices are int32 (batch, row, col) in row-major order (tests/test_model.py:20-23 of the reference)."""
        active = (x != 0).any(dim=-1)
        idx = active.nonzero().to(torch.int32)
        feats = x[active]
        return cls(feats, idx, x.shape[1:3], x.shape[0])

    def replace_feature(self, features: torch.Tensor) -> "SparseBatch":
        return SparseBatch(features, self.indices, self.spatial_shape, self.batch_size, self.indice_dict)

    def dense(self) -> torch.Tensor:
        """(batch, channels, H, W) dense tensor, as `SparseConvTensor.dense()`."""
        ch = self.features.shape[1]
        out = torch.zeros(self.batch_size, ch, *self.spatial_shape, dtype=self.features.dtype,
                          device=self.features.device)
        b, r, c = (self.indices[:, i].long() for i in range(3))
        out[b, :, r, c] = self.features
        return out
