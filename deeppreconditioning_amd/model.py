"""spconv-free `PreconditionerNet` forward pass (drop-in for `uibk/deep_preconditioning/model.py:13-59`).

`spconv` ships CUDA-only wheels (`pyproject.toml:20`), so on ROCm the reference model cannot even be imported.
This module re-states the pieces the network needs on plain PyTorch-ROCm ops -- the host code the north star
keeps in Python: a `SparseConvTensor`-like container (`utils.SparseBatch`), a regular (output-dilating)
`SparseConv2d` evaluated as rulebook gather -> GEMM -> scatter-add, and `SparseSequential`.  Module/parameter names
follow the reference so that a `best.pt` state_dict (`train.py:183-186`, keys `layers.{i}.weight|bias`) loads.

Parity status: UNPINNED.  Neither spconv nor a checkpoint is available in the build container.  The arithmetic is
pinned instead to a dense `torch.nn.functional.conv2d` restatement (tests/test_model.py); the one convention that
cannot be verified here is spconv's weight layout, taken as KRSC `(out_channels, kh, kw, in_channels)` (spconv
2.3.8 default, `uv.lock:2084`).  `SparseConv2d.weight_layout` documents it; `load_reference_state_dict` accepts
the alternative `(kh, kw, in, out)` layout as well.
"""

from __future__ import annotations

import math

import numpy as np
import torch
from torch import nn

from .utils import SparseBatch


class SparseConv2d(nn.Module):
    """Regular sparse convolution, stride 1: an output site is active when any input site lies in its window
    (`spconv.SparseConv2d`, used at model.py:27,36,40).  out(y,x) = sum_{ky,kx} in(y+ky-py, x+kx-px) W[ky,kx] + b."""

    weight_layout = "KRSC"  # (out_channels, kh, kw, in_channels)

    def __init__(self, in_channels: int, out_channels: int, kernel_size, padding=0, bias: bool = True):
        super().__init__()
        ks = (kernel_size, kernel_size) if isinstance(kernel_size, int) else tuple(kernel_size)
        pd = (padding, padding) if isinstance(padding, int) else tuple(padding)
        self.in_channels, self.out_channels, self.kernel_size, self.padding = in_channels, out_channels, ks, pd
        self.weight = nn.Parameter(torch.empty(out_channels, ks[0], ks[1], in_channels))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        fan_in = in_channels * ks[0] * ks[1]
        nn.init.kaiming_uniform_(self.weight.view(out_channels, -1), a=math.sqrt(5))
        if self.bias is not None:
            bound = 1 / math.sqrt(fan_in)
            nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, t: SparseBatch) -> SparseBatch:
        kh, kw = self.kernel_size
        ph, pw = self.padding
        H, W = t.spatial_shape
        Ho, Wo = H + 2 * ph - kh + 1, W + 2 * pw - kw + 1
        w = self.weight.reshape(self.out_channels, kh * kw, self.in_channels)
        if (kh, kw, ph, pw) == (1, 1, 0, 0):                                  # pointwise: the site set is unchanged
            out = t.features @ w[:, 0, :].t()
            if self.bias is not None:
                out = out + self.bias
            return SparseBatch(out, t.indices, [Ho, Wo], t.batch_size)
        idx = t.indices.long()
        b, y, x = idx[:, 0], idx[:, 1], idx[:, 2]
        nnz = idx.shape[0]
        # rulebook: input site (y,x) feeds output site (y - ky + ph, x - kx + pw) through W[ky,kx].  No boolean
        # compaction (each would be a host sync): a contribution that falls outside the output gets a sentinel key,
        # which sorts last and owns one scratch row of `out` (a sentinel is always appended, so that row always exists).
        sentinel = t.batch_size * Ho * Wo
        keys = []
        for ky in range(kh):
            for kx in range(kw):
                oy, ox = y - ky + ph, x - kx + pw
                ok = (oy >= 0) & (oy < Ho) & (ox >= 0) & (ox < Wo)
                keys.append(torch.where(ok, (b * Ho + oy) * Wo + ox, sentinel))
        keys.append(torch.full((1,), sentinel, device=idx.device, dtype=torch.long))
        uniq, dst = torch.unique(torch.cat(keys), sorted=True, return_inverse=True)   # sites in (batch,row,col) order
        out = t.features.new_zeros((uniq.numel(), self.out_channels))
        for k in range(kh * kw):                                              # one GEMM per kernel offset
            out.index_add_(0, dst[k * nnz:(k + 1) * nnz], t.features @ w[:, k, :].t())
        uniq, out = uniq[:-1], out[:-1]                                       # drop the scratch row
        if self.bias is not None:
            out = out + self.bias
        ob = uniq // (Ho * Wo)
        rem = uniq - ob * (Ho * Wo)
        indices = torch.stack((ob, rem // Wo, rem % Wo), dim=1).to(t.indices.dtype)
        return SparseBatch(out, indices, [Ho, Wo], t.batch_size)


class SparseSequential(nn.Sequential):
    """`spconv.SparseSequential`: dense modules (PReLU) act on the feature matrix of the sparse tensor."""

    def add(self, module: nn.Module) -> None:
        self.add_module(str(len(self)), module)

    def forward(self, t: SparseBatch) -> SparseBatch:
        for module in self:
            t = module(t) if isinstance(module, SparseConv2d) else t.replace_feature(module(t.features))
        return t


class PreconditionerNet(nn.Module):
    """Fully convolutional network mapping matrices to lower triangular matrices (model.py:13-59)."""

    def __init__(self, channels: list[int]) -> None:
        super().__init__()
        assert len(channels) % 2                                             # model.py:23
        self.layers = SparseSequential(SparseConv2d(channels[0], channels[1], 1), nn.PReLU())   # model.py:26-29
        for index, (cin, cout) in enumerate(zip(channels[1:-2], channels[2:-1])):   # model.py:32-37
            padding = (1, 0) if index < (len(channels) - 2) // 2 else (0, 1)
            self.layers.add(SparseConv2d(cin, cout, 2, padding=padding))
            self.layers.add(nn.PReLU())
        self.layers.add(SparseConv2d(channels[-2], channels[-1], 1))         # model.py:40

    def forward(self, input_: SparseBatch) -> SparseBatch:
        """The `L` part of the `L @ L.T` preconditioner (model.py:42-59)."""
        interim = self.layers(input_)
        rows, cols = interim.indices[:, 1], interim.indices[:, 2]
        feats = interim.features
        feats = torch.where((rows < cols).unsqueeze(-1), torch.zeros_like(feats), feats)          # model.py:53-54
        feats = torch.where((rows == cols).unsqueeze(-1), nn.functional.softplus(feats), feats)   # model.py:56-57
        return interim.replace_feature(feats)


def load_reference_state_dict(model: PreconditionerNet, state: dict) -> None:
    """Load a reference checkpoint (`assets/checkpoints/best.pt`).  Conv weights stored as (kh,kw,in,out) -- the
    layout of spconv's `Native` algorithm -- are permuted to KRSC; KRSC weights load as they are."""
    fixed = {}
    own = model.state_dict()
    for k, v in state.items():
        if k in own and v.dim() == 4 and tuple(v.shape) != tuple(own[k].shape) and tuple(v.permute(3, 0, 1, 2).shape) == tuple(own[k].shape):
            v = v.permute(3, 0, 1, 2).contiguous()
        fixed[k] = v
    model.load_state_dict(fixed)


def tril_batch_from_csr(matrices, dof_max: int | None = None, device=None) -> tuple[SparseBatch, tuple[int, ...]]:
    """What the reference's data sets emit (data_set.py:73-130): the lower triangles of a batch of scipy matrices as
    one sparse batch tensor (features fp32 (nnz,1), indices int32 (batch,row,col)), padded to `dof_max` with
    identity rows (data_set.py:94-97).  Returns (batch, original_sizes)."""
    import scipy.sparse as sp
    sizes = tuple(int(m.shape[0]) for m in matrices)
    dof_max = max(sizes) if dof_max is None else dof_max
    feats, idxs = [], []
    for bi, m in enumerate(matrices):
        t = sp.tril(m, format="coo")
        pad = np.arange(sizes[bi], dof_max)
        rows = np.concatenate([t.row, pad])
        cols = np.concatenate([t.col, pad])
        vals = np.concatenate([t.data, np.ones(len(pad))])
        feats.append(vals)
        idxs.append(np.column_stack((np.full(len(vals), bi), rows, cols)))
    features = torch.from_numpy(np.concatenate(feats)).float().unsqueeze(-1)
    indices = torch.from_numpy(np.vstack(idxs)).int()
    if device is not None:
        features, indices = features.to(device), indices.to(device)
    return SparseBatch(features, indices, [dof_max, dof_max], len(matrices)), sizes


def lower_factor_csr(output: SparseBatch, batch_index: int, original_size: int):
    """L of one sample as CSR parts (rowptr int32, col int32, val float64) on the tensor's device: the entries
    with col <= row < original_size (test.py:102 slices `[:n,:n]`; the strict upper part is zero, model.py:54).
    Sites come out of `SparseConv2d` sorted by (batch,row,col), so the diagonal is last in each row."""
    idx, feats = output.indices.long(), output.features[:, 0]
    keep = (idx[:, 0] == batch_index) & (idx[:, 2] <= idx[:, 1]) & (idx[:, 1] < original_size)
    rows, cols, vals = idx[keep, 1], idx[keep, 2], feats[keep].detach().to(torch.float64)
    order = torch.argsort(rows * original_size + cols)
    rows, cols, vals = rows[order], cols[order], vals[order]
    rowptr = torch.zeros(original_size + 1, dtype=torch.int64, device=rows.device)
    rowptr[1:] = torch.cumsum(torch.bincount(rows, minlength=original_size), 0)
    return rowptr.to(torch.int32), cols.to(torch.int32), vals
