"""OUT OF SCOPE -- fenced off.  `PreconditionerSparseUNet` with its sub-manifold, strided and inverse sparse convolutions
(`uibk/deep_preconditioning/model.py:62-179`), restated on plain torch ops.

SURVEY.md section 8-f1 names `model.py:13-59` (`PreconditionerNet`) only, and section 2 marks the U-Net variant out of scope; this
file exists because `params.yaml` of the reference can select `model: PreconditionerSparseUNet` and the compat shim resolves the
reference's import lines.  Nothing here is on the hot path, nothing here is hand-written HIP, and no parity claim is made
beyond tests/test_unet_extras.py (dense `conv2d` restatements on CPU).  `deeppreconditioning_amd.model` resolves these names
lazily (module `__getattr__`).

Also here, for the same reason (the reference's scripts/compare_meshes.py:65 calls `metrics.condition_loss`): the two dense
diagnostics of metrics.py:58-100, `hutchinson_trace` and `condition_loss`, as plain torch restatements;
`deeppreconditioning_amd.metrics` resolves them lazily too."""

from __future__ import annotations

import math

import torch
from torch import nn

from .model import SparseConv2d, SparseSequential
from .utils import SparseBatch


class SubMConv2d(nn.Module):
    """Sub-manifold convolution (`spconv.SubMConv2d`, model.py:71,79,87,95,...): the output lives on the INPUT's active
    sites only; out(s) = sum over kernel offsets of W[ky,kx] in(s + (ky,kx) - pad) for active neighbours, + bias."""

    weight_layout = "KRSC"
    takes_sparse_batch = True

    def __init__(self, in_channels: int, out_channels: int, kernel_size, padding=0, bias: bool = True,
                 indice_key: str | None = None):
        super().__init__()
        ks = (kernel_size, kernel_size) if isinstance(kernel_size, int) else tuple(kernel_size)
        pd = (padding, padding) if isinstance(padding, int) else tuple(padding)
        self.in_channels, self.out_channels, self.kernel_size, self.padding = in_channels, out_channels, ks, pd
        self.indice_key = indice_key
        self.weight = nn.Parameter(torch.empty(out_channels, ks[0], ks[1], in_channels))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        nn.init.kaiming_uniform_(self.weight.view(out_channels, -1), a=math.sqrt(5))
        if self.bias is not None:
            bound = 1 / math.sqrt(in_channels * ks[0] * ks[1])
            nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, t: SparseBatch) -> SparseBatch:
        kh, kw = self.kernel_size
        ph, pw = self.padding
        H, W = t.spatial_shape
        w = self.weight.reshape(self.out_channels, kh * kw, self.in_channels)
        if (kh, kw) == (1, 1):
            out = t.features @ w[:, 0, :].t()
        else:
            idx = t.indices.long()
            b, y, x = idx[:, 0], idx[:, 1], idx[:, 2]
            keys = (b * H + y) * W + x                            # sorted: sites are kept in (batch,row,col) order
            order = None
            if not bool((keys[1:] > keys[:-1]).all()):             # (a hand-built batch may come unsorted)
                order = torch.argsort(keys)
                keys = keys[order]
            feats_sorted = t.features if order is None else t.features[order]
            zero_row = t.features.new_zeros((1, self.in_channels))
            table = torch.cat((feats_sorted, zero_row))           # row nnz: the inactive neighbour
            out = t.features.new_zeros((idx.shape[0], self.out_channels))
            for ky in range(kh):
                for kx in range(kw):
                    ny, nx = y + ky - ph, x + kx - pw
                    inside = (ny >= 0) & (ny < H) & (nx >= 0) & (nx < W)
                    nkey = (b * H + ny) * W + nx
                    pos = torch.searchsorted(keys, nkey).clamp_(max=keys.numel() - 1)
                    hit = inside & (keys[pos] == nkey)
                    src = torch.where(hit, pos, keys.numel())
                    out = out + table[src] @ w[:, ky * kw + kx, :].t()
        if self.bias is not None:
            out = out + self.bias
        return SparseBatch(out, t.indices, t.spatial_shape, t.batch_size, t.indice_dict)


class SparseInverseConv2d(nn.Module):
    """`spconv.SparseInverseConv2d` (model.py:104,112,120,128): undoes the site change of the `SparseConv2d` that
    registered `indice_key` -- the output lives on that convolution's INPUT sites, and every (input site i, offset k,
    output site o) pair of its rulebook is walked backwards: out(i) += W[k] in(o)."""

    weight_layout = "KRSC"
    takes_sparse_batch = True

    def __init__(self, in_channels: int, out_channels: int, kernel_size, indice_key: str, bias: bool = True):
        super().__init__()
        ks = (kernel_size, kernel_size) if isinstance(kernel_size, int) else tuple(kernel_size)
        self.in_channels, self.out_channels, self.kernel_size, self.indice_key = in_channels, out_channels, ks, indice_key
        self.weight = nn.Parameter(torch.empty(out_channels, ks[0], ks[1], in_channels))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        nn.init.kaiming_uniform_(self.weight.view(out_channels, -1), a=math.sqrt(5))
        if self.bias is not None:
            bound = 1 / math.sqrt(in_channels * ks[0] * ks[1])
            nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, t: SparseBatch) -> SparseBatch:
        book = t.indice_dict[self.indice_key]
        kh, kw = self.kernel_size
        w = self.weight.reshape(self.out_channels, kh * kw, self.in_channels)
        dst = book["dst"]                                         # (kh*kw, nnz_in): row of `t` each input site fed, or scratch
        table = torch.cat((t.features, t.features.new_zeros((1, self.in_channels))))
        assert table.shape[0] == book["scratch"], "the tensor does not live on the sites its indice_key registered"
        out = t.features.new_zeros((dst.shape[1], self.out_channels))
        for k in range(kh * kw):
            out = out + table[dst[k]] @ w[:, k, :].t()
        if self.bias is not None:
            out = out + self.bias
        return SparseBatch(out, book["in_indices"], book["in_shape"], t.batch_size, t.indice_dict)


def sparse_add(a: SparseBatch, b: SparseBatch) -> SparseBatch:
    """`spconv.functional.sparse_add` (model.py:150,154,158,162): feature-wise sum over the union of the active sites."""
    if a.indices is b.indices or (a.indices.shape == b.indices.shape and bool((a.indices == b.indices).all())):
        return a.replace_feature(a.features + b.features)
    H, W = a.spatial_shape
    key = lambda i: (i[:, 0].long() * H + i[:, 1].long()) * W + i[:, 2].long()    # noqa: E731
    uniq, inv = torch.unique(torch.cat((key(a.indices), key(b.indices))), sorted=True, return_inverse=True)
    out = a.features.new_zeros((uniq.numel(), a.features.shape[1]))
    out.index_add_(0, inv, torch.cat((a.features, b.features)))
    ob, rem = uniq // (H * W), uniq % (H * W)
    indices = torch.stack((ob, rem // W, rem % W), dim=1).to(a.indices.dtype)
    return SparseBatch(out, indices, a.spatial_shape, a.batch_size, a.indice_dict)


class PreconditionerSparseUNet(nn.Module):
    """U-Net inspired architecture (model.py:62-179): three strided encoder stages, a bottleneck, inverse convolutions
    back up with additive skip connections, and a pointwise output layer; then the same lower-triangular / positive
    diagonal post-processing as `PreconditionerNet`.  Module names follow the reference (state_dict keys)."""

    def __init__(self, channels: list[int]) -> None:
        super().__init__()
        c = channels
        act = nn.LeakyReLU
        seq = SparseSequential
        self.enc1 = seq(SubMConv2d(c[0], c[1], 3, padding=1, indice_key="subm1"), act())
        self.down1 = seq(SparseConv2d(c[1], c[2], 3, stride=2, padding=1, indice_key="down1"), act())
        self.enc2 = seq(SubMConv2d(c[2], c[2], 3, padding=1, indice_key="subm2"), act())
        self.down2 = seq(SparseConv2d(c[2], c[3], 3, stride=2, padding=1, indice_key="down2"), act())
        self.enc3 = seq(SubMConv2d(c[3], c[3], 3, padding=1, indice_key="subm3"), act())
        self.down3 = seq(SparseConv2d(c[3], c[4], 3, stride=2, padding=1, indice_key="down3"), act())
        self.enc4 = seq(SubMConv2d(c[4], c[4], 3, padding=1, indice_key="subm4"), act())
        self.bottleneck = seq(SparseConv2d(c[4], c[5], 3, stride=2, padding=1, indice_key="bneck"), act())
        self.up3 = seq(SparseInverseConv2d(c[5], c[4], 3, indice_key="bneck"), act())
        self.dec3 = seq(SubMConv2d(c[4], c[4], 3, padding=1, indice_key="subm4"), act())
        self.up2 = seq(SparseInverseConv2d(c[4], c[3], 3, indice_key="down3"), act())
        self.dec2 = seq(SubMConv2d(c[3], c[3], 3, padding=1, indice_key="subm3"), act())
        self.up1 = seq(SparseInverseConv2d(c[3], c[2], 3, indice_key="down2"), act())
        self.dec1 = seq(SubMConv2d(c[2], c[2], 3, padding=1, indice_key="subm2"), act())
        self.up0 = seq(SparseInverseConv2d(c[2], c[1], 3, indice_key="down1"), act())
        self.dec0 = seq(SubMConv2d(c[1], c[1], 3, padding=1, indice_key="subm1"), act())
        self.out_conv = seq(SubMConv2d(c[1], c[5], 1, padding=0))                     # model.py:137-139

    def forward(self, input_: SparseBatch) -> SparseBatch:
        input_ = SparseBatch(input_.features, input_.indices, input_.spatial_shape, input_.batch_size, {})
        enc1 = self.enc1(input_)                                                      # model.py:143-147
        enc2 = self.enc2(self.down1(enc1))
        enc3 = self.enc3(self.down2(enc2))
        enc4 = self.enc4(self.down3(enc3))
        bottleneck = self.bottleneck(enc4)
        dec3 = self.dec3(sparse_add(self.up3(bottleneck), enc4))                      # model.py:152-164
        dec2 = self.dec2(sparse_add(self.up2(dec3), enc3))
        dec1 = self.dec1(sparse_add(self.up1(dec2), enc2))
        dec0 = self.dec0(sparse_add(self.up0(dec1), enc1))
        interim = self.out_conv(dec0)
        rows, cols = interim.indices[:, 1], interim.indices[:, 2]
        feats = interim.features
        feats = torch.where((rows < cols).unsqueeze(-1), torch.zeros_like(feats), feats)          # model.py:169-170
        feats = torch.where((rows == cols).unsqueeze(-1), nn.functional.softplus(feats), feats)   # model.py:172-173
        return interim.replace_feature(feats)




def _dense_pair(systems_tril, preconditioners_tril):
    """Dense L (batch, N, N) and the mirrored dense A, as metrics.py:45-49,68-71,91-95 build them."""
    pre = preconditioners_tril.dense()[:, 0]
    systems = systems_tril.dense()[:, 0]
    systems = systems + torch.tril(systems, -1).transpose(-1, -2)
    return pre, systems


def hutchinson_trace(systems_tril, preconditioners_tril) -> torch.Tensor:
    """mean_b || (L_b L_b^T - A_b) v_b ||_2 for one standard-normal probe v_b per sample (metrics.py:58-78)."""
    pre, systems = _dense_pair(systems_tril, preconditioners_tril)
    vector = torch.randn(systems.shape[:2], device=systems.device, dtype=systems.dtype).unsqueeze(-1)
    interim = torch.bmm(pre, torch.bmm(pre.transpose(-1, -2), vector))
    interim = interim - torch.bmm(systems, vector)
    return torch.linalg.vector_norm(interim.squeeze(-1), ord=2, dim=1).mean()


def condition_loss(systems_tril, preconditioners_tril) -> torch.Tensor:
    """mean_b sigma_max / sigma_min of L_b L_b^T A_b (metrics.py:81-100)."""
    pre, systems = _dense_pair(systems_tril, preconditioners_tril)
    sigmas = torch.linalg.svdvals(torch.matmul(torch.matmul(pre, pre.transpose(-1, -2)), systems))
    return (sigmas.max(dim=1)[0] / sigmas.min(dim=1)[0]).mean()
