"""spconv-free forward pass of `PreconditionerNet` (drop-in for `uibk/deep_preconditioning/model.py:13-59`; the U-Net
variant of model.py:62-179 is out of scope and fenced off in extras_unet.py, resolved lazily).

`spconv` ships CUDA-only wheels (`pyproject.toml:20`), so on ROCm the reference model cannot even be imported.
This module re-states the pieces the network needs on plain PyTorch-ROCm ops -- the host code the north star
keeps in Python: a `SparseConvTensor`-like container (`utils.SparseBatch`), a regular (output-dilating, optionally
strided) `SparseConv2d` evaluated as rulebook gather -> GEMM -> scatter-add, and `SparseSequential`.  Module/parameter names
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
    takes_sparse_batch = True   # SparseSequential hands such modules the SparseBatch, others the feature matrix

    def __init__(self, in_channels: int, out_channels: int, kernel_size, stride=1, padding=0, bias: bool = True,
                 indice_key: str | None = None):
        super().__init__()
        ks = (kernel_size, kernel_size) if isinstance(kernel_size, int) else tuple(kernel_size)
        pd = (padding, padding) if isinstance(padding, int) else tuple(padding)
        st = (stride, stride) if isinstance(stride, int) else tuple(stride)
        self.in_channels, self.out_channels, self.kernel_size, self.padding = in_channels, out_channels, ks, pd
        self.stride, self.indice_key = st, indice_key
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
        sh, sw = self.stride
        Ho, Wo = (H + 2 * ph - kh) // sh + 1, (W + 2 * pw - kw) // sw + 1
        w = self.weight.reshape(self.out_channels, kh * kw, self.in_channels)
        if (kh, kw, ph, pw, sh, sw) == (1, 1, 0, 0, 1, 1):                    # pointwise: the site set is unchanged
            out = t.features @ w[:, 0, :].t()
            if self.bias is not None:
                out = out + self.bias
            return SparseBatch(out, t.indices, [Ho, Wo], t.batch_size, t.indice_dict)
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
                ny, nx = y - ky + ph, x - kx + pw                              # = stride * output coordinate
                oy, ox = torch.div(ny, sh, rounding_mode="floor"), torch.div(nx, sw, rounding_mode="floor")
                ok = (ny >= 0) & (nx >= 0) & (oy * sh == ny) & (ox * sw == nx) & (oy < Ho) & (ox < Wo)
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
        result = SparseBatch(out, indices, [Ho, Wo], t.batch_size, t.indice_dict)
        if self.indice_key is not None:       # (spconv's indice_key: who fed whom through which offset, kept for whoever pairs an inverse convolution with this one)
            result.indice_dict[self.indice_key] = {"dst": dst[:-1].view(kh * kw, nnz), "scratch": uniq.numel() + 1,
                                                   "in_indices": t.indices, "in_shape": list(t.spatial_shape)}
        return result


class SparseSequential(nn.Sequential):
    """`spconv.SparseSequential`: dense modules (PReLU) act on the feature matrix of the sparse tensor."""

    def add(self, module: nn.Module) -> None:
        self.add_module(str(len(self)), module)

    def forward(self, t: SparseBatch) -> SparseBatch:
        for module in self:
            sparse = getattr(module, "takes_sparse_batch", False)
            t = module(t) if sparse else t.replace_feature(module(t.features))
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
        """The `L` part of the `L @ L.T` preconditioner (model.py:42-59).

        Inference on the GPU (no autograd graph wanted) runs the hand-written HIP path -- `dpcg_convnet_*` in
        include/dpcg.h: rulebooks built on the device once per sparsity pattern, gathered GEMMs on the fp32 matrix cores
        with bias / PReLU fused, the tril mask and the softplus of model.py:53-57 fused into the last layer, L written
        straight into a lower-triangular CSR (`output.lower_csr`, what `lower_factor_csr` / `LLtMultiply` take).  Training
        (autograd) and CPU tensors take the torch ops below; `DPCG_CNN_TORCH=1` forces them."""
        if _hip_forward_applies(self, input_):
            return _hip_forward(self, input_)
        interim = self.layers(input_)
        rows, cols = interim.indices[:, 1], interim.indices[:, 2]
        feats = interim.features
        feats = torch.where((rows < cols).unsqueeze(-1), torch.zeros_like(feats), feats)          # model.py:53-54
        feats = torch.where((rows == cols).unsqueeze(-1), nn.functional.softplus(feats), feats)   # model.py:56-57
        return interim.replace_feature(feats)


# ---- the HIP path of PreconditionerNet.forward (inference) -----------------------------------------------------------
class _ConvnetPlan:
    """Owner of a `dpcg_convnet_plan_t` (active sites + rulebooks of every layer for ONE sparsity pattern).  `rebuild`
    re-targets it at another pattern reusing its device memory (dpcg_convnet_plan_rebuild)."""

    def __init__(self, indices: torch.Tensor, batch: int, shape, kernels, paddings):
        import ctypes as C
        self.handle = C.c_void_p()
        self._build(indices, batch, shape, kernels, paddings, create=True)

    def rebuild(self, indices: torch.Tensor, batch: int, shape, kernels, paddings) -> None:
        self._build(indices, batch, shape, kernels, paddings, create=False)

    def _build(self, indices, batch, shape, kernels, paddings, create: bool) -> None:
        import ctypes as C
        from . import _lib as L
        self._L = L
        self.indices = indices                         # kept alive: the cache key is its storage
        n = len(kernels)
        k = (C.c_int32 * (2 * n))(*[v for kk in kernels for v in kk])
        p = (C.c_int32 * (2 * n))(*[v for pp in paddings for v in pp])
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        args = (int(batch), int(shape[0]), int(shape[1]), int(indices.shape[0]), C.c_void_p(indices.data_ptr()), n, k, p, stream)
        if create:
            L.check(L.lib().dpcg_convnet_plan_create(C.byref(self.handle), *args))
        else:
            L.check(L.lib().dpcg_convnet_plan_rebuild(self.handle, *args))
        sites, h, w, nl = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
        L.check(L.lib().dpcg_convnet_plan_info(self.handle, n - 1, C.byref(sites), C.byref(h), C.byref(w), C.byref(nl)))
        self.sites, self.out_shape, self.nnz_lower, self.batch = sites.value, [h.value, w.value], nl.value, int(batch)
        dev = indices.device
        # fresh output arrays per pattern (torch's caching allocator: no device allocation): results of an earlier pattern
        # that the caller still holds stay valid
        self.out_indices = torch.empty((self.sites, 3), dtype=torch.int32, device=dev)
        self.lower_rowptr = torch.empty(self.batch * h.value + 1, dtype=torch.int32, device=dev)
        self.lower_col = torch.empty(self.nnz_lower, dtype=torch.int32, device=dev)
        L.check(L.lib().dpcg_convnet_plan_output(self.handle, C.c_void_p(self.out_indices.data_ptr()),
                                                 C.c_void_p(self.lower_rowptr.data_ptr()), C.c_void_p(self.lower_col.data_ptr()),
                                                 stream))

    def close(self):
        if self.handle is not None and self.handle.value:
            self._L.lib().dpcg_convnet_plan_destroy(self.handle)
            self.handle = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


def _conv_layers(net):
    """[(conv, prelu or None)] of a PreconditionerNet, or None when a module is not one the HIP path knows."""
    out = []
    for m in net.layers:
        if isinstance(m, SparseConv2d):
            out.append([m, None])
        elif isinstance(m, nn.PReLU) and out and out[-1][1] is None and m.weight.numel() == 1:
            out[-1][1] = m
        else:
            return None
    return out


def _hip_forward_applies(net, t: SparseBatch) -> bool:
    import os
    if os.environ.get("DPCG_CNN_TORCH") == "1" or torch.is_grad_enabled():
        return False
    if not (t.features.is_cuda and t.features.dtype == torch.float32 and t.indices.dtype == torch.int32):
        return False
    layers = _conv_layers(net)
    if not layers:
        return False
    for conv, _ in layers:
        kh, kw = conv.kernel_size
        if conv.stride != (1, 1) or kh * kw > 4 or conv.weight.dtype != torch.float32:
            return False
    last, last_act = layers[-1]
    return (last.kernel_size == (1, 1) and last.padding == (0, 0) and last.out_channels == 1 and last_act is None
            and last.in_channels in (16, 32, 64) and t.features.shape[1] == layers[0][0].in_channels)


def _hip_forward(net, t: SparseBatch) -> SparseBatch:
    return hip_conv_stack(_conv_layers(net), t, lower=True, cache=net.__dict__.setdefault("_hip_plans", {}))


def hip_conv_stack(layers, t: SparseBatch, lower: bool = False, cache: dict | None = None, _sorted_once: bool = False) -> SparseBatch:
    """Run `[(SparseConv2d, nn.PReLU | None), ...]` (stride 1, windows up to 2 x 2) on the HIP path (`dpcg_convnet_*`).
    lower=True: the last layer is pointwise with one output channel and model.py:53-57 is fused into it (`lower_csr`)."""
    import ctypes as C
    from . import _lib as L
    layers = [list(x) for x in layers]
    cache = {} if cache is None else cache
    indices = t.indices.contiguous()
    feats = t.features.contiguous()
    kernels = [tuple(c.kernel_size) for c, _ in layers]
    paddings = [tuple(c.padding) for c, _ in layers]
    # (tensors created under torch.inference_mode() track no version counter: reading it raises)
    version = 0 if indices.is_inference() else indices._version
    key = (indices.data_ptr(), version, indices.shape[0], tuple(t.spatial_shape), t.batch_size, tuple(kernels), tuple(paddings))
    plan = cache.get(key)
    with torch.cuda.device(feats.device):
        if plan is None:
            # a plan holds rulebooks and feature buffers: keep a few, and re-target the oldest one at the new pattern
            # (its device memory is reused: a data set of similar matrices then costs no allocations per matrix)
            recycled_key = next(iter(cache)) if len(cache) >= 2 else None
            try:
                if recycled_key is not None:
                    plan = cache.pop(recycled_key)
                    try:
                        plan.rebuild(indices, t.batch_size, t.spatial_shape, kernels, paddings)
                    except L.DpcgError:
                        plan.close()
                        raise
                else:
                    plan = _ConvnetPlan(indices, t.batch_size, t.spatial_shape, kernels, paddings)
            except L.DpcgError as exc:
                if "sorted" not in str(exc) or _sorted_once:        # (sorting does not remove duplicate sites: one retry only)
                    raise
                # sites in another order: sort them once (spconv accepts any order; the data sets emit sorted ones)
                H, W = t.spatial_shape
                k = (indices[:, 0].long() * H + indices[:, 1].long()) * W + indices[:, 2].long()
                order = torch.argsort(k)
                return hip_conv_stack(layers, SparseBatch(feats[order], indices[order].contiguous(), t.spatial_shape, t.batch_size),
                                      lower, cache, _sorted_once=True)
            cache[key] = plan
        n = len(layers)
        chan = (C.c_int32 * (n + 1))(*([layers[0][0].in_channels] + [c.out_channels for c, _ in layers]))
        ptr = lambda x: C.c_void_p(x.data_ptr()) if x is not None else C.c_void_p()       # noqa: E731
        keep = [c.weight.detach().contiguous() for c, _ in layers]
        w = (C.c_void_p * n)(*[ptr(x) for x in keep])
        b = (C.c_void_p * n)(*[ptr(c.bias.detach() if c.bias is not None else None) for c, _ in layers])
        a = (C.c_void_p * n)(*[ptr(act.weight.detach() if act is not None else None) for _, act in layers])
        out_feats = torch.empty((plan.sites, layers[-1][0].out_channels), dtype=torch.float32, device=feats.device)
        lower_val = torch.empty(plan.nnz_lower, dtype=torch.float64, device=feats.device) if lower else None
        L.check(L.lib().dpcg_convnet_forward(plan.handle, chan, w, b, a, ptr(feats), ptr(out_feats), ptr(lower_val),
                                             1 if lower else 0, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    out = SparseBatch(out_feats, plan.out_indices, plan.out_shape, t.batch_size)
    if lower:
        out.lower_csr = (plan.lower_rowptr, plan.lower_col, lower_val)  # rows = batch * height, sample b at [b * H, (b + 1) * H)
    return out


def forward_cost(net, t: SparseBatch) -> dict:
    """Flop and byte model of `PreconditionerNet.forward` on the HIP path for the pattern of `t` (whose plan must be cached: call
    the net once first).  Per layer: sites = active output sites; flops = 2 * taps * C_in * C_out * sites (every tap of every
    output site is a C_in x C_out product on the matrix cores, absent neighbours included -- they are multiplied as zeros);
    bytes = the minimum the layer moves through HBM: its input features once (sites_in * C_in * 4), its output features
    (sites * C_out * 4; the last layer writes fp64 values of the lower triangle instead) and its rulebook (taps * sites * 4)."""
    layers = _conv_layers(net)
    plans = net.__dict__.get("_hip_plans", {})
    plan = next((p for p in plans.values() if p.indices.data_ptr() == t.indices.contiguous().data_ptr()), None)
    if plan is None or layers is None:
        raise ValueError("forward_cost: run the net on this input first (HIP path, cached plan)")
    import ctypes as C
    from . import _lib as L
    out, sites_in, total_f, total_b = [], int(t.indices.shape[0]), 0, 0
    for li, (conv, _) in enumerate(layers):
        sites, h, w, nl = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
        L.check(L.lib().dpcg_convnet_plan_info(plan.handle, li, C.byref(sites), C.byref(h), C.byref(w), C.byref(nl)))
        taps = conv.kernel_size[0] * conv.kernel_size[1]
        flops = 2 * taps * conv.in_channels * conv.out_channels * sites.value
        last = li == len(layers) - 1
        byts = sites_in * conv.in_channels * 4 + (plan.nnz_lower * 8 if last else sites.value * conv.out_channels * 4) + taps * sites.value * 4
        out.append({"layer": li, "kernel": list(conv.kernel_size), "c_in": conv.in_channels, "c_out": conv.out_channels,
                    "sites": sites.value, "flops": flops, "min_hbm_bytes": byts})
        total_f += flops
        total_b += byts
        sites_in = sites.value
    return {"layers": out, "flops": total_f, "min_hbm_bytes": total_b}


# `PreconditionerSparseUNet` and its sub-manifold / inverse convolutions (model.py:62-179 of the reference) are OUTSIDE the
# hot-path scope (SURVEY.md 8-f1 names model.py:13-59 only): they live, fenced off, in extras_unet.py and resolve lazily so
# that `params.yaml: model: PreconditionerSparseUNet` (test.py:215, train.py:154) and the reference's import lines keep working.
_UNET_NAMES = ("SubMConv2d", "SparseInverseConv2d", "sparse_add", "PreconditionerSparseUNet")


def __getattr__(name):
    if name in _UNET_NAMES:
        from . import extras_unet
        return getattr(extras_unet, name)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")


def load_reference_state_dict(model: nn.Module, state: dict) -> None:
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
    if getattr(output, "lower_csr", None) is not None:      # the HIP forward wrote L into a lower-triangular CSR already
        rp, ci, v = output.lower_csr
        H = output.spatial_shape[0]
        lo, hi = batch_index * H, batch_index * H + original_size
        ends = rp[[lo, hi]].tolist()                          # one small copy to size the slices
        return (rp[lo:hi + 1] - ends[0]).contiguous(), ci[ends[0]:ends[1]].contiguous(), v[ends[0]:ends[1]].contiguous()
    idx, feats = output.indices.long(), output.features[:, 0]
    keep = (idx[:, 0] == batch_index) & (idx[:, 2] <= idx[:, 1]) & (idx[:, 1] < original_size)
    rows, cols, vals = idx[keep, 1], idx[keep, 2], feats[keep].detach().to(torch.float64)
    order = torch.argsort(rows * original_size + cols)
    rows, cols, vals = rows[order], cols[order], vals[order]
    rowptr = torch.zeros(original_size + 1, dtype=torch.int64, device=rows.device)
    rowptr[1:] = torch.cumsum(torch.bincount(rows, minlength=original_size), 0)
    return rowptr.to(torch.int32), cols.to(torch.int32), vals
