"""The fenced-off U-Net restatement (deeppreconditioning_amd/extras_unet.py: OUT OF SCOPE per SURVEY.md 8-f1 / section 2)
against dense torch restatements.  CPU."""

import numpy as np
import pytest
import scipy.sparse as sp
import torch
import torch.nn.functional as F

from deeppreconditioning_amd import model as M
from deeppreconditioning_amd.utils import SparseBatch
from oracle import oracle as O


# ---- U-Net variant (model.py:62-179): sub-manifold, strided and inverse sparse convolutions ------------------
def _random_sparse(batch, H, W, cin, density, seed):
    g = torch.Generator().manual_seed(seed)
    mask = torch.rand(batch, H, W, generator=g) < density
    mask[:, 0, 0] = True
    dense = torch.randn(batch, H, W, cin, generator=g) * mask.unsqueeze(-1)
    dense = torch.where(mask.unsqueeze(-1) & (dense == 0), torch.ones_like(dense), dense)
    from deeppreconditioning_amd.utils import SparseBatch
    return SparseBatch.from_dense(dense), dense.permute(0, 3, 1, 2), mask


def test_submanifold_conv_matches_masked_dense_conv():
    torch.manual_seed(1)
    t, dense, mask = _random_sparse(2, 11, 9, 3, 0.3, seed=5)
    conv = M.SubMConv2d(3, 4, 3, padding=1)
    out = conv(t)
    assert torch.equal(out.indices, t.indices)                               # the site set does not change
    ref = torch.nn.functional.conv2d(dense, conv.weight.permute(0, 3, 1, 2), conv.bias, padding=1)
    got = out.dense()
    np.testing.assert_allclose(got.detach().numpy(), (ref * mask.unsqueeze(1)).detach().numpy(), atol=1e-5)
    one = M.SubMConv2d(3, 2, 1)                                              # the pointwise output layer
    np.testing.assert_allclose(one(t).dense().detach().numpy(),
                               (torch.nn.functional.conv2d(dense, one.weight.permute(0, 3, 1, 2), one.bias)
                                * mask.unsqueeze(1)).detach().numpy(), atol=1e-5)


def test_strided_conv_and_its_inverse():
    torch.manual_seed(2)
    t, dense, mask = _random_sparse(2, 12, 10, 3, 0.25, seed=6)
    down = M.SparseConv2d(3, 5, 3, stride=2, padding=1, indice_key="d")
    out = down(t)
    ref = torch.nn.functional.conv2d(dense, down.weight.permute(0, 3, 1, 2), None, stride=2, padding=1)
    active = torch.nn.functional.conv2d(mask.float().unsqueeze(1), torch.ones(1, 1, 3, 3), stride=2, padding=1)[:, 0] > 0
    assert out.spatial_shape == list(ref.shape[2:])
    got_mask = torch.zeros_like(active)
    got_mask[out.indices[:, 0].long(), out.indices[:, 1].long(), out.indices[:, 2].long()] = True
    assert torch.equal(got_mask, active)                                     # active where any input lies in the window
    np.testing.assert_allclose(out.dense().detach().numpy(),
                               ((ref + down.bias.view(1, -1, 1, 1)) * active.unsqueeze(1)).detach().numpy(), atol=1e-5)
    # the inverse convolution walks the same rulebook backwards: out(i) = sum_k W[k] in(o) over the pairs (i, k, o)
    up = M.SparseInverseConv2d(5, 2, 3, indice_key="d", bias=False)
    back = up(out)
    assert torch.equal(back.indices, t.indices) and back.spatial_shape == t.spatial_shape
    w = up.weight.detach()                                                   # (cout, kh, kw, cin)
    dense_out = out.dense().detach()
    expect = torch.zeros(t.indices.shape[0], 2)
    for row, (b, y, x) in enumerate(t.indices.tolist()):
        for ky in range(3):
            for kx in range(3):
                ny, nx = y - ky + 1, x - kx + 1
                if ny >= 0 and nx >= 0 and ny % 2 == 0 and nx % 2 == 0 and ny // 2 < dense_out.shape[2] and nx // 2 < dense_out.shape[3]:
                    expect[row] += w[:, ky, kx, :] @ dense_out[b, :, ny // 2, nx // 2]
    np.testing.assert_allclose(back.features.detach().numpy(), expect.numpy(), atol=1e-5)


def test_sparse_add_union():
    from deeppreconditioning_amd.utils import SparseBatch
    a = SparseBatch(torch.tensor([[1.0], [2.0]]), torch.tensor([[0, 0, 0], [0, 1, 1]], dtype=torch.int32), [2, 2], 1)
    b = SparseBatch(torch.tensor([[10.0], [20.0]]), torch.tensor([[0, 1, 1], [0, 1, 0]], dtype=torch.int32), [2, 2], 1)
    s = M.sparse_add(a, b)
    assert torch.equal(s.dense()[0, 0], torch.tensor([[1.0, 0.0], [20.0, 12.0]]))
    assert torch.equal(M.sparse_add(a, a).features, 2 * a.features)


def test_unet_forward_properties_and_state_dict_names():
    torch.manual_seed(69)
    net = M.PreconditionerSparseUNet([1, 4, 8, 8, 8, 1])
    A = sp.diags([-1.0, -1.0, 4.0, -1.0, -1.0], [-6, -1, 0, 1, 6], shape=(36, 36), format="csr")
    inp, sizes = M.tril_batch_from_csr([A, A])
    out = net(inp)
    assert torch.equal(out.indices, inp.indices)            # the U-Net keeps the sparsity pattern of tril(A)
    lower = out.dense()
    for b in range(2):
        assert torch.all(lower[b, 0].diag() > 0)
        assert torch.all(lower[b, 0].triu(diagonal=1) == 0)
        assert torch.any(lower[b, 0].tril(diagonal=-1) != 0)
    pre = lower[:, 0] @ lower[:, 0].transpose(-1, -2)
    assert torch.all(torch.linalg.eigvalsh(pre.double()) > 0)
    keys = set(net.state_dict().keys())
    for name in ("enc1.0.weight", "down1.0.bias", "bottleneck.0.weight", "up3.0.weight", "dec0.0.bias", "out_conv.0.weight"):
        assert name in keys
    assert net.state_dict()["down1.0.weight"].shape == (8, 3, 3, 4)
