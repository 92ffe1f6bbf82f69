"""spconv-free PreconditionerNet (deeppreconditioning_amd/model.py) against a dense torch fp32 restatement of the
same convolutions, plus the structural checks of the reference's own tests/test_model.py:25-42.  CPU; the GPU leg
that feeds the emitted L into the HIP solver is in tests/test_gpu_parity.py."""

import numpy as np
import pytest
import scipy.sparse as sp
import torch
import torch.nn.functional as F

from deeppreconditioning_amd import model as M
from deeppreconditioning_amd.utils import SparseBatch
from oracle import oracle as O

CHANNELS = [1, 16, 32, 64, 32, 16, 1]  # params.yaml:6-13


def _random_batch(B, H, W, C, density, seed):
    g = torch.Generator().manual_seed(seed)
    mask = torch.rand(B, H, W, generator=g) < density
    idx = mask.nonzero().int()
    feats = torch.randn(idx.shape[0], C, generator=g)
    return SparseBatch(feats, idx, [H, W], B), mask


def _dense_reference(conv, dense, mask):
    """conv2d on the dense image, outputs kept only where a sparse conv has an active site."""
    w = conv.weight.permute(0, 3, 1, 2)  # KRSC -> (out, in, kh, kw)
    y = F.conv2d(dense, w, conv.bias, padding=conv.padding)
    m = F.conv2d(mask.float().unsqueeze(1), torch.ones(1, 1, *conv.kernel_size), padding=conv.padding) > 0
    return y * m, m[:, 0]


@pytest.mark.parametrize("k,pad", [(1, 0), (2, (1, 0)), (2, (0, 1)), (3, 1)])
def test_sparse_conv_matches_dense_conv(k, pad):
    torch.manual_seed(0)
    t, mask = _random_batch(2, 13, 11, 5, 0.2, seed=k)
    conv = M.SparseConv2d(5, 7, k, padding=pad)
    out = conv(t)
    ref, m = _dense_reference(conv, t.dense(), mask)
    assert out.spatial_shape == list(ref.shape[2:])
    torch.testing.assert_close(out.dense(), ref, rtol=1e-5, atol=1e-5)
    # exactly the dilated active set, in (batch,row,col) order
    assert torch.equal(out.indices.long(), m.nonzero())


def test_forward_like_reference_test():
    """tests/test_model.py:25-42 of the reference: 64x64 identities, batch of 2."""
    torch.manual_seed(69)
    size, batch = 64, 2
    net = M.PreconditionerNet(CHANNELS)
    eye = sp.eye(size, format="csr")
    inp, sizes = M.tril_batch_from_csr([eye, eye])
    lower = net(inp).dense()
    assert lower.shape[2:] == torch.Size(inp.spatial_shape)
    for b in range(batch):
        assert torch.all(lower[b, 0].diag() != 0)
        assert torch.all(lower[b, 0].triu(diagonal=1) == 0)
        assert torch.any(lower[b, 0].tril(diagonal=-1) != 0)
    pre = lower.matmul(lower.transpose(-1, -2)).squeeze()
    assert pre.shape == (batch, size, size)
    assert torch.all(pre == pre.transpose(-1, -2))
    ev = torch.linalg.eigvalsh(pre.double())
    assert torch.all(ev > 0)
    # the same input built exactly as the reference's fixture does (SparseConvTensor.from_dense of stacked identities)
    from deeppreconditioning_amd.utils import SparseBatch
    dense_eye = torch.eye(size).unsqueeze(0).unsqueeze(0).expand(batch, -1, -1, -1)
    inp2 = SparseBatch.from_dense(dense_eye.permute(0, 2, 3, 1))
    assert inp2.batch_size == batch and inp2.spatial_shape == [size, size] and inp2.indices.dtype == torch.int32
    assert torch.equal(net(inp2).dense(), lower)


def test_network_matches_dense_restatement_and_dilation_pattern():
    torch.manual_seed(1)
    A = O.poisson2d(6)
    n = A.shape[0]
    net = M.PreconditionerNet(CHANNELS)
    inp, _ = M.tril_batch_from_csr([A, A.T.tocsr() * 0.5], dof_max=n + 3)   # padded with identity rows
    out = net(inp)
    # dense restatement of model.py:26-57
    x, mask = inp.dense(), inp.dense()[:, 0] != 0
    for layer in net.layers:
        if isinstance(layer, M.SparseConv2d):
            x, mask = _dense_reference(layer, x, mask)
        else:
            x = layer(x)
    N = inp.spatial_shape[0]
    r, c = torch.meshgrid(torch.arange(N), torch.arange(N), indexing="ij")
    x = torch.where(r < c, torch.zeros_like(x), x)
    x = torch.where((r == c) & mask.unsqueeze(1), F.softplus(x), x)
    torch.testing.assert_close(out.dense(), x, rtol=1e-4, atol=1e-5)
    # active set = input pattern dilated by [-2,2]^2 (four 2x2 windows), clipped to the image
    pat = torch.zeros(N, N, dtype=torch.bool)
    t0 = inp.dense()[0, 0] != 0
    for dr in range(-2, 3):
        for dc in range(-2, 3):
            src = t0[max(0, -dr):N - max(0, dr), max(0, -dc):N - max(0, dc)]
            pat[max(0, dr):N - max(0, -dr), max(0, dc):N - max(0, -dc)] |= src
    got = torch.zeros(N, N, dtype=torch.bool)
    i0 = out.indices[out.indices[:, 0] == 0].long()
    got[i0[:, 1], i0[:, 2]] = True
    assert torch.equal(got, pat)


def test_state_dict_layout_and_parameter_count():
    net = M.PreconditionerNet(CHANNELS)
    keys = list(net.state_dict())
    assert keys[:3] == ["layers.0.weight", "layers.0.bias", "layers.1.weight"] and "layers.10.bias" in keys
    assert sum(p.numel() for p in net.parameters()) == 20678   # matches the 87,844-byte checkpoint, dvc.lock:54
    assert net.layers[2].weight.shape == (32, 2, 2, 16)        # KRSC
    other = {k: (v.permute(1, 2, 3, 0).contiguous() if v.dim() == 4 else v.clone()) for k, v in net.state_dict().items()}
    net2 = M.PreconditionerNet(CHANNELS)
    M.load_reference_state_dict(net2, other)                   # (kh,kw,in,out) checkpoints are permuted on load
    for k, v in net.state_dict().items():
        assert torch.equal(v, net2.state_dict()[k])


def test_lower_factor_csr_roundtrip():
    torch.manual_seed(2)
    A = O.unstructured_like(O.poisson2d(5), 3)
    n = A.shape[0]
    net = M.PreconditionerNet([1, 4, 4, 4, 1])
    inp, sizes = M.tril_batch_from_csr([A, O.poisson2d(4)])
    out = net(inp)
    for b, size in enumerate(sizes):
        rp, ci, v = M.lower_factor_csr(out, b, size)
        L = sp.csr_matrix((v.numpy(), ci.numpy(), rp.numpy()), shape=(size, size))
        dense = out.dense()[b, 0, :size, :size].detach().double().numpy()
        np.testing.assert_allclose(L.toarray(), np.tril(dense), rtol=0, atol=0)
        assert np.all(L.diagonal() > 0)
        assert all(ci[rp[i + 1] - 1] == i for i in range(size))   # diagonal stored last


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
