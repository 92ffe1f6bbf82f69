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
