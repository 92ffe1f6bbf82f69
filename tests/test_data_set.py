"""Folder-backed data sets: files written the way generate_data.py:109-111 writes them come back as the item tuple of
data_set.py:73-130 / 173-214."""
import numpy as np
import scipy.sparse as sp
import torch

from deeppreconditioning_amd.data_set import SludgePatternDataSet, StAnDataSet
from oracle import oracle as O


def _write_case(folder, A, seed):
    folder.mkdir(parents=True)
    sp.save_npz(folder / "matrix.npz", sp.coo_matrix(A), compressed=False)          # generate_data.py:109
    rng = np.random.default_rng(seed)
    x = rng.uniform(-1, 1, A.shape[0])
    np.savetxt(folder / "solution.csv", x)                                            # generate_data.py:110
    np.savetxt(folder / "right_hand_side.csv", A @ x)                                 # generate_data.py:111
    return x


def test_sludge_pattern_folders(tmp_path):
    mats = [O.poisson2d(3 + (i % 3)) for i in range(10)]                             # 9, 16, 25 unknowns
    xs = [_write_case(tmp_path / "sludge_patterns" / f"case_{i:04d}", m, i) for i, m in enumerate(mats)]
    train = SludgePatternDataSet("train", batch_size=2, shuffle=False, root=tmp_path, device="cpu")
    test = SludgePatternDataSet("test", batch_size=1, shuffle=False, root=tmp_path, device="cpu")
    assert (len(train), len(test), train.dof_max) == (4, 2, 25)                      # 80/20 split of 10 sorted folders
    tril, sol, rhs, sizes = train[1]                                                 # cases 2 and 3
    assert sizes == (25, 9) and tril.batch_size == 2 and tril.spatial_shape == [25, 25]
    assert tril.features.dtype == torch.float32 and tril.indices.dtype == torch.int32
    dense = tril.dense()[:, 0].double().numpy()
    np.testing.assert_allclose(dense[0], sp.tril(mats[2]).toarray())
    expect = np.eye(25)
    expect[:9, :9] = sp.tril(mats[3]).toarray()                                      # identity rows pad to dof_max
    np.testing.assert_allclose(dense[1], expect)
    np.testing.assert_allclose(sol[1, :9].numpy(), xs[3], rtol=1e-6)
    assert torch.all(sol[1, 9:] == 1) and torch.all(rhs[1, 9:] == 1)                 # vectors padded with ones
    tril_t, _, _, sizes_t = test[0]
    assert sizes_t == (mats[8].shape[0],) and tril_t.batch_size == 1
    try:
        SludgePatternDataSet("validation", 1, root=tmp_path, device="cpu")
    except AssertionError:
        pass
    else:
        raise AssertionError("an invalid stage must be refused")


def test_stan_files(tmp_path):
    folder = tmp_path / "stand_small_test"
    folder.mkdir()
    A = sp.coo_matrix(O.poisson2d(4))
    x = np.arange(16, dtype=np.float64)
    np.savez(folder / "sample_0.npz", indices=np.vstack((A.row, A.col)), values=A.data, solution=x, rhs=A @ x)
    data = StAnDataSet("test", batch_size=1, shuffle=False, root=tmp_path, device="cpu")
    assert len(data) == 1 and data.dof_max == 5166
    tril, sol, rhs, sizes = data[0]
    assert sizes == (16,) and tril.spatial_shape == [5166, 5166]
    assert tril.features.shape[0] == sp.tril(A).nnz                                   # no trivial equations added
    assert torch.all(sol[0, 16:] == 0) and torch.equal(sol[0, :16], torch.from_numpy(x).float())
