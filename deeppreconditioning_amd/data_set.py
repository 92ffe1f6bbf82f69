"""Folder-backed data sets with the reference's constructors and item tuples (`uibk/deep_preconditioning/data_set.py`),
so that `BenchmarkSuite` and a training loop can be pointed at the reference's files unchanged.

An item is `(systems_tril, solutions, right_hand_sides, original_sizes)`: the lower triangles of `batch_size` matrices
as one sparse batch tensor (features fp32 (nnz, 1), indices int32 (batch, row, col), spatial shape dof_max x dof_max),
the solution and right-hand-side batches as fp32 (batch, dof_max), and the true sizes (data_set.py:73-130,173-214).
File parsing is host plumbing; nothing here touches the solve path.  Not provided: the Kaggle download and the random
SPD generator (data_set.py:132-139,217-), which need network access / are not read by train.py or test.py.
"""

from __future__ import annotations

import pathlib
import random

import numpy as np
import torch

from .utils import SparseBatch

ROOT = pathlib.Path("./assets/data/raw/")     # data_set.py:20


def _device(device):
    if device is not None:
        return torch.device(device)
    assert torch.cuda.is_available(), "CUDA is mandatory but not available"      # data_set.py:54,170
    return torch.device("cuda")


def _batch(samples, dof_max: int, pad_value: float, device) -> tuple:
    """samples: (rows, cols, values, solution, right_hand_side) per batch entry, full symmetric triplets."""
    feats, idxs, sols, rhss, sizes = [], [], [], [], ()
    for bi, (rows, cols, vals, sol, rhs) in enumerate(samples):
        n = len(sol)
        sizes += (n,)
        keep = rows >= cols                                       # lower triangle, symmetry (data_set.py:88-91,193-195)
        rows, cols, vals = rows[keep], cols[keep], vals[keep]
        if pad_value:                                             # sludge: trivial equations up to dof_max (data_set.py:93-96)
            pad = np.arange(n, dof_max)
            rows, cols = np.append(rows, pad), np.append(cols, pad)
            vals = np.append(vals, np.ones(len(pad)))
        feats.append(np.expand_dims(vals, -1))
        idxs.append(np.column_stack((np.full(len(vals), bi), rows, cols)))
        sols.append(np.pad(sol, (0, dof_max - n), constant_values=pad_value)[None])
        rhss.append(np.pad(rhs, (0, dof_max - n), constant_values=pad_value)[None])
    features = torch.from_numpy(np.vstack(feats)).float().to(device)
    indices = torch.from_numpy(np.vstack(idxs)).int().to(device)
    tril = SparseBatch(features, indices, [dof_max, dof_max], len(samples))
    return (tril, torch.from_numpy(np.vstack(sols)).float().to(device),
            torch.from_numpy(np.vstack(rhss)).float().to(device), sizes)


class SludgePatternDataSet(torch.utils.data.Dataset):
    """`<root>/sludge_patterns/case_*/{matrix.npz, solution.csv, right_hand_side.csv}` (generate_data.py:97-111), 80/20
    train/test split of the sorted folders (data_set.py:26-56)."""

    def __init__(self, stage: str, batch_size: int, shuffle: bool = True, root=ROOT, device=None) -> None:
        self._folders = sorted(pathlib.Path(root, "sludge_patterns").glob("case_*"))
        cut = len(self._folders) * 80 // 100
        if stage == "train":
            self.folders = self._folders[:cut]
        elif stage == "test":
            self.folders = self._folders[cut:]
        else:
            raise AssertionError(f"Invalid stage {stage}")
        if shuffle:
            random.shuffle(self.folders)
        self.batch_size = batch_size
        self.dof_max = self._compute_max_dof()
        self.device = _device(device)

    def _compute_max_dof(self) -> int:
        sizes = [int(np.load(f / "matrix.npz")["shape"].max()) for f in self._folders]     # data_set.py:57-69
        assert sizes and max(sizes) > 0, "Maximum degrees of freedom is zero"
        return max(sizes)

    def __len__(self) -> int:
        return len(self.folders) // self.batch_size

    def __getitem__(self, index: int):
        samples = []
        for bi in range(self.batch_size):
            folder = self.folders[index * self.batch_size + bi]
            with np.load(folder / "matrix.npz") as f:
                rows, cols, _fmt, _shape, vals = (f[k] for k in f.files)                 # data_set.py:85
            samples.append((rows, cols, vals, np.atleast_1d(np.loadtxt(folder / "solution.csv")),
                            np.atleast_1d(np.loadtxt(folder / "right_hand_side.csv"))))
        return _batch(samples, self.dof_max, 1.0, self.device)


class StAnDataSet(torch.utils.data.Dataset):
    """`<root>/stand_small_{train,test}/*.npz` with `indices (2,nnz), values, solution, rhs` (data_set.py:142-214);
    vectors are zero-padded, no trivial equations are added, dof_max is the data set's 5166."""

    def __init__(self, stage: str, batch_size: int, shuffle: bool, root=ROOT, device=None) -> None:
        if stage not in ("train", "test"):
            raise AssertionError(f"Invalid stage {stage}")
        self.files = sorted(pathlib.Path(root).glob(f"stand_small_{stage}/*.npz"))
        if shuffle:
            random.shuffle(self.files)
        self.batch_size = batch_size
        self.dof_max = 5166
        self.device = _device(device)

    def __len__(self) -> int:
        return len(self.files) // self.batch_size

    def __getitem__(self, index: int):
        samples = []
        for bi in range(self.batch_size):
            with np.load(self.files[index * self.batch_size + bi]) as f:
                indices, values, solution, rhs = (f[k] for k in f.files)                  # data_set.py:186-188
            samples.append((indices[0], indices[1], values, solution, rhs))
        return _batch(samples, self.dof_max, 0.0, self.device)
