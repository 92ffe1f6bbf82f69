"""The dense inverse loss of metrics.py:34-55 against values the reference itself returned (fixtures `metrics/*`, produced by
tests/golden/make_golden.py on a duck-typed batch).  These are plain torch restatements: they run on the CPU here."""
import numpy as np
import torch

from deeppreconditioning_amd import metrics
from deeppreconditioning_amd.utils import SparseBatch


def _batches(golden):
    sys_low = torch.from_numpy(golden["metrics/systems_tril"])
    pre_low = torch.from_numpy(golden["metrics/preconditioners_tril"])
    to_sparse = lambda d: SparseBatch.from_dense(d.permute(0, 2, 3, 1))   # noqa: E731  (batch, H, W, channels)
    return to_sparse(sys_low), to_sparse(pre_low)


def test_dense_losses_match_the_reference(golden):
    systems, pre = _batches(golden)
    assert float(metrics.inverse_loss_dense(systems, pre)) == np.float32(golden["metrics/inverse_loss"])
    assert not hasattr(metrics, "hutchinson_trace") and not hasattr(metrics, "condition_loss")     # outside the path


def _sparse_batches(golden):
    to_sparse = lambda d: SparseBatch.from_dense(torch.from_numpy(d).permute(0, 2, 3, 1))   # noqa: E731
    return to_sparse(golden["metrics_sparse/systems_tril"]), to_sparse(golden["metrics_sparse/preconditioners_tril"])


def test_dense_inverse_loss_on_the_sparse_fixture(golden):
    systems, pre = _sparse_batches(golden)
    np.testing.assert_allclose(float(metrics.inverse_loss_dense(systems, pre)), golden["metrics_sparse/inverse_loss"], rtol=1e-6)
    with np.testing.assert_raises(ValueError):                     # the sparse form is a GPU path: no silent CPU fallback
        metrics.inverse_loss(systems, pre)
