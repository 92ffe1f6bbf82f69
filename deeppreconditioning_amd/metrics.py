"""Losses of `uibk/deep_preconditioning/metrics.py` that touch the hot-path operators.

`frobenius_loss` runs on the HIP batched COO SpMV and is differentiable with respect to the network output
(`utils._SparseMatvec`); `inverse_loss` is the dense O(N^3) restatement of the reference (the loss actually
trained, train.py:59), plain torch ops on the GPU, kept for completeness at the reference's sizes.
"""

from __future__ import annotations

import torch

from .utils import sparse_matvec_mul


def frobenius_loss(lower_triangular, solution: torch.Tensor, right_hand_side: torch.Tensor) -> torch.Tensor:
    """sum_b || L_b L_b^T x_b - b_b ||_2  (metrics.py:13-31, arXiv:2305.16432 eq. 11)."""
    interim = sparse_matvec_mul(lower_triangular, solution, transpose=True)       # metrics.py:28
    interim = sparse_matvec_mul(lower_triangular, interim, transpose=False)       # metrics.py:29
    return torch.linalg.vector_norm(interim - right_hand_side, ord=2, dim=1).sum()


def inverse_loss(systems_tril, preconditioners_tril) -> torch.Tensor:
    """mean_b || L_b L_b^T A_b - I ||_F with dense matrices (metrics.py:34-55)."""
    pre = preconditioners_tril.dense()[:, 0]
    pre = torch.matmul(pre, pre.transpose(-1, -2))
    systems = systems_tril.dense()[:, 0]
    systems = systems + torch.tril(systems, -1).transpose(-1, -2)
    prod = torch.matmul(pre, systems)
    eye = torch.eye(systems.shape[1], device=prod.device).unsqueeze(0).expand((systems.shape[0], -1, -1))
    return torch.linalg.matrix_norm(prod - eye).mean()


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
