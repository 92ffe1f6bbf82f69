"""Dense checkers of the sparse losses (test infrastructure, not product)."""
import torch


def inverse_loss_dense(systems_tril, preconditioners_tril) -> torch.Tensor:
    """mean_b || L_b L_b^T A_b - I ||_F with dense N x N matrices, the form of the reference's metrics.py:34-55: O(N^3); what
    `deeppreconditioning_amd.metrics.inverse_loss` (panels of columns on the HIP kernels) is checked against."""
    factor = preconditioners_tril.dense()[:, 0]
    m = factor @ factor.transpose(-1, -2)
    a = systems_tril.dense()[:, 0]
    a = a + torch.tril(a, -1).transpose(-1, -2)
    residual = m @ a - torch.eye(a.shape[1], device=a.device).unsqueeze(0)
    return torch.linalg.matrix_norm(residual).mean()
