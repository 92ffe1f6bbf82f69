"""Training loop of `uibk/deep_preconditioning/train.py` on PyTorch-ROCm without DVC / dvclive.

Same pieces, same names: `_train_single_epoch` (inverse loss, Adam step per batch, train.py:40-63), `_validate`
(validation loss plus the PCG duration / iteration count of every validation system with the learned preconditioner,
train.py:66-109 -- here the call of the hot path at train.py:102-106 goes to the MI355X solver, with the system kept
sparse and `L` handed over as a factor instead of a dense `L L^T`), `EarlyStopping` (train.py:112-135) and `main`
(train.py:138-190; metrics go to `assets/metrics.csv` instead of dvclive, the checkpoint to
`assets/checkpoints/best.pt`, where `benchmark_suite.main()` looks for it).
"""

from __future__ import annotations

import csv
import pathlib
import random

import numpy as np
import torch
from torch.utils.data.dataset import random_split

from . import data_set as data_sets
from . import model as models
from .cg import preconditioned_conjugate_gradient
from .metrics import inverse_loss
from .model import lower_factor_csr
from .operators import CsrSystem, LLtMultiply

SEED: int = 69      # train.py:23


def _train_single_epoch(model, data_set, optimizer) -> float:
    """Average inverse loss over the training batches (train.py:40-63)."""
    model.train()
    running_loss = 0.0
    for index in range(len(data_set)):
        systems_tril, _, _, _ = data_set[index]
        preconditioners_tril = model(systems_tril)
        optimizer.zero_grad()
        loss = inverse_loss(systems_tril, preconditioners_tril)
        running_loss += loss.item()
        loss.backward()
        optimizer.step()
    return running_loss / max(len(data_set), 1)


def _system_from_tril(systems_tril, batch_index: int, original_size: int) -> CsrSystem:
    """The mirrored fp64 system of one sample (train.py:93-95) as CSR on the device, without the dense detour."""
    from .io import coo_to_csr_device
    idx = systems_tril.indices.long()
    keep = (idx[:, 0] == batch_index) & (idx[:, 1] < original_size) & (idx[:, 2] < original_size)
    r, c, v = idx[keep, 1], idx[keep, 2], systems_tril.features[keep, 0].to(torch.float64)
    off = r != c
    rowptr, col, val = coo_to_csr_device(torch.cat((r, c[off])), torch.cat((c, r[off])), torch.cat((v, v[off])),
                                         original_size, device=systems_tril.features.device)
    return CsrSystem(rowptr, col, val, original_size)


@torch.no_grad()
def _validate(model, data_set) -> tuple[float, float, float]:
    """(validation loss, mean PCG duration, mean PCG iterations) with M = L L^T from the model (train.py:66-109)."""
    model.eval()
    val_losses, durations, iterations = [], [], []
    for index in range(len(data_set)):
        systems_tril, _, right_hand_sides, original_sizes = data_set[index]
        preconditioners_tril = model(systems_tril)
        val_losses.append(inverse_loss(systems_tril, preconditioners_tril).item())
        for batch_index in range(systems_tril.batch_size):
            n = int(original_sizes[batch_index])
            system = _system_from_tril(systems_tril, batch_index, n)
            rhs = right_hand_sides[batch_index, :n].squeeze().to(torch.float64)
            factor = LLtMultiply(lower_factor_csr(preconditioners_tril, batch_index, n))
            duration, n_iterations, _ = preconditioned_conjugate_gradient(system, rhs, M=factor)   # train.py:102-106
            durations.append(duration)
            iterations.append(n_iterations)
            system.close()
    return float(np.mean(val_losses)), float(np.mean(durations)), float(np.mean(iterations))


class EarlyStopping:
    """Stop when the validation loss has not improved for `patience` epochs (train.py:112-135)."""

    def __init__(self, patience: int) -> None:
        self.patience = patience
        self.local_min = float("inf")
        self.counter = 0

    def __call__(self, val_loss: float) -> bool:
        if val_loss > self.local_min:
            self.counter += 1
        else:
            self.local_min = val_loss
            self.counter = 0
        return self.counter >= self.patience


def main(params_path="params.yaml", root=None, max_epochs: int | None = None, assets=pathlib.Path("assets")) -> dict:
    """train.py:138-190.  `max_epochs` bounds the loop for tests; the reference stops on `EarlyStopping` only."""
    import yaml
    assert torch.cuda.is_available(), "CUDA not available"
    random.seed(SEED)
    torch.manual_seed(SEED)
    params = yaml.safe_load(pathlib.Path(params_path).read_text())
    kwargs = {} if root is None else {"root": pathlib.Path(root)}
    data = getattr(data_sets, params["data"])(stage="train", batch_size=params["batch_size"], shuffle=True, **kwargs)
    train_data, val_data = random_split(data, lengths=[0.95, 0.05])
    model = getattr(models, params["model"])(params["channels"]).to("cuda")
    optimizer = torch.optim.Adam(model.parameters(), lr=params["learning_rate"])
    early_stopping = EarlyStopping(patience=params["patience"])
    checkpoints = pathlib.Path(assets) / "checkpoints"
    checkpoints.mkdir(parents=True, exist_ok=True)
    history = {"train/loss/inverse": [], "val/loss/inverse": [], "val/metric/durations": [], "val/metric/iterations": []}
    epoch = 0
    while max_epochs is None or epoch < max_epochs:
        train_loss = _train_single_epoch(model, train_data, optimizer)
        val_loss, durations, iterations = _validate(model, val_data)
        for key, value in zip(history, (train_loss, val_loss, durations, iterations)):
            history[key].append(value)
        if early_stopping(val_loss):
            break
        torch.save(model.state_dict(), checkpoints / "best.pt")                       # train.py:183
        epoch += 1
    with (pathlib.Path(assets) / "metrics.csv").open("w", newline="") as f:
        writer = csv.writer(f)
        writer.writerow(["step"] + list(history))
        for step, row in enumerate(zip(*history.values())):
            writer.writerow([step] + list(row))
    return history


if __name__ == "__main__":
    main()
