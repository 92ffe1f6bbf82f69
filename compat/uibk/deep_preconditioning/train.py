"""`uibk.deep_preconditioning.train` (train.py:40-190) on PyTorch-ROCm, without DVC / dvclive."""
from deeppreconditioning_amd.train import EarlyStopping, _train_single_epoch, _validate, main  # noqa: F401


if __name__ == "__main__":
    main()
