from .trainer import SupLoss, Trainer, train  # noqa: F401
