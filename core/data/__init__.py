from .create_dataset import create_dataloader  # noqa: F401
