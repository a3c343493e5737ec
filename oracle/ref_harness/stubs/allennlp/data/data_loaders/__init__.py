from .data_loader import DataLoader, MultiProcessDataLoader, TensorDict  # noqa: F401
