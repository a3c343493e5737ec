from .dataset_reader import DatasetReader  # noqa: F401
