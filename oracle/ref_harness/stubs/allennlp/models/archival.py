"""allennlp/models/archival.py (subset): load_archive on a model.tar.gz or an already-extracted directory."""
import os
import tarfile
import tempfile
from typing import NamedTuple

from allennlp.common import Params
from allennlp.data import DatasetReader

from .model import Model

CONFIG_NAME = "config.json"
_WEIGHTS_NAME = "weights.th"


class Archive(NamedTuple):
    model: Model
    config: Params
    dataset_reader: DatasetReader
    validation_dataset_reader: DatasetReader


def load_archive(archive_file, cuda_device: int = -1, overrides="", weights_file: str = None) -> Archive:
    resolved = str(archive_file)
    tempdir = None
    if os.path.isdir(resolved):
        serialization_dir = resolved
    else:
        tempdir = tempfile.mkdtemp(prefix="mvref_archive")
        with tarfile.open(resolved, "r:gz") as archive:
            archive.extractall(tempdir)
        serialization_dir = tempdir
    weights_path = weights_file or os.path.join(serialization_dir, _WEIGHTS_NAME)
    config = Params.from_file(os.path.join(serialization_dir, CONFIG_NAME), overrides)
    dataset_reader_params = config.get("dataset_reader")
    validation_dataset_reader_params = config.get("validation_dataset_reader", None)
    if validation_dataset_reader_params is None:
        validation_dataset_reader_params = dataset_reader_params.duplicate()
    dataset_reader = DatasetReader.from_params(dataset_reader_params.duplicate(), serialization_dir=serialization_dir)
    validation_dataset_reader = DatasetReader.from_params(validation_dataset_reader_params.duplicate(), serialization_dir=serialization_dir)
    model = Model.load(config.duplicate(), weights_file=weights_path, serialization_dir=serialization_dir, cuda_device=cuda_device)
    return Archive(model=model, config=config, dataset_reader=dataset_reader, validation_dataset_reader=validation_dataset_reader)
