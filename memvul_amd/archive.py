"""``load_archive`` for the hot path (reference call: predict_memory.py:62-73).

An AllenNLP archive ``model.tar.gz`` (or an already-extracted directory) holds ``config.json`` (the
training config, MemVul/config_memory.json rendered), ``weights.th`` (``torch.save(model.state_dict())``)
and ``vocabulary/`` (``labels.txt`` -> index of "same", model_memory.py:61).  The archive is the weight
source of the engine; torch is imported only here, to deserialise ``weights.th`` (or ``.safetensors`` /
``.npz`` for torch-free deployments).
"""
from __future__ import annotations

import os
import tarfile
import tempfile
from dataclasses import dataclass
from typing import Any, Dict, Optional

import numpy as np

from . import params as _params
from .registry import DatasetReader, Model, Vocabulary
from . import model_memory as _mm  # noqa: F401  (registers model_memory + embedders)
from . import reader_memory as _rm  # noqa: F401  (registers reader_memory)
from . import model_single as _ms  # noqa: F401  (registers model_single)
from . import reader_single as _rs  # noqa: F401  (registers reader_single)


@dataclass
class Archive:
    model: Any
    config: Dict[str, Any]
    dataset_reader: Any
    validation_dataset_reader: Any


def read_state_dict(path: str) -> Dict[str, np.ndarray]:
    if path.endswith(".npz"):
        with np.load(path) as z:
            return {k: z[k] for k in z.files}
    if path.endswith(".safetensors"):
        from safetensors.numpy import load_file

        return load_file(path)
    import torch

    sd = torch.load(path, map_location="cpu", weights_only=True)
    return {k: v.detach().to(torch.float32).numpy() if v.is_floating_point() else v.numpy() for k, v in sd.items()}


def _build_reader(cfg: Optional[Dict[str, Any]]):
    return None if cfg is None else DatasetReader.from_params(cfg)


def load_archive(archive_file: str, weights_file: Optional[str] = None, cuda_device: int = -1, overrides: Any = "",
                 engine_options: Optional[Dict[str, Any]] = None) -> Archive:
    tmp = None
    root = archive_file
    if os.path.isfile(archive_file):
        tmp = tempfile.TemporaryDirectory(prefix="memvul_archive_")
        with tarfile.open(archive_file, "r:*") as tf:
            tf.extractall(tmp.name, filter="data")  # no absolute paths / links / traversal out of the temp dir
        root = tmp.name
    try:
        config = _params.with_overrides(_params.load_config(os.path.join(root, "config.json")), overrides)
        vocab = Vocabulary.from_files(os.path.join(root, "vocabulary"))
        mcfg = dict(config["model"])
        if not str(mcfg.get("device", "cpu")).startswith("cuda") and cuda_device is not None and cuda_device >= 0:
            mcfg["device"] = f"cuda:{cuda_device}"  # AllenNLP moves the model to cuda_device (predict_memory.py:65)
        wpath = weights_file
        if wpath is None:
            for cand in ("weights.th", "weights.safetensors", "weights.npz"):
                if os.path.exists(os.path.join(root, cand)):
                    wpath = os.path.join(root, cand)
                    break
        if wpath is None:
            raise FileNotFoundError(f"no weights.th in {archive_file}")
        # the state dict is read (torch imported, CPU only) BEFORE the model creates its engine: torch, when it is
        # used at all, is always loaded ahead of libmemvul_hip.so (tests/test_gpu_parity.py: supported load order)
        state = read_state_dict(wpath)
        model = Model.from_params(mcfg, vocab=vocab, engine_options=engine_options)
        model.load_state_dict(state)
        reader = _build_reader(config.get("dataset_reader"))
        vreader = _build_reader(config.get("validation_dataset_reader")) or reader
        return Archive(model=model, config=config, dataset_reader=reader, validation_dataset_reader=vreader)
    finally:
        if tmp is not None:
            tmp.cleanup()
