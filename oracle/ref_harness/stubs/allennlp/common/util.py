"""allennlp/common/util.py (subset)."""
import importlib
import logging
import os
import pkgutil
import random
import sys

import numpy
import torch

logger = logging.getLogger(__name__)


def int_to_device(device):
    if isinstance(device, torch.device):
        return device
    return torch.device("cpu") if device is None or int(device) < 0 else torch.device(int(device))


def prepare_environment(params):
    seed = params.pop_int("random_seed", 13370)
    numpy_seed = params.pop_int("numpy_seed", 1337)
    torch_seed = params.pop_int("pytorch_seed", 133)
    if seed is not None:
        random.seed(seed)
    if numpy_seed is not None:
        numpy.random.seed(numpy_seed)
    if torch_seed is not None:
        torch.manual_seed(torch_seed)


def import_module_and_submodules(package_name: str, skipped=None) -> None:
    """Imports a package and every sub-module.  Stand-in difference: a sub-module that needs parts of AllenNLP this
    stub does not carry (MemVul/custom_trainer.py, callbacks.py: the training stack — off the predict_memory.py path)
    is logged and skipped instead of failing the whole import."""
    importlib.invalidate_caches()
    module = importlib.import_module(package_name)
    path = getattr(module, "__path__", [])
    for module_finder, name, _ in pkgutil.walk_packages(list(path)):
        sub = f"{package_name}.{name}"
        try:
            importlib.import_module(sub)
        except (ImportError, AttributeError) as e:
            logger.warning("allennlp stub: skipped %s (%s)", sub, e)
            if skipped is not None:
                skipped.append(sub)


def sanitize(x):
    """allennlp.common.util.sanitize: make a structure JSON-serialisable."""
    if isinstance(x, (str, float, int, bool)) or x is None:
        return x
    if isinstance(x, torch.Tensor):
        return x.cpu().tolist()
    if isinstance(x, numpy.ndarray):
        return x.tolist()
    if isinstance(x, numpy.number):
        return x.item()
    if isinstance(x, dict):
        return {k: sanitize(v) for k, v in x.items()}
    if isinstance(x, (list, tuple, set)):
        return [sanitize(v) for v in x]
    if hasattr(x, "to_json"):
        return x.to_json()
    raise ValueError(f"Cannot sanitize {x} of type {type(x)}")


def dump_metrics(file_path, metrics, log=False):
    import json

    s = json.dumps(metrics, indent=2)
    if file_path:
        with open(file_path, "w") as f:
            f.write(s)
    if log:
        logger.info("Metrics: %s", s)
