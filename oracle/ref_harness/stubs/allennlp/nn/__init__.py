"""allennlp/nn (subset): Activation registry, initializer / regularizer applicators (empty on the reference's configs)."""
import torch

from allennlp.common import Registrable
from . import util  # noqa: F401


class Activation(torch.nn.Module, Registrable):
    def __call__(self, tensor):
        raise NotImplementedError


Registrable._registry[Activation] = {
    "linear": (lambda: (lambda x: x), None), "relu": (torch.nn.ReLU, None), "tanh": (torch.nn.Tanh, None),
    "gelu": (torch.nn.GELU, None), "sigmoid": (torch.nn.Sigmoid, None),
}


class InitializerApplicator(Registrable):
    def __init__(self, regexes=None, prevent_regexes=None) -> None:
        self._initializers = regexes or []

    def __call__(self, module: torch.nn.Module) -> None:
        assert not self._initializers, "allennlp stub: only the empty (default) initializer is carried"


class RegularizerApplicator(Registrable):
    def __init__(self, regexes=None) -> None:
        self._regularizers = regexes or []

    def __call__(self, module: torch.nn.Module):
        return torch.tensor(0.0)
