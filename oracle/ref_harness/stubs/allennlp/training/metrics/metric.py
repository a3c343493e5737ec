from typing import Iterable

import torch

from allennlp.common import Registrable


class Metric(Registrable):
    def __call__(self, predictions, gold_labels, mask=None):
        raise NotImplementedError

    def get_metric(self, reset: bool):
        raise NotImplementedError

    def reset(self) -> None:
        raise NotImplementedError

    @staticmethod
    def detach_tensors(*tensors: torch.Tensor) -> Iterable[torch.Tensor]:
        return (x.detach() if isinstance(x, torch.Tensor) else x for x in tensors)
