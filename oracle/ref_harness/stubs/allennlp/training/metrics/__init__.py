"""allennlp/training/metrics (subset): CategoricalAccuracy (top-1, no tie-break) and FBetaMeasure (average None /
"weighted"), restated from AllenNLP 2.4.0 categorical_accuracy.py / fbeta_measure.py."""
from typing import List, Optional, Union

import torch

from . import metric  # noqa: F401
from .metric import Metric


@Metric.register("categorical_accuracy")
class CategoricalAccuracy(Metric):
    def __init__(self, top_k: int = 1, tie_break: bool = False) -> None:
        assert top_k == 1 and not tie_break
        self.correct_count = 0.0
        self.total_count = 0.0

    def __call__(self, predictions: torch.Tensor, gold_labels: torch.Tensor, mask: Optional[torch.BoolTensor] = None):
        predictions, gold_labels, mask = self.detach_tensors(predictions, gold_labels, mask)
        num_classes = predictions.size(-1)
        if gold_labels.dim() != predictions.dim() - 1:
            raise ValueError("gold_labels must have dimension == predictions.size() - 1")
        if (gold_labels >= num_classes).any():
            raise ValueError("A gold label passed to Categorical Accuracy contains an id >= {}".format(num_classes))
        predictions = predictions.view((-1, num_classes))
        gold_labels = gold_labels.view(-1).long()
        top_k = predictions.max(-1)[1].unsqueeze(-1)
        correct = top_k.eq(gold_labels.unsqueeze(-1)).float()
        if mask is not None:
            correct *= mask.view(-1, 1)
            _total = mask.sum()
        else:
            _total = torch.tensor(gold_labels.numel())
        self.correct_count += correct.sum().item()
        self.total_count += _total.item()

    def get_metric(self, reset: bool = False) -> float:
        accuracy = float(self.correct_count) / float(self.total_count) if self.total_count > 1e-12 else 0.0
        if reset:
            self.reset()
        return accuracy

    def reset(self):
        self.correct_count = 0.0
        self.total_count = 0.0


def _prf_divide(numerator, denominator):
    result = numerator / denominator
    mask = denominator == 0.0
    if not mask.any():
        return result
    result[mask] = 0.0
    return result


@Metric.register("fbeta")
class FBetaMeasure(Metric):
    def __init__(self, beta: float = 1.0, average: str = None, labels: List[int] = None) -> None:
        average_options = {None, "micro", "macro", "weighted"}
        if average not in average_options:
            raise ValueError(f"`average` has to be one of {average_options}.")
        if beta <= 0:
            raise ValueError("`beta` should be >0 in the F-beta score.")
        if labels is not None and len(labels) == 0:
            raise ValueError("`labels` cannot be an empty list.")
        self._beta = beta
        self._average = average
        self._labels = labels
        self._true_positive_sum: Union[None, torch.Tensor] = None
        self._total_sum: Union[None, torch.Tensor] = None
        self._pred_sum: Union[None, torch.Tensor] = None
        self._true_sum: Union[None, torch.Tensor] = None

    def __call__(self, predictions: torch.Tensor, gold_labels: torch.Tensor, mask: Optional[torch.BoolTensor] = None):
        predictions, gold_labels, mask = self.detach_tensors(predictions, gold_labels, mask)
        num_classes = predictions.size(-1)
        if (gold_labels >= num_classes).any():
            raise ValueError("A gold label passed to FBetaMeasure contains an id >= {}".format(num_classes))
        if self._true_positive_sum is None:
            self._true_positive_sum = torch.zeros(num_classes, device=predictions.device)
            self._true_sum = torch.zeros(num_classes, device=predictions.device)
            self._pred_sum = torch.zeros(num_classes, device=predictions.device)
            self._total_sum = torch.zeros(num_classes, device=predictions.device)
        if mask is None:
            mask = torch.ones_like(gold_labels).bool()
        gold_labels = gold_labels.float()
        argmax_predictions = predictions.max(dim=-1)[1].float()
        true_positives = (gold_labels == argmax_predictions) & mask
        true_positives_bins = gold_labels[true_positives]
        if true_positives_bins.shape[0] == 0:
            true_positive_sum = torch.zeros(num_classes, device=predictions.device)
        else:
            true_positive_sum = torch.bincount(true_positives_bins.long(), minlength=num_classes).float()
        pred_bins = argmax_predictions[mask].long()
        if pred_bins.shape[0] != 0:
            pred_sum = torch.bincount(pred_bins, minlength=num_classes).float()
        else:
            pred_sum = torch.zeros(num_classes, device=predictions.device)
        gold_labels_bins = gold_labels[mask].long()
        if gold_labels.shape[0] != 0:
            true_sum = torch.bincount(gold_labels_bins, minlength=num_classes).float()
        else:
            true_sum = torch.zeros(num_classes, device=predictions.device)
        self._true_positive_sum += true_positive_sum
        self._pred_sum += pred_sum
        self._true_sum += true_sum
        self._total_sum += mask.sum().to(torch.float)

    def get_metric(self, reset: bool = False):
        if self._true_positive_sum is None:
            raise RuntimeError("You never call this metric before.")
        tp_sum, pred_sum, true_sum = self._true_positive_sum, self._pred_sum, self._true_sum
        if self._labels is not None:
            tp_sum, pred_sum, true_sum = tp_sum[self._labels], pred_sum[self._labels], true_sum[self._labels]
        if self._average == "micro":
            tp_sum, pred_sum, true_sum = tp_sum.sum(), pred_sum.sum(), true_sum.sum()
        beta2 = self._beta ** 2
        precision = _prf_divide(tp_sum, pred_sum)
        recall = _prf_divide(tp_sum, true_sum)
        fscore = (1 + beta2) * precision * recall / (beta2 * precision + recall)
        fscore[tp_sum == 0] = 0.0
        if self._average == "macro":
            precision, recall, fscore = precision.mean(), recall.mean(), fscore.mean()
        elif self._average == "weighted":
            weights = true_sum
            weights_sum = true_sum.sum()
            precision = _prf_divide((weights * precision).sum(), weights_sum)
            recall = _prf_divide((weights * recall).sum(), weights_sum)
            fscore = _prf_divide((weights * fscore).sum(), weights_sum)
        if reset:
            self.reset()
        if self._average is None:
            return {"precision": precision.tolist(), "recall": recall.tolist(), "fscore": fscore.tolist()}
        return {"precision": precision.item(), "recall": recall.item(), "fscore": fscore.item()}

    def reset(self) -> None:
        self._true_positive_sum = None
        self._pred_sum = None
        self._true_sum = None
        self._total_sum = None


class F1Measure(FBetaMeasure):  # import surface only (model_memory.py:19)
    pass
