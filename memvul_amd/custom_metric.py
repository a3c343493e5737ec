"""``SiameseMeasureV1`` + ``find_best_thres`` + ``cal_f1`` — the PR-AUC "sufficient statistics" of the
hot path (reference: MemVul/custom_metric.py:9-97), on ndarrays instead of Python lists.

Same results as the reference, to the bit for the counts and thresholds:
  * thresholds are ``np.arange(0.5, 0.9, 0.01)`` exactly as the reference builds them (l.38) and a score
    is positive iff ``s >= thres`` (l.41);
  * F1 ties resolve to the LAST threshold (``>=`` at l.47), and if every F1 is 0 the last threshold wins;
  * ROC-AUC = ``auc(roc_curve(...))`` and AP = ``average_precision_score`` from sklearn (l.88-90), the
    reference's own dependency.
The 40 x N Python loop of the reference becomes one sort + 40 binary searches.
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional

import numpy as np

from .registry import Metric


def _prf(TP: int, FN: int, TN: int, FP: int) -> Dict[str, Any]:
    prec = pd = f = 0
    if TP + FN != 0:
        pd = TP / (TP + FN)
    if TP + FP != 0:
        prec = TP / (TP + FP)
    if pd + prec != 0:
        f = 2 * pd * prec / (pd + prec)
    return {"TP": TP, "FN": FN, "TN": TN, "FP": FP, "precision": prec, "recall": pd, "f1": f}


def cal_f1(test_label, pred) -> Dict[str, Any]:
    """custom_metric.py:9-32 (labels/predictions in {0,1})."""
    y = np.asarray(test_label).astype(np.int64)
    p = np.asarray(pred).astype(np.int64)
    TP = int(np.sum((y == 1) & (p == 1)))
    FN = int(np.sum((y == 1) & (p != 1)))
    TN = int(np.sum((y == 0) & (p == 0)))
    FP = int(np.sum((y == 0) & (p != 0)))
    return _prf(TP, FN, TN, FP)


def threshold_confusion_table(test_label, pred_score, interval=(0.5, 0.9)) -> np.ndarray:
    """int64 [T,4] = (TP, FN, TN, FP) per threshold of ``np.arange(lo, hi, 0.01)`` — additive over
    shards, so ranks may all-reduce it instead of gathering scores."""
    y = np.asarray(test_label).astype(np.int64)
    s = np.asarray(pred_score, dtype=np.float64)  # Python floats in the reference are doubles of the fp32 scores
    th = np.arange(interval[0], interval[1], 0.01)
    pos = np.sort(s[y == 1])
    neg = np.sort(s[y == 0])
    tp = len(pos) - np.searchsorted(pos, th, side="left")  # count of s >= thres
    fp = len(neg) - np.searchsorted(neg, th, side="left")
    return np.stack([tp, len(pos) - tp, len(neg) - fp, fp], 1).astype(np.int64)


def best_from_table(table: np.ndarray, interval=(0.5, 0.9)) -> Optional[Dict[str, Any]]:
    th = np.arange(interval[0], interval[1], 0.01)
    best_f1, best = 0, None
    for t, (TP, FN, TN, FP) in zip(th, table.tolist()):
        m = _prf(TP, FN, TN, FP)
        if m["f1"] >= best_f1:
            best_f1 = m["f1"]
            m["thres"] = t
            best = m
    return best


def find_best_thres(test_label, pred_score, interval=(0.5, 0.9)):
    """custom_metric.py:35-52."""
    return best_from_table(threshold_confusion_table(test_label, pred_score, interval), interval)


def roc_auc_ap(test_label, pred_score):
    from sklearn import metrics

    fpr, tpr, _ = metrics.roc_curve(test_label, pred_score, pos_label=1)
    return metrics.auc(fpr, tpr), metrics.average_precision_score(test_label, pred_score, pos_label=1)


def siamese_metrics(labels: np.ndarray, scores: np.ndarray) -> Dict[str, Any]:
    """What ``SiameseMeasureV1.get_metric(reset=True)`` returns (custom_metric.py:74-94) for the given
    accumulated ``(label, score)`` arrays."""
    out = {"precision": 0, "recall": 0, "f1": 0, "thres": 0, "auc": 0, "ave_precision_score": 0}
    if len(scores) == 0:
        return out
    out = find_best_thres(labels, scores, interval=(0.5, 0.9))
    out["auc"], out["ave_precision_score"] = roc_auc_ap(np.asarray(labels), np.asarray(scores, dtype=np.float64))
    return out


@Metric.register("siamese_measure_v1")
class SiameseMeasureV1(Metric):
    """Accumulates ``(label in {0,1}, P(same) of the best anchor)`` per issue report
    (custom_metric.py:63-72) in growing ndarrays."""

    def __init__(self, same_idx, thres=0.5) -> None:
        self._same_idx = same_idx
        self._thres = thres
        self._labels: List[np.ndarray] = []
        self._scores: List[np.ndarray] = []

    def __call__(self, predictions, metadata: List[Dict[str, Any]] = None, mask=None):
        p = np.asarray(predictions.detach().cpu().numpy() if hasattr(predictions, "detach") else predictions)
        lab = np.fromiter((0 if m["instance"][0]["label"] == "neg" else 1 for m in metadata), dtype=np.uint8, count=len(metadata))
        self._labels.append(lab)
        self._scores.append(p[: len(metadata), self._same_idx].astype(np.float32))

    def add_arrays(self, labels: np.ndarray, scores: np.ndarray):
        """Array-native accumulation used by the resident-corpus path."""
        self._labels.append(np.asarray(labels, np.uint8))
        self._scores.append(np.asarray(scores, np.float32))

    def arrays(self):
        if not self._scores:
            return np.zeros(0, np.uint8), np.zeros(0, np.float32)
        return np.concatenate(self._labels), np.concatenate(self._scores)

    def get_metric(self, reset: bool):
        metrics_pos = {"precision": 0, "recall": 0, "f1": 0, "thres": 0, "auc": 0, "ave_precision_score": 0}
        labels, scores = self.arrays()
        if len(scores) == 0:
            return metrics_pos
        if reset:
            metrics_pos = siamese_metrics(labels, scores)
            self.reset()
        return metrics_pos

    def reset(self) -> None:
        self._labels.clear()
        self._scores.clear()
