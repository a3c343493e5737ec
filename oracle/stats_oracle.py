"""CPU ORACLE (test infrastructure — NOT the product path): scalar restatement of the
reference's decision / PR-AUC statistics.

* ``cal_f1``           MemVul/custom_metric.py:9-32
* ``find_best_thres``  MemVul/custom_metric.py:35-52 — thresholds ``np.arange(0.5, 0.9, 0.01)``
                       (40 values, float accumulation of arange kept), ``s >= thres`` -> 1,
                       ties in F1 resolved to the LAST threshold (``>=`` at l.47).
* ``siamese_get_metric`` MemVul/custom_metric.py:74-94 — best-threshold metrics + ROC-AUC
                       (``roc_curve``+``auc``) + ``average_precision_score`` (sklearn).
* ``model_measure``    predict_memory.py:117-156.
* ``cal_metrics_records`` predict_memory.py:159-197 on in-memory records: per-IR score =
                       max over anchors of P(same) (l.170-171), ``score >= thres`` -> pos
                       (l.174-177), label "neg" -> 0 else 1 (l.182-183).

Written as plain loops on purpose (it is the checker for the vectorised product code in
``memvul_amd/custom_metric.py``); use on small N only.
"""
from __future__ import annotations

import numpy as np
from sklearn import metrics


def cal_f1(test_label, pred):
    TP = FN = TN = FP = 0
    for i in range(len(test_label)):
        if pred[i] == test_label[i] == 1:
            TP += 1
        elif test_label[i] == 1 and pred[i] != test_label[i]:
            FN += 1
        elif pred[i] == test_label[i] == 0:
            TN += 1
        elif test_label[i] == 0 and pred[i] != test_label[i]:
            FP += 1
    prec = pd = f_measure = 0
    if TP + FN != 0:
        pd = TP / (TP + FN)
    if TP + FP != 0:
        prec = TP / (TP + FP)
    if pd + prec != 0:
        f_measure = 2 * pd * prec / (pd + prec)
    return {"TP": TP, "FN": FN, "TN": TN, "FP": FP, "precision": prec, "recall": pd, "f1": f_measure}


def find_best_thres(test_label, pred_score, interval=(0.5, 0.9)):
    best_f1 = 0
    best_metric = None
    for thres in np.arange(interval[0], interval[1], 0.01):
        pred = [1 if s >= thres else 0 for s in pred_score]
        m = cal_f1(test_label, pred)
        if m["f1"] >= best_f1:
            best_f1 = m["f1"]
            m["thres"] = thres
            best_metric = m
    return best_metric


def siamese_get_metric(labels, scores):
    out = {"precision": 0, "recall": 0, "f1": 0, "thres": 0, "auc": 0, "ave_precision_score": 0}
    if len(scores) == 0:
        return out
    out = find_best_thres(list(labels), list(scores), interval=(0.5, 0.9))
    fpr, tpr, _ = metrics.roc_curve(labels, scores, pos_label=1)
    out["auc"] = metrics.auc(fpr, tpr)
    out["ave_precision_score"] = metrics.average_precision_score(labels, scores, pos_label=1)
    return out


def model_measure(test_label, pred, pred_score):
    TP = FN = TN = FP = 0
    pd = prec = f_measure = 0
    for i in range(len(test_label)):
        if pred[i] == test_label[i] == 1:
            TP += 1
        elif test_label[i] == 1 and pred[i] != test_label[i]:
            FN += 1
        elif pred[i] == test_label[i] == 0:
            TN += 1
        elif test_label[i] == 0 and pred[i] != test_label[i]:
            FP += 1
    if TP + FN != 0:
        pd = TP / (TP + FN)
    if TP + FP != 0:
        prec = TP / (TP + FP)
    if pd + prec != 0:
        f_measure = 2 * pd * prec / (pd + prec)
    fpr, tpr, _ = metrics.roc_curve(test_label, pred_score, pos_label=1)
    auc = metrics.auc(fpr, tpr)
    ap = metrics.average_precision_score(test_label, pred_score, pos_label=1)
    return {"TP": TP, "FN": FN, "TN": TN, "FP": FP, "pd&recall": pd, "prec": prec, "f1": f_measure, "ap": ap, "auc": auc}


def cal_metrics_records(records, thres=0.5):
    """records: list of {"Issue_Url", "label", "predict": {cwe: P(same)}} (model_memory.py:186-189)."""
    label, pred, score = [], [], []
    for s in records:
        vote = float(np.max(list(s["predict"].values())))
        score.append(vote)
        pred.append(1 if vote >= thres else 0)
        label.append(0 if s["label"] == "neg" else 1)
    m = model_measure(label, pred, score)
    m["thres"] = thres
    return m
