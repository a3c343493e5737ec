"""Generate tests/golden/ref/ by running the REFERENCE'S OWN CODE (build container only; /root/reference + the
tests-only AllenNLP stand-in under oracle/ref_harness/stubs/).  The outputs are committed:

    CWE_anchor_golden_project.json, test_project.json, xxxCVE_dict.json, vocab.txt, config.json   fixture inputs
    ref_predictions.jsonl   the predictions file predict_memory.test_siamese wrote (predict_memory.py:103-110)
    ref_metrics.json        ModelMemory.get_metrics(reset=True) through AllenNLP evaluate (model_memory.py:194-217)
    ref_metric_all.json     predict_memory.cal_metrics at the run's own threshold (predict_memory.py:159-197)
    ref_tensors.npz         _golden_instances_embeddings and every batch's `probs` (model_memory.py:105-115, 135-143)
    ref_reader.json         token ids / labels / metadata of every Instance ReaderMemory emitted, in order
    ref_stats_cases.json    cal_f1 / find_best_thres / SiameseMeasureV1 / model_measure called on seeded vectors
    meta.json               versions, seeds, order of issue reports and anchors

    python tests/golden/make_ref_golden.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle.ref_harness import run_reference  # noqa: E402

# tests/golden/ref12: the 12-layer TRAINED-LIKE regime (stable LayerNorm outlier dimensions, peaked attention,
# matcher x29 -> |logit| ~ 3) through the reference's own code; adds `logits` (a hook on ModelMemory._projector) to the tensors
REF12 = dict(layers=12, n_irs=20, n_anchors=6, seed=12, weight_kwargs=dict(qk_scale=2.0, match_scale=29.0, trained_like=True),
             structured=False, long_texts=True)

if __name__ == "__main__":
    which = sys.argv[1:] or ["ref", "ref12"]
    if "ref" in which:
        res = run_reference.generate(os.path.join(ROOT, "tests", "golden", "ref"))
        print({k: res["metrics"][k] for k in ("accuracy", "f1-score", "s_f1-score", "s_thres", "s_auc")})
    if "ref12" in which:
        res = run_reference.generate(os.path.join(ROOT, "tests", "golden", "ref12"), **REF12)
        import numpy as np
        print("ref12: max |logit|", float(np.abs(res["logits"]).max()), "probs", res["probs"].shape,
              {k: res["metrics"][k] for k in ("accuracy", "s_f1-score", "s_thres", "s_auc")})
