"""MemVul-m (model_single / reader_single / predict_single, reference MemVul/model_single.py, predict_single.py): the
same encoder hot path followed by a 512 x 2 classifier.  CPU: plumbing with the oracle-backed engine; GPU: the HIP
engine against that run."""
import json
import os

import numpy as np
import pytest

import plumbing_util as pu
from memvul_amd import model_single, predict_single, synth
from oracle import memvul_oracle as orc
from oracle import stats_oracle as so


def make_single_fixture(n_irs=36, layers=2, seed=11):
    root, arch, golden, test_path, w, dims = pu.make_fixture(n_irs=n_irs, layers=layers, seed=seed)
    rng = np.random.default_rng(seed)
    sd = {k: v for k, v in w.items() if not k.startswith("_projector")}
    sd[model_single.KEY_HEAD_W] = w[synth.KEY_HEAD_W]
    sd[model_single.KEY_HEAD_B] = w[synth.KEY_HEAD_B]
    sd[model_single.KEY_CLS_W] = (rng.standard_normal((2, 512)) * 0.3).astype(np.float32)
    np.savez(os.path.join(arch, "weights.npz"), **sd)
    cfg = json.load(open(os.path.join(arch, "config.json")))
    cfg["dataset_reader"] = {"type": "reader_single", "sample_neg": 0.01, "train_iter": 1, "target": "Security_Issue_Full",
                             "tokenizer": cfg["dataset_reader"]["tokenizer"], "token_indexers": cfg["dataset_reader"]["token_indexers"]}
    cfg["model"] = {"type": "model_single", "label_namespace": "class_labels", "dropout": 0.1, "device": "cuda:0",
                    "PTM": "bert-base-uncased", "text_field_embedder": cfg["model"]["text_field_embedder"]}
    json.dump(cfg, open(os.path.join(arch, "config.json"), "w"))
    open(os.path.join(arch, "vocabulary", "class_labels.txt"), "w").write("neg\npos\n")
    return root, arch, test_path, sd


def _run(fx, tag):
    root, arch, test_path, sd = fx
    out_metric = os.path.join(root, "test_results", f"{tag}_metric.json")
    out_results = os.path.join(root, "test_results", f"{tag}_result.json")
    metrics = predict_single.test(archive_file=arch, input_file=test_path, test_config={"model": {"device": "cuda:0"}},
                                  output_file=out_metric, predictions_output_file=out_results, batch_size=16, cuda_device=0,
                                  engine_options=dict(max_tokens=16 * 256, max_batch=16, max_anchors=4))
    records = []
    for line in open(out_results):
        records.extend(json.loads(line))
    return metrics, records


def _check(fx, metrics, records, tag):
    root = fx[0]
    recs_in = json.load(open(fx[2]))
    assert len(records) == len(recs_in)
    for r in records:
        assert set(r) == {"Issue_Url", "label", "predict", "prob"} and r["predict"] in ("pos", "neg") and 0.0 <= r["prob"] <= 1.0
        assert (r["predict"] == "pos") == (r["prob"] > 0.5) or abs(r["prob"] - 0.5) < 1e-6
    m = predict_single.cal_metrics(f"{tag}_result", data_path=root)
    lab = [1 if r["label"] == "pos" else 0 for r in records]
    pred = [1 if r["predict"] == "pos" else 0 for r in records]
    ref = so.model_measure(lab, pred, [r["prob"] for r in records])
    for k in ("TP", "FN", "TN", "FP", "f1", "auc", "ap"):
        assert m[k] == pytest.approx(ref[k], abs=1e-12), k
    assert 0.0 <= metrics["accuracy"] <= 1.0 and "pos_f1-score" in metrics and "neg_recall" in metrics
    assert metrics["accuracy"] == pytest.approx(np.mean(np.array(lab) == np.array(pred)), abs=1e-12)


def test_single_plumbing_cpu_with_oracle_engine(monkeypatch):
    fx = make_single_fixture()
    monkeypatch.setattr(model_single, "Engine", pu.OracleEngine)
    metrics, records = _run(fx, "oracle")
    _check(fx, metrics, records, "oracle")


def test_single_head_matches_the_oracle_restatement():
    rng = np.random.default_rng(0)
    u = np.maximum(rng.standard_normal((9, 512)), 0).astype(np.float32)
    wc = (rng.standard_normal((2, 512)) * 0.2).astype(np.float32)
    m = model_single.ModelSingle.__new__(model_single.ModelSingle)
    m._cls_w = wc
    probs, logits = m.classify(u)
    lg, p = orc.classify_single(u, wc)
    assert np.abs(logits - lg).max() < 1e-5 and np.abs(probs - p).max() < 1e-6


@pytest.mark.gpu
def test_single_gpu_matches_oracle_run(monkeypatch):
    fx = make_single_fixture()
    metrics, records = _run(fx, "hip")
    _check(fx, metrics, records, "hip")
    monkeypatch.setattr(model_single, "Engine", pu.OracleEngine)
    metrics_o, records_o = _run(fx, "oracle")
    assert [r["Issue_Url"] for r in records] == [r["Issue_Url"] for r in records_o]
    g = np.array([r["prob"] for r in records])
    o = np.array([r["prob"] for r in records_o])
    assert np.abs(g - o).max() <= 1e-3
    clear = np.abs(o - 0.5) > 1e-3
    assert [r["predict"] for r, c in zip(records, clear) if c] == [r["predict"] for r, c in zip(records_o, clear) if c]


def test_custom_validation_callback_rebuilds_the_anchor_bank(monkeypatch):
    """callbacks.py:41-53: reset the bank, forward the anchors in chunks of 128 (here 8 anchors -> one chunk)."""
    from memvul_amd import callbacks, model_memory
    from memvul_amd.archive import load_archive

    root, arch, golden, test_path, w, dims = pu.make_fixture()
    monkeypatch.setattr(model_memory, "Engine", pu.OracleEngine)
    archive = load_archive(arch, overrides=pu.TEST_CONFIG, engine_options=dict(max_tokens=4096, max_batch=16, max_anchors=16))
    model = archive.model
    cb = callbacks.CustomValidation(anchor_path=golden, data_reader=archive.validation_dataset_reader)

    class _Trainer:
        pass

    tr = _Trainer()
    tr.model = model
    cb.on_epoch(tr, {}, 0, True)
    v1 = model._golden_instances_embeddings.copy()
    assert v1.shape == (8, 512) and len(model._golden_instances_labels) == 8
    cb.on_epoch(tr, {}, 1, True)  # a second epoch replaces the bank instead of appending to it
    assert np.array_equal(model._golden_instances_embeddings, v1) and len(model._golden_instances_labels) == 8
