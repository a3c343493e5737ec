"""predict_memory.py plumbing (BASELINE.json configs[0], scaled down): reader -> model -> metrics -> JSON-lines
-> cal_metrics.  On CPU the engine is replaced by the oracle-backed stand-in (tests only); on the GPU the real
HIP engine runs the same files and must agree with the oracle-backed run."""
import json
import os

import numpy as np
import pytest

import plumbing_util as pu
from memvul_amd import model_memory, predict_memory
from oracle import stats_oracle as so


def _run(fx, tag, sweep=False):
    root, arch, golden, test_path, w, dims = fx
    out_metric = os.path.join(root, "test_results", f"{tag}_metric.json")
    out_results = os.path.join(root, "test_results", f"{tag}_result.json")
    metrics = predict_memory.test_siamese(archive_file=arch, input_file=test_path, input_golden_file=golden, test_config=pu.TEST_CONFIG,
                                          output_file=out_metric, predictions_output_file=out_results, batch_size=16, cuda_device=0,
                                          engine_options=dict(max_tokens=16 * 256, max_batch=16, max_anchors=16), sweep=sweep)
    records = []
    for line in open(out_results):
        records.extend(json.loads(line))
    return metrics, records, out_results


def _check_format_and_metrics(fx, metrics, records, tag):
    root = fx[0]
    recs_in = json.load(open(fx[3]))
    assert len(records) == len(recs_in)
    # positives first (reversed concatenation), every record has one score per anchor
    anchors = list(json.load(open(fx[2])).keys())
    assert [r["label"] != "neg" for r in records][: sum(r["Security_Issue_Full"] == "1" for r in recs_in)] == [True] * sum(r["Security_Issue_Full"] == "1" for r in recs_in)
    for r in records:
        assert set(r) == {"Issue_Url", "label", "predict"} and list(r["predict"].keys()) == anchors
        assert all(0.0 <= v <= 1.0 for v in r["predict"].values())
    # second pass (cal_metrics) == the scalar oracle on the same records
    m = predict_memory.cal_metrics(f"{tag}_result", thres=0.5, data_path=root)
    ref = so.cal_metrics_records(records, thres=0.5)
    for k in ("TP", "FN", "TN", "FP", "f1", "auc", "ap"):
        assert m[k] == pytest.approx(ref[k], abs=1e-12), k
    assert os.path.exists(os.path.join(root, "test_results", f"{tag}_metric_all.json"))
    # evaluate()'s streaming metrics: siamese stats recomputed from the written records
    labels = [0 if r["label"] == "neg" else 1 for r in records]
    scores = [float(np.float32(max(r["predict"].values()))) for r in records]
    s = so.siamese_get_metric(labels, scores)
    assert metrics["s_f1-score"] == pytest.approx(s["f1"], abs=1e-9) and metrics["s_auc"] == pytest.approx(s["auc"], abs=1e-9)
    assert 0.0 <= metrics["accuracy"] <= 1.0 and "same_f1-score" in metrics and "diff_recall" in metrics


def test_plumbing_cpu_with_oracle_engine(monkeypatch):
    fx = pu.make_fixture()
    monkeypatch.setattr(model_memory, "Engine", pu.OracleEngine)
    metrics, records, _ = _run(fx, "oracle")
    _check_format_and_metrics(fx, metrics, records, "oracle")


def test_use_header_false_plumbing_cpu(monkeypatch):
    """model_memory.py:69-73 with `use_header: false` (no reference config sets it, but it is a constructor argument of the
    registered model): no _projector_single, the matcher is Linear(3 * 768, 2) on the pooler output.  The drop-in flow builds
    the engine with proj_dim = 768, and a state dict that contradicts the flag is rejected."""
    fx = pu.make_fixture(use_header=False)
    seen = {}

    class Spy(pu.OracleEngine):
        def __init__(self, device=0, **kw):
            seen.update(kw)
            super().__init__(device, **kw)

    monkeypatch.setattr(model_memory, "Engine", Spy)
    metrics, records, _ = _run(fx, "nohdr")
    assert seen["proj_dim"] == 768
    _check_format_and_metrics(fx, metrics, records, "nohdr")
    # the same archive read with use_header forced on: the weights lack the header -> a clear error, not garbage
    from memvul_amd.archive import load_archive
    with pytest.raises(ValueError, match="_projector_single"):
        load_archive(fx[1], cuda_device=0, overrides={"model": {"use_header": True}})


def test_sweep_driver_writes_the_same_files_cpu(monkeypatch):
    """test_siamese(sweep=True): one resident length-bucketed sweep instead of a forward per batch — same records in
    the same order with the same per-batch line grouping, same metrics (oracle-backed engine: exactly the same)."""
    fx = pu.make_fixture()
    monkeypatch.setattr(model_memory, "Engine", pu.OracleEngine)
    metrics, records, path = _run(fx, "loop")
    metrics_s, records_s, path_s = _run(fx, "sweep", sweep=True)
    assert [len(json.loads(l)) for l in open(path)] == [len(json.loads(l)) for l in open(path_s)]
    assert [r["Issue_Url"] for r in records] == [r["Issue_Url"] for r in records_s]
    a = np.array([list(r["predict"].values()) for r in records])
    b = np.array([list(r["predict"].values()) for r in records_s])
    assert np.abs(a - b).max() < 1e-6  # the oracle pads differently per chunk: fp32 rounding only
    for k in metrics:
        assert metrics[k] == pytest.approx(metrics_s[k], abs=1e-6), k
    _check_format_and_metrics(fx, metrics_s, records_s, "sweep")


def test_arrays_driver_writes_the_same_bytes_cpu(monkeypatch):
    """test_siamese(sweep="arrays"): reader -> arrays (batched tokenisation, no Instances) -> chunked resident sweeps with
    the records written by a thread.  With the oracle-backed engine and one chunk the predictions file is the sweep
    driver's BYTE FOR BYTE (the hand-built JSON lines equal json.dumps of the records) and the metrics are equal."""
    fx = pu.make_fixture(n_irs=75)
    monkeypatch.setattr(model_memory, "Engine", pu.OracleEngine)
    metrics_s, records_s, path_s = _run(fx, "sweep", sweep=True)
    metrics_a, records_a, path_a = _run(fx, "arrays", sweep="arrays")
    assert open(path_a, "rb").read() == open(path_s, "rb").read()
    for k in metrics_s:
        assert metrics_a[k] == pytest.approx(metrics_s[k], abs=1e-12), k
    _check_format_and_metrics(fx, metrics_a, records_a, "arrays")


def test_arrays_driver_chunked_and_without_a_predictions_file(monkeypatch):
    """Several chunks (chunk_batches=1, a ragged last one) through the writer thread, and the metrics-only form."""
    from memvul_amd.archive import load_archive

    fx = pu.make_fixture(n_irs=53)
    root, arch, golden, test_path, w, dims = fx
    monkeypatch.setattr(model_memory, "Engine", pu.OracleEngine)
    metrics_s, records_s, path_s = _run(fx, "sweep", sweep=True)
    archive = load_archive(arch, cuda_device=0, overrides=pu.TEST_CONFIG, engine_options=dict(max_tokens=16 * 256, max_batch=16, max_anchors=16))
    model = archive.model
    model.eval()
    model._golden_instances_embeddings = None
    model._golden_instances_labels = None
    model.forward_on_instances(list(archive.validation_dataset_reader.read(golden)))
    arrays = archive.dataset_reader.read_arrays(test_path)
    assert arrays["ids"].dtype == np.int32 and arrays["ids"].shape[0] == len(arrays["lens"]) == len(records_s)
    assert arrays["urls"] == [r["Issue_Url"] for r in records_s] and arrays["labels"] == [r["label"] for r in records_s]
    out = os.path.join(root, "test_results", "chunked_result.json")
    m1 = predict_memory.evaluate_arrays(model, arrays, 16, predictions_output_file=out, chunk_batches=1)
    recs = [r for line in open(out) for r in json.loads(line)]
    assert [len(json.loads(l)) for l in open(out)] == [len(json.loads(l)) for l in open(path_s)]
    a = np.array([list(r["predict"].values()) for r in recs])
    b = np.array([list(r["predict"].values()) for r in records_s])
    assert [r["Issue_Url"] for r in recs] == [r["Issue_Url"] for r in records_s] and np.abs(a - b).max() < 1e-6
    m2 = predict_memory.evaluate_arrays(model, arrays, 16)  # no predictions file: no writer thread, no probabilities
    for k in metrics_s:
        assert m1[k] == pytest.approx(metrics_s[k], abs=1e-6) and m2[k] == pytest.approx(metrics_s[k], abs=1e-6), k


def test_model_tar_gz_with_torch_weights_th(monkeypatch, tmp_path):
    """The form a trained model ships in (predict_memory.py:62): ``model.tar.gz`` holding config.json, vocabulary/ and
    ``weights.th`` = torch.save(state_dict) — loaded through the tar + torch path, same results as the extracted
    directory with weights.npz."""
    import tarfile

    import torch

    fx = pu.make_fixture(n_irs=20)
    root, arch, golden, test_path, w, dims = fx
    monkeypatch.setattr(model_memory, "Engine", pu.OracleEngine)
    metrics_d, records_d, _ = _run(fx, "dir", sweep="arrays")
    stage = tmp_path / "stage"
    (stage / "vocabulary").mkdir(parents=True)
    for name in ("config.json", "vocabulary/labels.txt", "vocabulary/non_padded_namespaces.txt"):
        (stage / name).write_bytes(open(os.path.join(arch, name), "rb").read())
    sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in w.items()}
    sd["_text_field_embedder.token_embedder_tokens.transformer_model.embeddings.position_ids"] = torch.arange(512).unsqueeze(0)  # an int64 buffer HF checkpoints carry
    torch.save(sd, stage / "weights.th")
    tar_path = tmp_path / "model.tar.gz"
    with tarfile.open(tar_path, "w:gz") as tf:
        for name in ("config.json", "weights.th", "vocabulary"):
            tf.add(stage / name, arcname=name)
    fx_tar = (root, str(tar_path), golden, test_path, w, dims)
    metrics_t, records_t, _ = _run(fx_tar, "tar", sweep="arrays")
    assert records_t == records_d
    for k in metrics_d:
        assert metrics_t[k] == pytest.approx(metrics_d[k], abs=1e-12), k


def test_record_writer_bytes_equal_json_dumps_including_worker_processes(tmp_path):
    """records.RecordWriter (in-thread and fanned out over spawned processes) == json.dumps of the reference's records:
    duplicate anchor labels (last wins, first position), labels / urls that need escaping or contain '%'."""
    from memvul_amd.records import RecordWriter

    golden = ["CWE-79", "CWE-89", 'we"ird %d %', "CWE-79", "ünï"]
    rng = np.random.default_rng(11)
    batches = []
    for n in (7, 3, 1):
        p = rng.random((n, len(golden)), dtype=np.float32)
        urls = [f"https://example.invalid/{i}?q=%20\"x\"" for i in range(n)]
        labels = [golden[i % len(golden)] if i % 2 else "neg" for i in range(n)]
        batches.append((urls, labels, p))
    order = {name: i for i, name in enumerate(golden)}
    want = ""
    for urls, labels, p in batches:
        recs = [{"Issue_Url": u, "label": lab, "predict": dict(zip(order.keys(), p[i, list(order.values())].astype(np.float64).tolist()))}
                for i, (u, lab) in enumerate(zip(urls, labels))]
        want += json.dumps(recs) + "\n"
    for workers in (0, 2):
        path = tmp_path / f"w{workers}.json"
        with RecordWriter(str(path), golden, workers=workers) as rw:
            for urls, labels, p in batches:
                rw.submit(urls, labels, p)
        assert path.read_text() == want, workers


def test_batch_ids_equals_tokenize_text_by_text():
    from memvul_amd.tokenizer import PretrainedTransformerTokenizer

    rng = np.random.default_rng(3)
    texts = [pu._text(rng, int(n)) for n in rng.integers(0, 400, size=40)] + ["", "Héllo, wörld! <script>alert(1)</script>"]
    for max_length, special in ((256, True), (32, True), (None, True), (16, False)):
        tok = PretrainedTransformerTokenizer(max_length=max_length, add_special_tokens=special)
        ids, lens = tok.batch_ids(texts)
        ids_mp, lens_mp = tok.batch_ids(texts, workers=3)
        assert np.array_equal(ids, ids_mp) and np.array_equal(lens, lens_mp)
        for i, t in enumerate(texts):
            ref = [k.text_id for k in tok.tokenize(t)]
            assert lens[i] == len(ref) and ids[i, :lens[i]].tolist() == ref and not ids[i, lens[i]:].any()


def test_wordpiece_vocab_path_tokenize_and_batch_ids(tmp_path):
    """With a vocab.txt reachable the tokenizer is the real lower-cased WordPiece (BertTokenizerFast) with the BERT
    special ids; truncation counts [CLS] / [SEP] (max_length 256 / 512 in the reference configs), and the batched
    array form gives the same ids as text-by-text tokenisation."""
    from memvul_amd.tokenizer import CLS_ID, SEP_ID, UNK_ID, PretrainedTransformerTokenizer

    words = sorted(set(pu.WORDS))
    vocab = ["[PAD]"] + [f"[unused{i}]" for i in range(99)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]"] + list(".,!<>()/") + words + ["##s", "##ing", "##ed"]
    (tmp_path / "vocab.txt").write_text("\n".join(vocab) + "\n")
    tok = PretrainedTransformerTokenizer(model_name=str(tmp_path), max_length=12)
    assert tok.vocab_size == len(vocab)
    got = tok.tokenize("Buffer OVERFLOWS crashed. Zzzunknown heap")
    assert [t.text for t in got] == ["[CLS]", "buffer", "overflow", "##s", "crash", "##ed", ".", "[UNK]", "heap", "[SEP]"]
    assert got[0].text_id == CLS_ID and got[-1].text_id == SEP_ID and got[7].text_id == UNK_ID
    rng = np.random.default_rng(5)
    texts = [pu._text(rng, int(n)) for n in rng.integers(1, 40, size=25)] + [""]
    ids, lens = tok.batch_ids(texts)
    assert ids.shape[1] == 12 and lens.max() == 12  # truncated to max_length INCLUDING the two specials
    for i, t in enumerate(texts):
        ref = [k.text_id for k in tok.tokenize(t)]
        assert ids[i, :lens[i]].tolist() == ref and ref[0] == CLS_ID and ref[-1] == SEP_ID and not ids[i, lens[i]:].any()


@pytest.mark.gpu
def test_sweep_driver_gpu_matches_batch_loop():
    fx = pu.make_fixture(n_irs=70)
    metrics, records, _ = _run(fx, "loop")
    metrics_s, records_s, _ = _run(fx, "sweep", sweep=True)
    assert [r["Issue_Url"] for r in records] == [r["Issue_Url"] for r in records_s]
    a = np.array([list(r["predict"].values()) for r in records])
    b = np.array([list(r["predict"].values()) for r in records_s])
    assert np.abs(a - b).max() <= 1e-3  # a row may run at a different padded length in the two drivers
    _check_format_and_metrics(fx, metrics_s, records_s, "sweep")


@pytest.mark.gpu
def test_arrays_driver_gpu_matches_sweep_driver():
    """The array-form driver on the real engine: same records in the same order as the Instance sweep (both length-bucketed;
    one chunk, so the same batches at the same padded lengths -> the same bytes)."""
    fx = pu.make_fixture(n_irs=70)
    metrics_s, records_s, path_s = _run(fx, "sweep", sweep=True)
    metrics_a, records_a, path_a = _run(fx, "arrays", sweep="arrays")
    assert [r["Issue_Url"] for r in records_a] == [r["Issue_Url"] for r in records_s]
    a = np.array([list(r["predict"].values()) for r in records_a])
    b = np.array([list(r["predict"].values()) for r in records_s])
    assert np.abs(a - b).max() <= 1e-6
    for k in metrics_s:
        assert metrics_a[k] == pytest.approx(metrics_s[k], abs=1e-6), k
    _check_format_and_metrics(fx, metrics_a, records_a, "arrays")


@pytest.mark.gpu
def test_sharded_driver_single_rank_gpu_equals_arrays_driver():
    """test_siamese_sharded with world size 1 on the real engine: no collective, same predictions file and metrics as
    test_siamese(sweep="arrays") (the 2-rank exchange itself is covered on CPU over gloo, tests/test_distributed_cpu.py)."""
    fx = pu.make_fixture(n_irs=60)
    root, arch, golden, test_path, w, dims = fx
    metrics_a, records_a, path_a = _run(fx, "arrays", sweep="arrays")
    out = os.path.join(root, "test_results", "sharded1_result.json")
    metrics_s = predict_memory.test_siamese_sharded(archive_file=arch, input_file=test_path, input_golden_file=golden, test_config=pu.TEST_CONFIG,
                                                    predictions_output_file=out, batch_size=16,
                                                    engine_options=dict(max_tokens=16 * 256, max_batch=16, max_anchors=16))
    assert open(out, "rb").read() == open(path_a, "rb").read()
    for k in metrics_a:
        assert metrics_s[k] == pytest.approx(metrics_a[k], abs=1e-9), k


@pytest.mark.gpu
def test_plumbing_gpu_matches_oracle_run(monkeypatch):
    fx = pu.make_fixture()
    metrics, records, _ = _run(fx, "hip")
    _check_format_and_metrics(fx, metrics, records, "hip")
    monkeypatch.setattr(model_memory, "Engine", pu.OracleEngine)
    metrics_o, records_o, _ = _run(fx, "oracle")
    gpu = np.array([list(r["predict"].values()) for r in records])
    ref = np.array([list(r["predict"].values()) for r in records_o])
    assert [r["Issue_Url"] for r in records] == [r["Issue_Url"] for r in records_o]
    assert np.abs(gpu - ref).max() <= 1e-3
    # CIR/NCIR decisions at 0.5 agree wherever the oracle score is not within tolerance of the threshold
    sg, sr = gpu.max(1), ref.max(1)
    clear = np.abs(sr - 0.5) > 1e-3
    assert np.array_equal((sg >= 0.5)[clear], (sr >= 0.5)[clear])
    assert metrics["accuracy"] == pytest.approx(metrics_o["accuracy"], abs=0.1)


@pytest.mark.gpu
def test_plumbing_cfg1_scale_gpu_matches_oracle_run(monkeypatch):
    """BASELINE.json configs[0] at its stated size: 1 000 synthetic issue reports of up to 128 tokens, 32 CWE anchors, batch 64,
    reader -> model -> metrics -> JSON-lines -> cal_metrics, the HIP engine against the oracle-backed run of the same files."""
    fx = pu.make_fixture(n_irs=1000, n_anchors=32, body_words=(20, 120))
    root, arch, golden, test_path, w, dims = fx

    def run(tag):
        out_metric = os.path.join(root, "test_results", f"{tag}_metric.json")
        out_results = os.path.join(root, "test_results", f"{tag}_result.json")
        m = predict_memory.test_siamese(archive_file=arch, input_file=test_path, input_golden_file=golden, test_config=pu.TEST_CONFIG,
                                        output_file=out_metric, predictions_output_file=out_results, batch_size=64, cuda_device=0,
                                        engine_options=dict(max_tokens=64 * 256, max_batch=64, max_anchors=32))
        return m, [r for line in open(out_results) for r in json.loads(line)]

    metrics, records = run("hip1k")
    monkeypatch.setattr(model_memory, "Engine", pu.OracleEngine)
    metrics_o, records_o = run("oracle1k")
    assert len(records) == 1000 and [r["Issue_Url"] for r in records] == [r["Issue_Url"] for r in records_o]
    assert all(len(r["predict"]) == 32 for r in records)
    gpu = np.array([list(r["predict"].values()) for r in records])
    ref = np.array([list(r["predict"].values()) for r in records_o])
    assert np.abs(gpu - ref).max() <= 1e-3
    sg, sr = gpu.max(1), ref.max(1)
    clear = np.abs(sr - 0.5) > 1e-3
    assert np.array_equal((sg >= 0.5)[clear], (sr >= 0.5)[clear])
    for k in ("s_auc", "s_ave_precision_score", "accuracy"):
        assert metrics[k] == pytest.approx(metrics_o[k], abs=0.02), k
