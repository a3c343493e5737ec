"""Host-side logic of the drop-in boundary (CPU only): metrics vs the scalar oracle, reader branches and
emission order, config parsing of the reference's own files, registry construction."""
import json
import os

import numpy as np
import pytest

from memvul_amd import custom_metric as cm
from memvul_amd import params
from memvul_amd.data import DataLoader, collate
from memvul_amd.predict_memory import measure_arrays, model_measure
from memvul_amd.reader_memory import ReaderMemory
from memvul_amd.registry import DatasetReader, Vocabulary
from memvul_amd.tokenizer import PretrainedTransformerIndexer, PretrainedTransformerTokenizer
from oracle import stats_oracle as so

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed,n,pos", [(0, 300, 0.3), (1, 1000, 0.01), (2, 64, 0.5)])
def test_find_best_thres_matches_reference_loops(seed, n, pos):
    rng = np.random.default_rng(seed)
    y = (rng.random(n) < pos).astype(int)
    y[0] = 1; y[1] = 0
    s = np.clip(rng.normal(0.55 + 0.15 * y, 0.15), 0, 1).astype(np.float32)
    s[5] = np.float32(0.6)  # a score exactly on a nominal threshold value
    ref = so.find_best_thres(list(y), [float(v) for v in s])
    got = cm.find_best_thres(y, s)
    assert got == ref  # counts, precision/recall/f1 and the chosen threshold (float arange value) identical
    full_ref = so.siamese_get_metric(list(y), [float(v) for v in s])
    full = cm.siamese_metrics(y, s)
    for k in full_ref:
        assert full[k] == pytest.approx(full_ref[k], rel=0, abs=1e-12), k


def test_find_best_thres_all_zero_f1_picks_last_threshold():
    y = np.array([0, 0, 1, 1]); s = np.array([0.9, 0.95, 0.1, 0.2], np.float32)
    ref = so.find_best_thres(list(y), [float(v) for v in s]); got = cm.find_best_thres(y, s)
    assert got == ref and got["f1"] >= 0


def test_threshold_table_is_additive_over_shards():
    rng = np.random.default_rng(3)
    y = (rng.random(500) < 0.2).astype(int); s = rng.random(500).astype(np.float32)
    whole = cm.threshold_confusion_table(y, s)
    parts = sum(cm.threshold_confusion_table(y[a:b], s[a:b]) for a, b in [(0, 100), (100, 333), (333, 500)])
    assert np.array_equal(whole, parts)
    assert cm.best_from_table(parts) == cm.find_best_thres(y, s)


def test_cal_f1_and_model_measure_match_reference():
    rng = np.random.default_rng(4)
    y = (rng.random(200) < 0.3).astype(int); pred = (rng.random(200) < 0.4).astype(int); sc = rng.random(200)
    assert cm.cal_f1(y, pred) == so.cal_f1(list(y), list(pred))
    got, _, _ = model_measure(y, pred, sc)
    ref = so.model_measure(list(y), list(pred), list(sc))
    assert got == ref
    m = measure_arrays(sc, y, thres=0.5)
    assert m["TP"] == int(np.sum((sc >= 0.5) & (y == 1)))


def test_siamese_measure_accumulates_like_reference():
    m = cm.SiameseMeasureV1(same_idx=0)
    probs = np.array([[0.7, 0.3], [0.2, 0.8], [0.55, 0.45]], np.float32)
    meta = [{"instance": [{"label": "CWE-79"}]}, {"instance": [{"label": "neg"}]}, {"instance": [{"label": "neg"}]}]
    m(probs, meta); m(probs[:1], meta[:1])
    lab, sc = m.arrays()
    assert lab.tolist() == [1, 0, 0, 1] and np.allclose(sc, [0.7, 0.2, 0.55, 0.7])
    assert m.get_metric(reset=False)["f1"] == 0  # only computed when the whole evaluation is done (l.84)
    out = m.get_metric(reset=True)
    assert out == {**so.siamese_get_metric([1, 0, 0, 1], [float(np.float32(v)) for v in [0.7, 0.2, 0.55, 0.7]])}
    assert m.arrays()[0].size == 0


def test_reference_configs_parse_unchanged():
    ref = "/root/reference"
    if not os.path.isdir(ref):
        pytest.skip("reference tree not mounted (GPU box)")
    c = params.load_config(os.path.join(ref, "MemVul", "config_memory.json"))
    t = params.load_config(os.path.join(ref, "test_config_memory.json"))
    merged = params.with_overrides(c, t)
    assert merged["model"]["type"] == "model_memory" and merged["model"]["text_field_embedder"]["token_embedders"]["tokens"]["type"] == "custom_pretrained_transformer"
    assert merged["dataset_reader"]["tokenizer"]["max_length"] == 256
    assert merged["validation_dataset_reader"]["tokenizer"]["max_length"] == 512
    assert merged["validation_data_loader"] == {"batch_size": 512, "shuffle": False}
    r = DatasetReader.from_params(merged["validation_dataset_reader"])
    assert isinstance(r, ReaderMemory) and r._tokenizer._max_length == 512


def test_jsonnet_subset():
    txt = 'local a = "x";\nlocal n = 3;\n{ "k": a, "v": [n, 2,], "s": "a stays a", // c\n }'
    assert params.parse_jsonnet_subset(txt) == {"k": "x", "v": [3, 2], "s": "a stays a"}


def _reader(max_length=32):
    tok = PretrainedTransformerTokenizer("bert-base-uncased", add_special_tokens=True, max_length=max_length)
    return ReaderMemory(tokenizer=tok, token_indexers={"tokens": PretrainedTransformerIndexer("bert-base-uncased", namespace="tags")})


def test_reader_branches_and_order():
    import pathlib
    import tempfile

    # NOT pytest's tmp_path: its name contains "test_", and the reader dispatches on path substrings
    # (reader_memory.py:138,146,155 — "path may accidentally contain the keywords")
    tmp_path = pathlib.Path(tempfile.mkdtemp(prefix="mvreader"))
    golden = tmp_path / "CWE_anchor_golden_project.json"
    golden.write_text(json.dumps({"CWE-79": "cross site scripting in web pages", "CWE-89": "sql injection " * 40}))
    r = _reader(max_length=16)
    g = list(r.read(str(golden)))
    assert [i["metadata"].metadata["instance"][0]["label"] for i in g] == ["CWE-79", "CWE-89"]
    assert all(i["metadata"].metadata["type"] == "golden" and "label" not in i.fields for i in g)
    assert len(g[1]["sample1"]) == 16 and g[1]["sample1"].tokens[0].text_id == 101 and g[1]["sample1"].tokens[-1].text_id == 102

    recs = [{"Issue_Title": f"t{i}", "Issue_Body": f"body {i}", "Security_Issue_Full": "1" if i in (1, 3) else "0",
             "Issue_Url": f"u{i}", "CVE_ID": f"CVE-{i}", "CWE_ID": "CWE-79" if i == 1 else "CWE-89"} for i in range(5)]
    for name, typ in (("test_project.json", "unlabel"), ("validation_project.json", "test")):
        p = tmp_path / name
        p.write_text(json.dumps(recs))
        ins = list(r.read(str(p)))
        urls = [i["metadata"].metadata["instance"][0]["Issue_Url"] for i in ins]
        assert urls == ["u3", "u1", "u4", "u2", "u0"]  # reversed concatenation: positives first (l.150-152)
        assert [i["label"].label for i in ins] == ["same", "same", "diff", "diff", "diff"]
        assert [i["metadata"].metadata["instance"][0]["label"] for i in ins] == ["CWE-89", "CWE-79", "neg", "neg", "neg"]
        assert all(i["metadata"].metadata["type"] == typ for i in ins)
    with pytest.raises(NotImplementedError):
        p = tmp_path / "train_project.json"; p.write_text(json.dumps(recs)); list(r.read(str(p)))

    vocab = Vocabulary({"labels": ["same", "diff"]})
    batch = collate(ins[:3], vocab)
    t = batch["sample1"]["tokens"]
    assert t["token_ids"].dtype == np.int64 and t["mask"].dtype == bool and t["token_ids"].shape == t["mask"].shape
    assert batch["label"].tolist() == [0, 0, 1] and len(batch["metadata"]) == 3
    dl = DataLoader(reader=r, data_path=str(tmp_path / "test_project.json"), batch_size=2); dl.index_with(vocab)
    assert len(dl) == 3 and [len(b["metadata"]) for b in dl] == [2, 2, 1]


def test_hash_tokenizer_must_be_asked_for(monkeypatch):
    """without a reachable vocab.txt the tokenizer used to fall back SILENTLY to CRC32 hashing (meaningless ids
    for trained weights).  Now that needs MEMVUL_ALLOW_HASH_TOKENIZER=1 (tests/conftest.py sets it for the synthetic
    fixtures) and warns; otherwise the constructor raises and says what to do."""
    monkeypatch.delenv("MEMVUL_ALLOW_HASH_TOKENIZER", raising=False)
    monkeypatch.delenv("MEMVUL_BERT_VOCAB", raising=False)
    with pytest.raises(RuntimeError, match="MEMVUL_BERT_VOCAB"):
        PretrainedTransformerTokenizer("bert-base-uncased", max_length=16)
    monkeypatch.setenv("MEMVUL_ALLOW_HASH_TOKENIZER", "1")
    assert len(PretrainedTransformerTokenizer("bert-base-uncased", max_length=16).tokenize("a b c")) == 5


def test_custom_validation_takes_the_reference_argument_order(tmp_path):
    """callbacks.py:27-31: CustomValidation(anchor_path, data_reader, data_loader, serialization_dir) — positional use and the
    (unused) data_loader keyword must work as in the reference."""
    import pathlib
    import tempfile

    from memvul_amd.callbacks import CustomValidation

    d = pathlib.Path(tempfile.mkdtemp(prefix="mvcb"))
    golden = d / "CWE_anchor_golden_project.json"
    golden.write_text(json.dumps({"CWE-79": "cross site scripting", "CWE-89": "sql injection"}))
    cb = CustomValidation(str(golden), _reader(), data_loader=None, serialization_dir=str(d))
    assert len(cb._anchors) == 2 and cb.serialization_dir == str(d)


def test_default_compute_dtype_is_the_contract_holding_one(monkeypatch):
    """ADVICE r3 (medium) / VERDICT r3 #1: without an explicit choice every Python entry (Engine.load_state_dict, ModelMemory, ModelSingle,
    load_archive) finalises the weights as MV_F16X8 — the mode that holds the reference's 1e-3 logit tolerance on trained-like weights;
    MV_F16 is an explicit opt-in ("f16" / "fast", by argument or $MEMVUL_COMPUTE); an unknown name raises before ctypes sees it."""
    from memvul_amd import binding

    monkeypatch.delenv("MEMVUL_COMPUTE", raising=False)
    assert binding.default_compute() == "precise" and binding.compute_dtype_of(None) == binding.MV_F16X8 == 6
    assert binding.compute_dtype_of("precise") == binding.compute_dtype_of("f16x8") == binding.MV_F16X8
    assert binding.compute_dtype_of("fast") == binding.compute_dtype_of("f16") == binding.compute_dtype_of(1) == binding.MV_F16 == 1
    monkeypatch.setenv("MEMVUL_COMPUTE", "fast")
    assert binding.compute_dtype_of(None) == binding.MV_F16
    monkeypatch.setenv("MEMVUL_COMPUTE", "bf16")
    with pytest.raises(ValueError):
        binding.compute_dtype_of(None)
    with pytest.raises(ValueError):
        binding.compute_dtype_of(5)  # MV_F16X2 of round 2 is gone
    import inspect

    from memvul_amd import model_memory, model_single
    for mod in (model_memory, model_single):  # neither names a default of its own: both defer to binding.default_compute()
        assert 'compute_dtype", None)' in inspect.getsource(mod)


def test_documented_switch_defaults_match_the_library_source():
    """bench.py and the docs name the library's default of MEMVUL_CLS_ASIDE / MEMVUL_CLS_ASIDE_MIN_LEN; the library source is the authority
    (engine.hip mv_handle): a default flipped in one place only would make bench.py report the wrong form under `precise_cls_aside_*`."""
    import re

    import bench

    src = open(os.path.join(ROOT, "memvul_amd", "csrc", "engine.hip")).read()
    m = re.search(r"bool cls_aside = (true|false);", src)
    assert m and bench.DEFAULT_CLS_ASIDE == ("1" if m.group(1) == "true" else "0")
    n = re.search(r"int cls_min_len = (\d+);", src)
    assert n and int(n.group(1)) == 128
    hdr = open(os.path.join(ROOT, "include", "memvul_hip.h")).read()
    assert "MEMVUL_CLS_ASIDE          1 (default) | 0" in hdr and "MEMVUL_CLS_ASIDE_MIN_LEN  1 .. 512 (default 128)" in hdr
    q = re.search(r"int qkv_aside_mask = (\d+);", src)  # 0 = "none" (round 6), 1 = "q" (rounds 4 - 6a)
    assert q and int(q.group(1)) == 0 and 'MEMVUL_QKV_ASIDE          a subset of "qkv", "" or "none" (default "none")' in hdr
    for doc in ("INTEGRATION.md", "DESIGN.md"):
        text = " ".join(open(os.path.join(ROOT, doc)).read().split())
        assert "default `none`" in text or "default is `none`" in text, doc
    # the development knobs are NOT product switches: the header names none of them as read by this library (engine.hip reads them under MEMVUL_DEV_SWITCHES only)
    from memvul_amd import binding
    block = src[src.index("#ifdef MEMVUL_DEV_SWITCHES"):src.index("#endif", src.index("#ifdef MEMVUL_DEV_SWITCHES"))]
    for k in binding.DEV_SWITCHES:
        assert src.count('"%s"' % k) == block.count('"%s"' % k) >= 1, k


def test_forward_by_length_groups_rows_by_their_own_padded_length():
    """binding.Engine.forward_by_length (ModelMemory.forward's engine call): a pad-to-longest batch of unsorted issue reports (predict_memory.py:97-101) is scored
    in one pass per padded length of the rows' OWN token counts; groups too small to fill a pass travel with the next longer one; every row's result lands in
    its place.  Host logic only: a recording stand-in for the engine."""
    from memvul_amd.binding import Engine

    class Rec:
        P, n_anchors = 4, 3
        BY_LENGTH_MIN_TOKENS = Engine.BY_LENGTH_MIN_TOKENS
        forward_by_length = Engine.forward_by_length

        def __init__(self):
            self.calls = []

        def forward(self, ids, lens, want_logits=True, want_probs=True, want_embed=False):
            assert ids.flags["C_CONTIGUOUS"] and ids.dtype == np.int32 and int(lens.max()) <= ids.shape[1]
            self.calls.append((ids.shape, lens.copy()))
            key = ids[:, 0].astype(np.float32)  # a row's "result" = a function of the row alone
            n = len(lens)
            return {"logits": None, "probs": np.repeat(key, 6).reshape(n, 3, 2) if want_probs else None, "best": np.stack([key, lens.astype(np.float32)], 1),
                    "best_idx": lens.astype(np.int32), "embed": None}

    rng = np.random.default_rng(3)
    B, S = 512, 512
    lens = rng.integers(5, S + 1, B).astype(np.int32)
    lens[7] = S
    ids = np.zeros((B, S), np.int32)
    ids[:, 0] = np.arange(B)
    e = Rec()
    out = e.forward_by_length(ids, lens, want_logits=False)
    assert out["logits"] is None and out["embed"] is None
    assert np.array_equal(out["best"][:, 0], np.arange(B)) and np.array_equal(out["best_idx"], lens) and np.array_equal(out["probs"][:, 2, 1], np.arange(B))
    widths = [c[0][1] for c in e.calls]
    assert widths == sorted(widths) and set(widths) <= {64, 128, 192, 256, 384, 512} and sum(c[0][0] for c in e.calls) == B
    for (shape, ls) in e.calls[:-1]:
        assert shape[0] * shape[1] >= Rec.BY_LENGTH_MIN_TOKENS  # every pass but (possibly) the last is worth a launch
    for (shape, ls) in e.calls:
        assert int(ls.max()) > (shape[1] - (64 if shape[1] <= 256 else 128))  # the pass runs at the padded length of its longest row
    padded_tokens = sum(c[0][0] * c[0][1] for c in e.calls)
    assert padded_tokens < 0.7 * B * S
    # a small batch, and a batch of one padded length, stay ONE call (the second at its own width)
    e = Rec(); e.forward_by_length(ids[:16], lens[:16]); assert len(e.calls) == 1 and e.calls[0][0] == (16, S)
    e = Rec(); l2 = np.full(B, 200, np.int32); e.forward_by_length(ids, l2); assert len(e.calls) == 1 and e.calls[0][0] == (B, 256)
    # min_tokens larger than the batch: everything travels together, at the longest row's length
    e = Rec(); e.forward_by_length(ids, lens, min_tokens=B * S); assert [c[0] for c in e.calls] == [(B, S)]


def test_native_record_formatter_prints_what_json_dumps_prints():
    """records.format_batch through mv_format_records (host-only code of libmemvul_hip.so: CPython's repr(float) restated in C++) gives the bytes of the Python
    formatter — i.e. of json.dumps(make_output_human_readable(...)) (model_memory.py:169-191 -> predict_memory.py:111; tests/test_plumbing.py pins that one) —
    on probabilities as the engine returns them (float32 -> double), on doubles of every magnitude (random bit patterns), and on the corner cases of the
    fixed / exponent switch; a non-finite value falls back to json's spelling."""
    from memvul_amd import build, records as R

    build.build(verbose=False)
    lib = R.native_formatter()
    assert lib is not None
    rng = np.random.default_rng(17)
    names = ['CWE-%d "q" \\ é' % i for i in range(37)]
    cols, fmt, nm = R.record_layout(names)
    urls = ["https://example.invalid/issues/%d?x=\"%%s\"\n" % i for i in range(300)]
    labels = ["neg" if i % 3 else "CWE-79" for i in range(300)]
    logit = rng.standard_normal((300, 37)).astype(np.float32) * 9
    p = (1.0 / (1.0 + np.exp(-logit))).astype(np.float32)
    py = R.format_batch(fmt, nm, urls, labels, p.astype(np.float64))
    assert R.format_batch(fmt, nm, urls, labels, p.astype(np.float64), lib) == py
    assert json.loads(py)[7]["predict"][names[5]] == float(p[7, 5]) and json.loads(py)[7]["Issue_Url"] == urls[7]

    def one(v):
        v = np.asarray(v, np.float64).reshape(-1, 1)
        return R.format_batch_native(lib, ["a"], ["u"] * len(v), ["l"] * len(v), v)

    def ref(v):
        return "[" + ", ".join('{"Issue_Url": "u", "label": "l", "predict": {"a": %r}}' % float(x) for x in np.atleast_1d(v)) + "]"

    bits = rng.integers(0, 2 ** 63, size=400000, dtype=np.int64).view(np.float64)
    bits = bits[np.isfinite(bits)]
    bits[::2] *= -1.0
    assert one(bits) == ref(bits)
    near = np.concatenate([10.0 ** np.arange(-8, 20), np.nextafter(10.0 ** np.arange(-8, 20), 0), np.nextafter(10.0 ** np.arange(-8, 20), np.inf),
                           [0.0, -0.0, 5e-324, 1.7976931348623157e308, 0.1, 1 / 3, 2.5, 9999999999999998.0, 123456789012345678.0, 1.5e-7, 1e22, 2.0 ** -20]])
    assert one(near) == ref(near)
    f32 = rng.random(600000).astype(np.float32).astype(np.float64) ** 3
    assert one(f32) == ref(f32)
    bad = p.astype(np.float64).copy(); bad[3, 4] = np.nan
    assert R.format_batch(fmt, nm, urls, labels, bad, lib) == R.format_batch(fmt, nm, urls, labels, bad) and "NaN" in R.format_batch(fmt, nm, urls, labels, bad, lib)


def test_evaluate_keeps_one_batch_in_flight_and_collects_in_order(tmp_path):
    """predict_memory.evaluate with a model that scores in two halves (ModelMemory.forward_begin / forward_end): batch k + 1 is begun BEFORE batch k is
    collected, every batch is collected exactly once and in order (the metric accumulators see the reference's order, predict_memory.py:103-110), the records
    come out in order, and an error in either half surfaces on the caller's thread with nothing left in flight.  Host logic only: a recording stand-in."""
    from memvul_amd.predict_memory import evaluate

    class Model:
        _same_idx = 0
        _golden_labels = ["CWE-1", "CWE-2"]

        def __init__(self, fail_at=None):
            self.log, self.in_flight, self.max_in_flight, self.fail_at = [], 0, 0, fail_at

        def eval(self):
            pass

        def forward_begin(self, sample1=None, label=None, metadata=None):
            k = metadata[0]["k"]
            if self.fail_at == ("begin", k):
                raise ValueError("begin %d" % k)
            self.log.append(("begin", k))
            self.in_flight += 1
            self.max_in_flight = max(self.max_in_flight, self.in_flight)
            return (k, metadata)

        def forward_end(self, pending):
            k, metadata = pending
            self.in_flight -= 1
            self.log.append(("end", k))
            if self.fail_at == ("end", k):
                raise ValueError("end %d" % k)
            p = np.full((len(metadata), 2, 2), 0.25 + 0.01 * k, np.float32)
            return {"meta": metadata, "probs": p}

        def get_metrics(self, reset=False):
            return {"n_end": sum(1 for e in self.log if e[0] == "end")}

    def loader(n):
        for k in range(n):
            yield {"sample1": None, "label": None,
                   "metadata": [{"type": "unlabel", "k": k, "instance": ({"Issue_Url": "u%d_%d" % (k, i), "label": "neg"},)} for i in range(3)]}

    m = Model()
    out = str(tmp_path / "pred.json")
    assert evaluate(m, loader(5), predictions_output_file=out) == {"n_end": 5}
    assert [e for e in m.log if e[0] == "end"] == [("end", k) for k in range(5)] and m.max_in_flight == 2 and m.in_flight == 0
    assert m.log.index(("begin", 1)) < m.log.index(("end", 0)) and m.log.index(("begin", 4)) < m.log.index(("end", 3))
    lines = [json.loads(l) for l in open(out)]
    assert [r[0]["Issue_Url"] for r in lines] == ["u%d_0" % k for k in range(5)] and lines[3][1]["predict"]["CWE-2"] == float(np.float32(0.28))
    for fail_at in (("begin", 2), ("end", 2), ("end", 4)):
        m = Model(fail_at)
        with pytest.raises(ValueError):
            evaluate(m, loader(5), predictions_output_file=out)
        assert m.in_flight == 0  # whatever was begun has been collected
