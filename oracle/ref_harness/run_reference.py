"""CPU ORACLE (test infrastructure — NOT the product path): run the REFERENCE'S OWN CODE on a synthetic fixture.

`/root/reference` is pure Python but needs AllenNLP 2.4.0 / overrides, which are not installable here.  This harness
puts a tests-only stand-in for the AllenNLP surface (oracle/ref_harness/stubs/) in front of sys.path, imports the
reference's files VERBATIM from /root/reference and executes

    predict_memory.test_siamese   (predict_memory.py:49-114)  ->  load_archive, ReaderMemory.read (reader_memory.py),
                                   ModelMemory.forward_on_instances / forward / make_output_human_readable /
                                   get_metrics (model_memory.py), custom_PTM_embedder.forward, SiameseMeasureV1,
                                   find_best_thres, cal_f1 (custom_metric.py), AllenNLP evaluate (stand-in)
    predict_memory.cal_metrics / model_measure  (predict_memory.py:117-197)

on a seeded synthetic archive (random-init BERT geometry from memvul_amd/synth.py, a synthetic WordPiece vocabulary)
and synthetic issue-report / anchor / CVE files.  What third-party arithmetic runs underneath is the installed
`transformers` BertModel + torch (the reference pins transformers 4.1.0 / torch 1.8.1, README.md:25-27: version skew
documented in oracle/memvul_oracle.py).  tests/golden/make_ref_golden.py calls `generate()` and commits the outputs
under tests/golden/ref/; nothing here can run on the GPU box (/root/reference is absent there) and nothing under
memvul_amd/ imports it.

The only things touched in the reference's modules at run time are module GLOBALS the reference itself expects a
user to edit (`DATA_PATH = "xxx"`, predict_memory.py:200) and wrappers around names it imported (load_archive), used
to read the model's state back out; no reference source line is modified or copied.
"""
from __future__ import annotations

import json
import os
import sys
import tempfile
from typing import Dict

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
STUBS = os.path.join(HERE, "stubs")
REFERENCE = os.environ.get("MEMVUL_REFERENCE", "/root/reference")

CONFIG = {  # MemVul/config_memory.json with its Jsonnet locals substituted; trainer section dropped (unused at test time)
    "random_seed": 2021, "numpy_seed": 2021, "pytorch_seed": 2021,
    "dataset_reader": {
        "type": "reader_memory", "sample_neg": 0.01, "train_iter": 1, "same_diff_ratio": {"diff": 16, "same": 16},
        "anchor_path": "CWE_anchor_golden_project.json",
        "tokenizer": {"type": "pretrained_transformer", "model_name": "bert-base-uncased", "add_special_tokens": True, "max_length": 256},
        "token_indexers": {"tokens": {"type": "pretrained_transformer", "model_name": "bert-base-uncased", "namespace": "tags"}},
    },
    "train_data_path": "train_project.json", "validation_data_path": "validation_project.json",
    "model": {
        "type": "model_memory", "label_namespace": "labels", "dropout": 0.1, "device": "cuda:0", "use_header": True,
        "PTM": "bert-base-uncased", "temperature": 0.1,
        "text_field_embedder": {"token_embedders": {"tokens": {
            "type": "custom_pretrained_transformer", "model_name": "bert-base-uncased", "train_parameters": True,
            "pretrained_model_path": "further_pretrain/out_wwm/"}}},
    },
    "data_loader": {"batch_size": 32, "shuffle": False},
    "validation_data_loader": {"batch_size": 512, "shuffle": False},
}

WORDS = ("buffer overflow heap stack sql injection xss csrf auth bypass token leak race deadlock crash null pointer "
         "deref format string path traversal upload parser json yaml xml regex dos memory use after free double "
         "integer underflow privilege escalation sandbox escape cookie session header redirect ssrf the a an in of to "
         "when with attacker remote code execution allows via crafted request server client version before fixed "
         "issue bug error exception fails cannot please help update release build test docs typo feature").split()
SUFFIXES = ["s", "ed", "ing", "er", "able", "ly", "tion", "ness"]


def make_vocab(path: str) -> int:
    toks = ["[PAD]"] + [f"[unused{i}]" for i in range(1, 100)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]"]
    chars = list("abcdefghijklmnopqrstuvwxyz0123456789") + list(".,:;!?()[]{}-_/\\'\"#@=+*<>%$&|~`^")
    toks += chars + ["##" + c for c in chars if c.isalnum()]
    toks += sorted(set(WORDS)) + ["##" + s for s in SUFFIXES]
    toks = list(dict.fromkeys(toks))  # unique, order kept ("a" is a character and a word)
    with open(path, "w", encoding="utf-8") as f:
        f.write("\n".join(toks) + "\n")
    return len(toks)


def _text(rng, n):
    out = []
    for _ in range(n):
        w = str(rng.choice(WORDS))
        r = rng.random()
        if r < 0.15:
            w += str(rng.choice(SUFFIXES))            # word + known suffix piece
        elif r < 0.22:
            w = "".join(rng.choice(list("qzxvkj"), size=int(rng.integers(3, 8))))  # out-of-lexicon: character pieces
        elif r < 0.26:
            w = "CVE-2020-" + str(int(rng.integers(1000, 9999)))
        elif r < 0.29:
            w = "é中"                         # unknown characters -> [UNK]
        out.append(w)
    return " ".join(out)


def structured_matcher(w_random: np.ndarray) -> np.ndarray:
    """A matcher that discriminates (so thresholds, AUC and decisions in the fixture are non-trivial): P(same) falls
    with the L1 distance |u - v| and rises with sum(u); a random part keeps the two rows from being exact negatives."""
    P = w_random.shape[1] // 3
    wm = w_random.astype(np.float32).copy()
    wm[0, :P] += np.float32(0.0385)
    wm[1, :P] -= np.float32(0.0385)
    wm[0, 2 * P:] -= np.float32(0.45)
    wm[1, 2 * P:] += np.float32(0.45)
    return wm


def make_fixture(root: str, layers: int = 2, n_irs: int = 70, n_anchors: int = 9, seed: int = 11, weight_kwargs: Dict = None,
                 structured: bool = True, long_texts: bool = False) -> Dict:
    """Synthetic archive + data files under `root` (file names carry the substrings the reader dispatches on).
    Defaults = tests/golden/ref (2 layers, a discriminating matcher).  tests/golden/ref12: ``layers=12``, the trained-like
    weights of SURVEY.md §8(d) (``weight_kwargs``), synth's own x29 matcher (``structured=False``: |logit| ~ 3) and
    ``long_texts`` (most issue reports reach the 256-token truncation, anchors run up to 512 tokens)."""
    import torch
    from transformers import BertConfig, BertModel

    sys.path.insert(0, ROOT)
    from memvul_amd import synth

    rng = np.random.default_rng(seed)
    hf_dir = os.path.join(root, "out_wwm")
    arch = os.path.join(root, "archive")
    os.makedirs(hf_dir)
    os.makedirs(os.path.join(arch, "vocabulary"))
    os.makedirs(os.path.join(root, "test_results"))
    V = make_vocab(os.path.join(hf_dir, "vocab.txt"))
    dims = synth.BertDims(layers=layers, vocab_size=V)
    wk = dict(weight_kwargs) if weight_kwargs else dict(qk_scale=2.0, match_scale=2.0)
    w = synth.make_weights(dims, **wk)
    if structured:
        w[synth.KEY_MATCH_W] = structured_matcher(w[synth.KEY_MATCH_W])
    cfg = BertConfig(vocab_size=V, hidden_size=dims.hidden, num_hidden_layers=layers, num_attention_heads=dims.heads,
                     intermediate_size=dims.intermediate, max_position_embeddings=dims.max_pos, type_vocab_size=dims.type_vocab,
                     layer_norm_eps=dims.ln_eps, hidden_act="gelu")
    bert = BertModel(cfg, add_pooling_layer=True)
    sd = {k[len(synth.PFX_BERT):]: torch.from_numpy(v.copy()) for k, v in w.items() if k.startswith(synth.PFX_BERT)}
    sd["pooler.dense.weight"] = torch.from_numpy(w[synth.KEY_POOL_W].copy())  # `bert-base-uncased`'s own pooler: what
    sd["pooler.dense.bias"] = torch.from_numpy(w[synth.KEY_POOL_B].copy())    # BertPooler(PTM) deep-copies (model_memory.py:64)
    missing, unexpected = bert.load_state_dict(sd, strict=False)
    assert not unexpected and all(m.endswith("position_ids") for m in missing), (missing, unexpected)
    bert.save_pretrained(hf_dir)
    config = json.loads(json.dumps(CONFIG))
    config["model"]["text_field_embedder"]["token_embedders"]["tokens"]["pretrained_model_path"] = hf_dir
    json.dump(config, open(os.path.join(arch, "config.json"), "w"), indent=1)
    open(os.path.join(arch, "vocabulary", "labels.txt"), "w").write("same\ndiff\n")
    open(os.path.join(arch, "vocabulary", "non_padded_namespaces.txt"), "w").write("*tags\n*labels\n")
    cwes = [f"CWE-{100 + 7 * i}" for i in range(n_anchors)]
    golden = os.path.join(root, "CWE_anchor_golden_project.json")
    anchors = {c: _text(rng, int(rng.integers(60, 420) if long_texts else rng.integers(12, 90))) for c in cwes}
    json.dump(anchors, open(golden, "w"), indent=0)
    recs, cve = [], {}
    for i in range(n_irs):
        pos = i % 6 == 2
        title = _text(rng, int(rng.integers(3, 9)))
        nbody = int(rng.integers(90, 330)) if long_texts else int(rng.integers(4, 140 if i % 9 else 400))
        rec = {"Issue_Title": title, "Issue_Body": _text(rng, nbody),
               "Security_Issue_Full": "1" if pos else "0", "Issue_Url": f"https://example.invalid/repo/issues/{i}"}
        if pos:
            cid = f"CVE-2020-{1000 + i}"
            rec["CVE_ID"] = cid
            cwe = str(rng.choice(cwes))
            # one positive whose CVE has no CWE id: the reader drops it (reader_memory.py:103-105)
            cve[cid] = {"CVE_Description": "a crafted request allows `remote` code execution via http://x.invalid/a.b see CWE-79",
                        "CWE_ID": None if i == 14 else cwe}
            if i % 12 != 8:  # most positives quote their CWE's description (closer embeddings), some do not (misses)
                words = anchors[cwe].split()
                rec["Issue_Body"] = " ".join(words[: max(4, int(len(words) * rng.uniform(0.5, 1.0)))]) + " " + _text(rng, int(rng.integers(2, 12)))
        elif i % 11 == 5:  # a few negatives quote an anchor too (false alarms)
            words = anchors[str(rng.choice(cwes))].split()
            rec["Issue_Body"] = " ".join(words[: max(4, len(words) // 2)]) + " " + _text(rng, int(rng.integers(5, 30)))
        recs.append(rec)
    test_path = os.path.join(root, "test_project.json")
    json.dump(recs, open(test_path, "w"), indent=0)
    json.dump(cve, open(os.path.join(root, "xxxCVE_dict.json"), "w"), indent=0)  # data_path = "xxx" + 'CVE_dict.json' (reader_memory.py:62-64)
    return dict(root=root, archive=arch, hf_dir=hf_dir, golden=golden, test=test_path, dims=dims, weights=w, weight_kwargs=wk,
                vocab_size=V, seed=seed, layers=layers, structured=structured)


def _prepare_imports(hf_dir: str):
    for p in (REFERENCE, STUBS):
        if p in sys.path:
            sys.path.remove(p)
        sys.path.insert(0, p)
    os.environ["ALLENNLP_STUB_MODEL_DIR"] = hf_dir
    import numpy.lib.npyio as npyio

    if not hasattr(npyio, "load"):  # predict_memory.py:39 `from numpy.lib.npyio import load` (numpy 1.x location)
        npyio.load = np.load
    import transformers.utils.dummy_pt_objects as dummies

    if not hasattr(dummies, "ElectraForMaskedLM"):  # reader_memory.py:30, an unused import that transformers 5.x no longer carries
        dummies.ElectraForMaskedLM = type("ElectraForMaskedLM", (), {})


def write_weights_th(fx: Dict) -> None:
    """weights.th of the archive = the state dict of the REFERENCE model class built from the config, filled with the
    synthetic weights (keys as the reference's attribute names produce them)."""
    import torch
    from allennlp.common import Params
    from allennlp.data import Vocabulary
    from allennlp.models import Model

    config = Params.from_file(os.path.join(fx["archive"], "config.json"))
    vocab = Vocabulary.from_files(os.path.join(fx["archive"], "vocabulary"))
    mp = config.get("model")
    mp.params["device"] = "cpu"
    model = Model.from_params(vocab=vocab, params=mp)
    sd = model.state_dict()
    w = fx["weights"]
    inner = "_text_field_embedder.token_embedder_tokens.transformer_model."
    out = {}
    for k in sd:
        if k in w:
            out[k] = torch.from_numpy(w[k].copy())
        elif k.startswith(inner + "pooler.dense."):
            out[k] = torch.from_numpy(w["_bert_pooler.pooler.dense." + k.rsplit(".", 1)[1]].copy())
        elif k.endswith("position_ids"):
            out[k] = sd[k]
        else:
            raise KeyError(f"no synthetic weight for reference parameter {k}")
        assert tuple(out[k].shape) == tuple(sd[k].shape), k
    torch.save(out, os.path.join(fx["archive"], "weights.th"))
    fx["state_dict_keys"] = sorted(out.keys())


def run(fx: Dict, batch_size: int = 16) -> Dict:
    """Execute the reference on the fixture; returns everything a fixture file needs."""
    import torch

    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    _prepare_imports(fx["hf_dir"])
    cwd = os.getcwd()
    os.chdir(fx["root"])  # the reader opens "xxx" + "CVE_dict.json" and the anchor file relative to the CWD
    try:
        import predict_memory as pm  # the reference's file, verbatim
        from allennlp.common.util import import_module_and_submodules

        skipped = []
        import_module_and_submodules("MemVul", skipped)
        write_weights_th(fx)
        captured = {"probs": [], "meta": [], "logits": []}
        orig_load = pm.load_archive

        def load_and_hook(*a, **kw):
            archive = orig_load(*a, **kw)
            captured["model"] = archive.model
            captured["reader"] = archive.dataset_reader
            captured["golden_reader"] = archive.validation_dataset_reader

            def hook(_m, _inp, out):
                if isinstance(out, dict) and "probs" in out:
                    captured["probs"].extend(out["probs"])
                    captured["meta"].extend(out["meta"])

            archive.model.register_forward_hook(hook)
            # the match logits (model_memory.py:141, `self._projector(torch.cat([...]))`): output_dict carries only their softmax
            archive.model._projector.register_forward_hook(lambda _m, _i, out: captured["logits"].append(out.detach().numpy().copy()))
            return archive

        pm.load_archive = load_and_hook
        pm.DATA_PATH = fx["root"]
        test_config = {  # test_config_memory.json; device as predict_memory.py:210 sets it, here for the CPU
            "validation_dataset_reader": {
                "type": "reader_memory", "target": "Security_Issue_Full",
                "tokenizer": {"type": "pretrained_transformer", "model_name": "bert-base-uncased", "add_special_tokens": True, "max_length": 512},
                "token_indexers": {"tokens": {"type": "pretrained_transformer", "model_name": "bert-base-uncased", "namespace": "tags"}}},
            "model": {"device": "cpu"},
            "validation_data_loader": {"batch_size": 512, "shuffle": False}}
        out_metric = os.path.join(fx["root"], "test_results", "out_memvul_metric.json")
        out_result = os.path.join(fx["root"], "test_results", "out_memvul_result.json")
        metrics = pm.test_siamese(archive_file=fx["archive"], input_file=fx["test"], input_golden_file=fx["golden"],
                                  test_config=test_config, weights_file=None, output_file=out_metric,
                                  predictions_output_file=out_result, batch_size=batch_size, cuda_device=-1, seed=2021)
        model = captured["model"]
        thres = float(metrics["s_thres"])
        pm.cal_metrics("out_memvul_result", thres=thres)
        metric_all = json.load(open(os.path.join(fx["root"], "test_results", "out_memvul_metric_all.json")))
        # the reader's view of both files (token ids, order, labels, metadata)
        def dump_instances(reader, path):
            rows = []
            for ins in reader.read(path):
                f = ins.fields
                rows.append({"ids": [t.text_id for t in f["sample1"].tokens], "label": f["label"].label if "label" in f else None,
                             "meta": f["metadata"].metadata})
            return rows
        reader_dump = {"golden": dump_instances(captured["golden_reader"], fx["golden"]),
                       "test": dump_instances(captured["reader"], fx["test"])}
        res = dict(
            metrics=metrics, metric_all=metric_all, thres=thres,
            predictions_text=open(out_result).read(), metrics_file_text=open(out_metric).read(),
            anchors=model._golden_instances_embeddings.detach().numpy().astype(np.float32),
            anchor_labels=list(model._golden_instances_labels),
            probs=np.asarray(captured["probs"], np.float32), logits=np.concatenate(captured["logits"], 0).astype(np.float32),
            meta=captured["meta"], reader=reader_dump, same_idx=int(model._same_idx),
            skipped_submodules=skipped, state_dict_keys=fx["state_dict_keys"],
        )
        res["stats_cases"] = stats_cases(pm)
        return res
    finally:
        os.chdir(cwd)


def stats_cases(pm) -> list:
    """custom_metric.py:9-97 and predict_memory.py:117-156 on seeded score / label vectors: the reference's own
    functions, called directly."""
    import torch
    from MemVul.custom_metric import SiameseMeasureV1, cal_f1, find_best_thres

    cases = []
    for seed, n, pos_rate, shape in ((1, 400, 0.08, "beta"), (2, 257, 0.3, "uniform"), (3, 64, 0.5, "ties"), (4, 40, 0.0, "nopos"),
                                     (5, 90, 0.1, "allhigh")):
        rng = np.random.default_rng(seed)
        label = (rng.random(n) < pos_rate).astype(int)
        if shape == "beta":
            score = np.where(label == 1, rng.beta(4, 2, n), rng.beta(2, 5, n))
        elif shape == "uniform":
            score = rng.random(n)
        elif shape == "ties":
            score = np.round(rng.random(n) * 20) / 20.0  # many exact ties, also on the 0.01 threshold grid
        elif shape == "nopos":
            score = rng.random(n)
        else:
            score = 0.9 + 0.1 * rng.random(n)
        score = score.astype(np.float32).astype(float)  # what probs.tolist() of an fp32 tensor holds
        case = {"name": shape, "label": label.tolist(), "score": score.tolist()}
        pred = [1 if s >= 0.5 else 0 for s in score]
        case["cal_f1"] = cal_f1(label.tolist(), pred)
        if label.sum() > 0 and label.sum() < n:
            best = find_best_thres(label.tolist(), score.tolist())
            case["find_best_thres"] = {k: (float(v) if not isinstance(v, int) else v) for k, v in best.items()}
            m = SiameseMeasureV1(same_idx=0)
            probs = torch.tensor(np.stack([score, 1 - score], 1), dtype=torch.float32)
            meta = [{"instance": [{"label": "CWE-1" if l else "neg"}]} for l in label]
            for s in range(0, n, 37):
                m(probs[s:s + 37], meta[s:s + 37])
            got = m.get_metric(reset=True)
            case["siamese_measure"] = {k: (float(v) if not isinstance(v, int) else v) for k, v in got.items()}
            assert m._result == []
            mm, fpr, tpr = pm.model_measure(label.tolist(), pred, score.tolist(), list(range(n)))
            case["model_measure"] = {k: (float(v) if not isinstance(v, int) else v) for k, v in mm.items()}
        cases.append(case)
    return cases


def generate(out_dir: str, **fixture_kw) -> Dict:
    """Build the fixture in a temporary directory, run the reference, write inputs + outputs under out_dir."""
    import shutil

    root = tempfile.mkdtemp(prefix="mvref")  # no "test_" / "golden" in the directory name
    fx = make_fixture(root, **fixture_kw)
    res = run(fx)
    os.makedirs(out_dir, exist_ok=True)
    for name in ("CWE_anchor_golden_project.json", "test_project.json", "xxxCVE_dict.json"):
        shutil.copy(os.path.join(root, name), os.path.join(out_dir, name))
    shutil.copy(os.path.join(fx["hf_dir"], "vocab.txt"), os.path.join(out_dir, "vocab.txt"))
    cfg = json.load(open(os.path.join(fx["archive"], "config.json")))
    cfg["model"]["text_field_embedder"]["token_embedders"]["tokens"]["pretrained_model_path"] = "further_pretrain/out_wwm/"
    json.dump(cfg, open(os.path.join(out_dir, "config.json"), "w"), indent=1)
    open(os.path.join(out_dir, "ref_predictions.jsonl"), "w").write(res["predictions_text"])
    json.dump(res["metrics"], open(os.path.join(out_dir, "ref_metrics.json"), "w"), indent=1)
    json.dump(res["metric_all"], open(os.path.join(out_dir, "ref_metric_all.json"), "w"), indent=1)
    json.dump(res["reader"], open(os.path.join(out_dir, "ref_reader.json"), "w"))
    json.dump(res["stats_cases"], open(os.path.join(out_dir, "ref_stats_cases.json"), "w"))
    np.savez_compressed(os.path.join(out_dir, "ref_tensors.npz"), anchors=res["anchors"], probs=res["probs"], logits=res["logits"])
    import transformers
    import torch
    meta = dict(
        generator="oracle/ref_harness/run_reference.py (reference files executed verbatim from /root/reference)",
        reference_files=["predict_memory.py", "MemVul/model_memory.py", "MemVul/reader_memory.py", "MemVul/custom_metric.py",
                         "MemVul/custom_PTM_embedder.py", "MemVul/util.py"],
        skipped_submodules=res["skipped_submodules"], transformers=transformers.__version__, torch=torch.__version__,
        numpy=np.__version__, layers=fx["layers"], vocab_size=fx["vocab_size"], weight_seed=2021,
        weight_kwargs=fx["weight_kwargs"], structured_matcher=bool(fx["structured"]),
        matcher="oracle.ref_harness.run_reference.structured_matcher(synth matcher)" if fx["structured"] else "synth matcher (match_scale)", fixture_seed=fx["seed"], batch_size=16, thres=res["thres"], same_idx=res["same_idx"],
        anchor_labels=res["anchor_labels"], issue_urls=[m["instance"][0]["Issue_Url"] for m in res["meta"]],
        issue_labels=[m["instance"][0]["label"] for m in res["meta"]], state_dict_keys=res["state_dict_keys"],
    )
    json.dump(meta, open(os.path.join(out_dir, "meta.json"), "w"), indent=1)
    shutil.rmtree(root, ignore_errors=True)
    return res


if __name__ == "__main__":
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "ref")
    r = generate(out)
    print(json.dumps(r["metrics"], indent=1))
    print("anchors", r["anchors"].shape, "probs", r["probs"].shape, "skipped", r["skipped_submodules"])
