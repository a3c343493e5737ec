"""Subprocess body of tests/test_reference_driver_over_plugin.py: the REFERENCE'S OWN DRIVER,
`/root/reference/predict_memory.py::test_siamese`, unmodified, with `package="memvul_amd"` — i.e. AllenNLP's
`import_module_and_submodules` imports THIS repository's plugin instead of the reference's `MemVul` package, and
`load_archive` / `DataLoader` / `evaluate` (AllenNLP's, here the tests-only stand-in of oracle/ref_harness/stubs) find
`reader_memory`, `model_memory`, `custom_pretrained_transformer` in AllenNLP's registry under the names the reference's
configs use.  Runs in its own process because the stand-in must be importable as `allennlp` BEFORE `memvul_amd.registry` is
imported (it then takes its HAVE_ALLENNLP branch: the product classes subclass AllenNLP's Model / DatasetReader / ...).

The engine is the numpy-oracle stand-in of tests/plumbing_util.py (no GPU here); argv: <fixture name> <work dir> <out json>.
"""
import json
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REFERENCE = os.environ.get("MEMVUL_REFERENCE", "/root/reference")
STUBS = os.path.join(ROOT, "oracle", "ref_harness", "stubs")


def main(which: str, work: str, out_json: str, engine: str = "oracle"):
    for p in (HERE, ROOT, REFERENCE, STUBS):  # STUBS ends up first: `import allennlp` is the stand-in
        if p in sys.path:
            sys.path.remove(p)
        sys.path.insert(0, p)
    import torch

    torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))
    import allennlp  # noqa: F401  (the stand-in; must resolve before memvul_amd.registry is imported)

    REF = os.path.join(HERE, "golden", which)
    from oracle.ref_harness import run_reference as rr

    rr._prepare_imports(REF)  # numpy / transformers import shims of the reference's (unused) imports; tokenizer vocab = REF/vocab.txt
    from memvul_amd import registry, synth

    assert registry.HAVE_ALLENNLP, "memvul_amd.registry did not take its AllenNLP branch"
    meta = json.load(open(os.path.join(REF, "meta.json")))
    dims = synth.BertDims(layers=meta["layers"], vocab_size=meta["vocab_size"])
    w = synth.make_weights(dims, seed=meta["weight_seed"], **meta["weight_kwargs"])
    if meta.get("structured_matcher", True):
        w[synth.KEY_MATCH_W] = rr.structured_matcher(w[synth.KEY_MATCH_W])

    # the archive as AllenNLP writes it: config.json, vocabulary/, weights.th = torch.save(model.state_dict()) with the
    # REFERENCE model's key set (meta.json: the keys of the reference run's own state dict)
    root = os.path.join(work, "mvrefdrv")  # no "test_" / "golden" in the directory name
    arch = os.path.join(root, "archive")
    os.makedirs(os.path.join(arch, "vocabulary"))
    os.makedirs(os.path.join(root, "test_results"))
    for name in ("CWE_anchor_golden_project.json", "test_project.json", "xxxCVE_dict.json"):
        shutil.copy(os.path.join(REF, name), os.path.join(root, name))
    shutil.copy(os.path.join(REF, "xxxCVE_dict.json"), os.path.join(root, "CVE_dict.json"))
    shutil.copy(os.path.join(REF, "config.json"), os.path.join(arch, "config.json"))
    open(os.path.join(arch, "vocabulary", "labels.txt"), "w").write("same\ndiff\n")
    open(os.path.join(arch, "vocabulary", "non_padded_namespaces.txt"), "w").write("*tags\n*labels\n")
    inner = "_text_field_embedder.token_embedder_tokens.transformer_model."
    sd = {}
    for k in meta["state_dict_keys"]:
        if k in w:
            sd[k] = torch.from_numpy(w[k].copy())
        elif k.startswith(inner + "pooler.dense."):
            sd[k] = torch.from_numpy(w["_bert_pooler.pooler.dense." + k.rsplit(".", 1)[1]].copy())
        elif k.endswith("position_ids"):
            sd[k] = torch.arange(dims.max_pos).unsqueeze(0)
        else:
            raise KeyError(k)
    torch.save(sd, os.path.join(arch, "weights.th"))

    os.chdir(root)  # the readers open the anchor file / CVE_dict.json relative to the CWD (reader_memory.py:62-68)
    os.environ["MEMVUL_DATA_PATH"] = root
    import predict_memory as pm  # /root/reference/predict_memory.py, verbatim

    assert os.path.realpath(pm.__file__) == os.path.realpath(os.path.join(REFERENCE, "predict_memory.py")), pm.__file__
    if engine == "oracle":
        import plumbing_util as pu
        from memvul_amd import model_memory

        model_memory.Engine = pu.OracleEngine
    pm.DATA_PATH = root
    test_config = {  # test_config_memory.json; the device as predict_memory.py:210 would set it
        "validation_dataset_reader": {
            "type": "reader_memory", "target": "Security_Issue_Full",
            "tokenizer": {"type": "pretrained_transformer", "model_name": "bert-base-uncased", "add_special_tokens": True, "max_length": 512},
            "token_indexers": {"tokens": {"type": "pretrained_transformer", "model_name": "bert-base-uncased", "namespace": "tags"}}},
        "model": {"device": "cpu" if engine == "oracle" else "cuda:0"},
        "validation_data_loader": {"batch_size": 512, "shuffle": False}}
    out_metric = os.path.join(root, "test_results", "drv_metric.json")
    out_result = os.path.join(root, "test_results", "drv_result.json")
    metrics = pm.test_siamese(archive_file=arch, input_file=os.path.join(root, "test_project.json"),
                              input_golden_file=os.path.join(root, "CWE_anchor_golden_project.json"), test_config=test_config,
                              weights_file=None, output_file=out_metric, predictions_output_file=out_result, batch_size=16,
                              cuda_device=-1 if engine == "oracle" else 0, seed=2021, package="memvul_amd")
    assert "MemVul" not in sys.modules, "the reference's own plugin package was imported: the run would not prove the drop-in"
    pm.cal_metrics("drv_result", thres=float(meta["thres"]))
    metric_all = json.load(open(os.path.join(root, "test_results", "drv_metric_all.json")))
    from memvul_amd.model_memory import ModelMemory
    from allennlp.models import Model

    json.dump({"metrics": metrics, "metric_all": metric_all, "predictions_text": open(out_result).read(),
               "metrics_file": json.load(open(out_metric)),
               "model_class": f"{ModelMemory.__module__}.{ModelMemory.__name__}",
               "model_is_allennlp_model": issubclass(ModelMemory, Model), "model_is_torch_module": issubclass(ModelMemory, torch.nn.Module),
               "registered_model": f"{Model.by_name('model_memory').__module__}"}, open(out_json, "w"))


if __name__ == "__main__":
    main(*sys.argv[1:])
