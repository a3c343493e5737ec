"""the reference's OWN driver over the product plugin.

`/root/reference/predict_memory.py::test_siamese(package="memvul_amd", ...)` — the file imported verbatim, not a line of it
changed — with AllenNLP's surface provided by the tests-only stand-in of oracle/ref_harness/stubs, so that
`memvul_amd/registry.py` takes its HAVE_ALLENNLP branch: the product's `reader_memory` / `model_memory` /
`custom_pretrained_transformer` classes subclass AllenNLP's DatasetReader / Model (a torch.nn.Module) / TokenEmbedder and are
found by AllenNLP's `load_archive`, `DataLoader.from_params`, `Model.forward_on_instances` callers and `evaluate` under the
names the reference's configs use.  The outputs are compared with what the reference's OWN plugin (`package="MemVul"`)
produced on the same fixture (tests/golden/ref, tests/golden/ref12: ref_predictions.jsonl, ref_metrics.json,
ref_metric_all.json).  The engine behind the plugin is the numpy-oracle stand-in here (no GPU); tests/ref_driver_worker.py
runs in its own process because the stand-in must be importable as `allennlp` before memvul_amd.registry is imported.
Needs /root/reference (absent on the GPU box: skipped there)."""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REFERENCE = os.environ.get("MEMVUL_REFERENCE", "/root/reference")

pytestmark = pytest.mark.skipif(not os.path.isfile(os.path.join(REFERENCE, "predict_memory.py")),
                                reason="the reference checkout is not on this machine")


@pytest.mark.parametrize("which", ["ref", "ref12"])
def test_reference_test_siamese_runs_unmodified_over_the_product_plugin(which, tmp_path):
    out = tmp_path / "out.json"
    env = dict(os.environ, OMP_NUM_THREADS="4", MEMVUL_REFERENCE=REFERENCE)
    env.pop("MEMVUL_BERT_VOCAB", None)
    r = subprocess.run([sys.executable, os.path.join(HERE, "ref_driver_worker.py"), which, str(tmp_path), str(out)], env=env,
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-4000:]
    got = json.load(open(out))
    # the classes AllenNLP resolved are the product's, living in AllenNLP's class hierarchy
    assert got["model_class"] == "memvul_amd.model_memory.ModelMemory" and got["registered_model"] == "memvul_amd.model_memory"
    assert got["model_is_allennlp_model"] and got["model_is_torch_module"]

    REF = os.path.join(HERE, "golden", which)
    want_lines = [json.loads(x) for x in open(os.path.join(REF, "ref_predictions.jsonl"))]
    got_lines = [json.loads(x) for x in got["predictions_text"].splitlines()]
    assert [len(x) for x in got_lines] == [len(x) for x in want_lines]  # AllenNLP's evaluate: one line per batch of 16
    worst = 0.0
    for a, b in zip((r_ for line in got_lines for r_ in line), (r_ for line in want_lines for r_ in line)):
        assert a["Issue_Url"] == b["Issue_Url"] and a["label"] == b["label"]
        assert set(a["predict"]) == set(b["predict"])  # (key ORDER in the reference file is a str-hash order: model_memory.py:176 builds it from a set)
        worst = max(worst, max(abs(a["predict"][k] - b["predict"][k]) for k in b["predict"]))
    assert worst <= 1e-5, worst  # the numpy oracle against the reference's torch run (tests/test_reference_pin.py: 5e-6)
    want_metrics = json.load(open(os.path.join(REF, "ref_metrics.json")))
    assert set(got["metrics"]) == set(want_metrics)  # AllenNLP evaluate's final dict: ModelMemory.get_metrics(reset=True)
    for k, v in want_metrics.items():
        assert got["metrics"][k] == pytest.approx(v, abs=1e-6), k
    assert got["metrics_file"] == got["metrics"]
    for k, v in json.load(open(os.path.join(REF, "ref_metric_all.json"))).items():  # the reference's cal_metrics on the product's file
        assert got["metric_all"][k] == pytest.approx(v, abs=1e-6), k
