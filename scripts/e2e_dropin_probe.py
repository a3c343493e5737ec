"""End-to-end rate of the drop-in driver on one GPU, host work included: synthetic archive (random-init BERT-base) +
golden anchors + a test file of N issue reports of realistic length -> test_siamese(sweep="arrays") phases timed one
by one.  Usage (GPU box): python scripts/e2e_dropin_probe.py [N] [record_workers]"""
import json
import os
os.environ.setdefault("MEMVUL_ALLOW_HASH_TOKENIZER", "1")  # synthetic corpus: the hashing stand-in tokenizer is what this probe times
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import plumbing_util as pu  # noqa: E402
from memvul_amd import predict_memory as pm  # noqa: E402
from memvul_amd.archive import load_archive  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
    rw = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    t = {}
    root, arch, golden, test_path, w, dims = pu.make_fixture(n_irs=n, n_anchors=124, layers=12, body_words=(40, 330))
    t0 = time.perf_counter()
    archive = load_archive(arch, cuda_device=0, overrides=pu.TEST_CONFIG,
                           engine_options=dict(max_tokens=512 * 256, max_batch=512, max_anchors=128))
    model = archive.model
    model.eval()
    t["load_archive + engine + weights"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    model._golden_instances_embeddings = None
    model._golden_instances_labels = None
    model.forward_on_instances(list(archive.validation_dataset_reader.read(golden)))
    t["anchor bank (124 anchors)"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    arrays = archive.dataset_reader.read_arrays(test_path)
    t["read + tokenise (hashing stand-in tokenizer)"] = time.perf_counter() - t0
    lens = arrays["lens"]
    out = os.path.join(root, "test_results", "e2e_result.json")
    pm.evaluate_arrays(model, {k: (v[:2048] if hasattr(v, "__len__") and not isinstance(v, str) and len(v) == len(lens) else v)
                               for k, v in arrays.items()}, 512)  # warm-up on the first 2048 issue reports
    res = {}
    for name, kw in (("sweep + metrics (no predictions file)", {}),
                     ("sweep + JSON-lines in the writer thread + metrics", dict(predictions_output_file=out, record_workers=0)),
                     (f"sweep + JSON-lines with {rw} record workers + metrics", dict(predictions_output_file=out, record_workers=rw))):
        t0 = time.perf_counter()
        res[name] = pm.evaluate_arrays(model, arrays, 512, **kw)
        t[name] = time.perf_counter() - t0
    print(f"N = {n} issue reports, token lengths mean {lens.mean():.0f} / max {lens.max()} (cap 256), G = 124, batch 512, host cores {os.cpu_count()}")
    for k, v in t.items():
        print(f"  {k:58s} {v:8.2f} s   {n / v:10.0f} IR/s")
    print(f"  predictions file: {os.path.getsize(out) / 1e6:.1f} MB; s_f1 of the three runs:", [round(r['s_f1-score'], 6) for r in res.values()])


if __name__ == "__main__":
    main()
