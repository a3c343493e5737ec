"""Host-side cost of the drop-in path per issue report, GPU excluded: JSON -> reader -> instances -> collate ->
(engine replaced by a stub that returns random scores at once) -> human-readable records -> JSON-lines -> cal_metrics.
Tells how many issue reports per second ONE Python process can feed / drain around the engine.
Usage: python scripts/host_pipeline_probe.py [N] [G]"""
import json
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import plumbing_util as pu  # noqa: E402
from memvul_amd import model_memory as mm  # noqa: E402
from memvul_amd import predict_memory as pm  # noqa: E402
from memvul_amd.archive import load_archive  # noqa: E402
from memvul_amd.data import DataLoader  # noqa: E402


class StubEngine:
    """binding.Engine surface used by ModelMemory; scores are random, returned immediately."""

    def __init__(self, device=0, **kw):
        self.same_idx = kw.get("same_idx", 0)
        self.g = 0
        self.rng = np.random.default_rng(0)

    def load_state_dict(self, sd, dtype=None): pass
    def anchor_reset(self): self.g = 0
    def anchor_append(self, ids, lens): self.g += len(lens)
    def anchor_count(self): return self.g
    def anchors_host(self): return np.zeros((self.g, 512), np.float32)

    def bucketed_sweep(self, ids, lens, batch_size, with_probs=False):
        n = len(lens)
        base = np.asarray(ids)[:, 1].astype(np.float64) * 1e-3 + np.asarray(lens)  # a row's scores depend on the row only
        p = (np.sin(base[:, None] * (np.arange(self.g) + 1.0)) * 0.5 + 0.5).astype(np.float32)
        idx = p.argmax(1).astype(np.int32)
        b = p[np.arange(n), idx]
        best = np.stack([b, 1 - b], 1) if self.same_idx == 0 else np.stack([1 - b, b], 1)
        return best.astype(np.float32), idx, (p if with_probs else None)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    g = int(sys.argv[2]) if len(sys.argv) > 2 else 124
    root, arch, golden, test_path, w, dims = pu.make_fixture(n_irs=n, n_anchors=g, layers=1)
    mm.Engine = StubEngine
    t = {}
    t0 = time.perf_counter()
    archive = load_archive(arch, overrides=pu.TEST_CONFIG, cuda_device=0)
    model, reader, reader_val = archive.model, archive.dataset_reader, archive.validation_dataset_reader
    model.eval()
    t["load_archive"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    gold = list(reader_val.read(golden))
    model._golden_instances_embeddings = None
    model._golden_instances_labels = None
    for s0 in range(0, len(gold), 128):
        model.forward_on_instances(gold[s0:s0 + 128])
    t["golden"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    loader = DataLoader.from_params(params={"batch_size": 512, "shuffle": False}, reader=reader, data_path=test_path)
    loader.index_with(model.vocab)
    instances = list(loader.iter_instances())
    t["read + tokenise (hash tokenizer) + instances"] = time.perf_counter() - t0
    out = os.path.join(root, "test_results", "probe_result.json")
    t0 = time.perf_counter()
    metrics = pm.evaluate_sweep(model, loader, output_file=os.path.join(root, "m.json"), predictions_output_file=out)
    t["sweep: collate + records + JSON-lines + metrics"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    pm.cal_metrics("probe_result", thres=0.5, data_path=root)
    t["cal_metrics (JSON-lines round trip)"] = time.perf_counter() - t0
    # ---- array form: batched tokenisation, no Instances, records written by a thread while the engine runs
    for workers in (0, 8):
        t0 = time.perf_counter()
        reader._dataset.clear()
        arrays = reader.read_arrays(test_path, workers=workers)
        t[f"arrays: read + tokenise (hash tokenizer, workers={workers})"] = time.perf_counter() - t0
    out_a = os.path.join(root, "test_results", "probe_arrays_result.json")
    for rw in (0, 6):
        t0 = time.perf_counter()
        pm.evaluate_arrays(model, arrays, 512, predictions_output_file=out_a, record_workers=rw)
        t[f"arrays: sweep + JSON-lines (record_workers={rw}) + metrics"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    pm.evaluate_arrays(model, arrays, 512)
    t["arrays: sweep + metrics, no predictions file"] = time.perf_counter() - t0
    same = open(out, "rb").read() == open(out_a, "rb").read() if g else None
    print(f"N = {len(instances)} issue reports, G = {g} anchors; arrays-path predictions file identical: {same}")
    for k, v in t.items():
        print(f"  {k:52s} {v:8.2f} s   {len(instances) / v:10.0f} IR/s")
    print(f"  predictions file: {os.path.getsize(out) / 1e6:.1f} MB")


if __name__ == "__main__":
    main()
