"""Round 6: the CPU number of the REFERENCE'S OWN FILES — `predict_memory.test_siamese` of /root/reference executed verbatim behind the tests-only
AllenNLP stand-in (oracle/ref_harness) — next to the `port` (oracle/hf_reference.py) that bench.py times on the GPU box, where /root/reference does not exist.
12-layer trained-like model, issue reports of ~256 tokens (most reach the truncation), 16 anchors of up to 512 tokens, batch 64, torch CPU threads = the
container's cores.  Times the reference's `evaluate` loop alone (model forward + its own host work: p.tolist(), the B x G dict loop with deepcopy, json.dumps), i.e. kind = "reference".
Runs in the BUILD container only.  Usage: python scripts/r06_reference_cpu_timing.py [n_irs]"""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.ref_harness import run_reference as rr  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 192
    root = tempfile.mkdtemp(prefix="mvreftime")
    fx = rr.make_fixture(root, layers=12, n_irs=n, n_anchors=16, weight_kwargs=dict(qk_scale=2.0, match_scale=29.0, trained_like=True), structured=False, long_texts=True)
    rr._prepare_imports(fx["hf_dir"])
    import predict_memory as pm  # the reference's file

    t = {}
    orig_eval = pm.evaluate

    def timed_eval(*a, **kw):
        t0 = time.perf_counter()
        try:
            return orig_eval(*a, **kw)
        finally:
            t["evaluate"] = time.perf_counter() - t0

    pm.evaluate = timed_eval
    t0 = time.perf_counter()
    res = rr.run(fx, batch_size=64)
    t["total"] = time.perf_counter() - t0
    n_scored = len(res["meta"])
    lens = [len(r["ids"]) for r in res["reader"]["test"]]
    import torch

    out = {"kind": "reference", "what": "the reference's own predict_memory.test_siamese / evaluate (files of /root/reference, verbatim) on torch CPU behind the tests-only AllenNLP stand-in",
           "issue_reports": n_scored, "mean_tokens": sum(lens) / len(lens), "anchors": 16, "batch": 64, "layers": 12, "threads": torch.get_num_threads(),
           "host_cores": os.cpu_count(), "evaluate_s": round(t["evaluate"], 2), "value": round(n_scored / t["evaluate"], 2), "unit": "issue-reports/s",
           "total_s_including_archive_load_anchor_bank_and_the_fixture_dumps": round(t["total"], 2)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
