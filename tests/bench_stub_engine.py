"""Test-only stand-in for memvul_amd.binding.Engine used by bench.py under MEMVUL_BENCH_STUB_ENGINE=1 (CPU regression test of
the N > 1 control flow and JSON contract, tests/test_distributed_cpu.py).  No arithmetic of the path: scores are a hash of the ids."""
import time

import numpy as np


class Engine:
    def __init__(self, device=0, **kw):
        import os

        if os.environ.get("MEMVUL_BENCH_STUB_FAIL_RANK") == os.environ.get("RANK", "0"):
            raise RuntimeError("stand-in engine: this rank was told to fail (MEMVUL_BENCH_STUB_FAIL_RANK)")
        self.device = device
        self.G = 0
        self.ids = None
        self.prof = False
        self.n_launch = 0

    def load_state_dict(self, sd, compute_dtype=1):
        pass

    def set_streams(self, n):
        pass

    def anchor_append(self, ids, lens):
        self.G += ids.shape[0]

    def anchor_get(self):
        return np.zeros((self.G, 512), np.float32)

    def anchor_set(self, v):
        self.G = len(v)

    def corpus_upload(self, ids, lens):
        self.ids = np.asarray(ids)
        self.best = np.zeros((len(lens), 2), np.float32)
        self.idx = np.zeros(len(lens), np.int32)

    def corpus_run(self, first, count, batch, keep_probs=False, s_eff=0):
        p = (self.ids[first:first + count, 1:9].sum(1) % 1000).astype(np.float32) / 1000.0
        self.best[first:first + count, 0] = p
        self.best[first:first + count, 1] = 1 - p
        self.n_launch += 1
        time.sleep(0.0005)

    def corpus_results(self, first, count, with_probs=False):
        return self.best[first:first + count].copy(), self.idx[first:first + count].copy(), None

    def topk(self, u, k):
        self.n_topk = getattr(self, "n_topk", 0) + 1
        u = np.asarray(u)
        return np.zeros((len(u), k), np.float32), np.zeros((len(u), k), np.int32)

    def sync(self):
        pass

    def profile_enable(self, on=True):
        self.prof = on

    def profile_select(self, names=None):
        pass

    def profile_read(self):
        n, self.n_launch = self.n_launch, 0
        t, self.n_topk = getattr(self, "n_topk", 0), 0
        return {k: (0.3 * n, 11 * n) for k in ("gemm_qkv", "gemm_attn_out", "gemm_ffn1_gelu", "gemm_ffn2")} | {
            "attention": (0.1 * n, 11 * n), "match": (0.03 * t, t), "topk": (0.005 * t, t)}

    def close(self):
        pass
