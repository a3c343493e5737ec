"""The N>1 path on CPU: contiguous sharding + the single all-gather of (score,label) statistics over gloo."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from memvul_amd import custom_metric as cm
from memvul_amd import distributed as mvdist
from memvul_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 8, 9, 1221677):
        for world in (1, 2, 3, 8):
            spans = [mvdist.shard_range(n, r, world) for r in range(world)]
            assert sum(c for _, c in spans) == n
            pos = 0
            for f, c in spans:
                assert f == pos or c == 0
                pos += c
    assert mvdist.shard_range(1221677, 0, 8) == (0, 152710) and mvdist.shard_range(1221677, 7, 8) == (1068970, 152707)


@pytest.mark.parametrize("n", [1001, 3])
def test_all_gather_stats_world2_gloo(tmp_path, n):
    out = tmp_path / "res.json"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    port = 29600 + (os.getpid() % 200) + n % 7
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "dist_worker.py"), str(out), str(n)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.load(open(out))
    assert res["ok"] and res["same_best"] and res["world"] == 2 and res["tmax"] == 1.0
    # equals the single-process statistics
    rng = np.random.default_rng(123)
    labels = synth.make_labels(n, pos_rate=0.1)
    scores = np.clip(rng.normal(0.5 + 0.2 * labels, 0.15), 0, 1).astype(np.float32)
    if labels.min() != labels.max():
        m = cm.siamese_metrics(labels, scores)
        assert res["f1"] == m["f1"] and res["thres"] == float(m["thres"]) and res["auc"] == float(m["auc"])
