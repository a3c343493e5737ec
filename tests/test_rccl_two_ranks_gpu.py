"""SURVEY.md §8(e) on hardware — the moment two GPUs are visible (VERDICT r3 next #5a).  A `gpurun` box has ONE MI355X, so this
test SKIPS there; on the driver's multi-GPU node it is the first place `ncclCommInitRank` sees world > 1, BEFORE the scaling
bench does: two processes, one GPU each, the agreed transport of memvul_amd/distributed.py must be RCCL bound inside
libmemvul_hip.so, RCCL itself must report two ranks (mv_comm_info -> ncclCommCount), and the all-gather of the per-rank
(score, label) statistics must be the rank-ordered concatenation bit for bit on every rank."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.gpu
def test_two_rank_rccl_all_gather_of_stats(tmp_path):
    from memvul_amd.binding import device_count

    n_gpu = device_count()
    if n_gpu < 2:
        pytest.skip(f"{n_gpu} GPU visible: a two-rank RCCL communicator needs two (RCCL refuses two ranks on one device, "
                    "profiles/r03_k_two_ranks_one_gpu_rccl_attempt.json)")
    world = 2
    port = _free_port()
    out = str(tmp_path / "res")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MEMVUL_RUN_TOKEN=f"rccl-test-{port}")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "rccl_worker.py"), str(r), str(world), str(port), out],
                              cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    errs = []
    for p in procs:
        try:
            _, e = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            p.kill()
            _, e = p.communicate()
        errs.append(e)
    assert all(p.returncode == 0 for p in procs), [e[-2000:] for e in errs]
    res = [json.load(open(f"{out}.rank{r}")) for r in range(world)]
    want_s, want_l = [], []
    for r in range(world):
        rng = np.random.default_rng(100 + r)
        n = 3 + 2 * r
        want_s += rng.random(n, dtype=np.float32).tolist()
        want_l += (rng.random(n) < 0.3).astype(np.uint8).tolist()
    rows = sum((res[r]["own_rows"] for r in range(world)), [])
    for r in range(world):
        assert res[r]["note"].startswith("rccl"), res[r]["note"]                       # the AGREED transport is RCCL, not the hub
        assert res[r]["info"]["rccl_ranks"] == world and res[r]["info"]["rccl_rank"] == r and res[r]["info"]["rccl_version"] > 0, res[r]["info"]
        assert res[r]["scores"] == want_s and res[r]["labels"] == want_l               # bit-equal, rank order, true counts
        assert res[r]["rows"] == rows and res[r]["max"] == world - 0.5
