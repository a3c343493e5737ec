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


@pytest.mark.parametrize("backend", ["gloo", "tcp"])
def test_sharded_driver_world2_equals_single_process(tmp_path, monkeypatch, backend):
    """test_siamese_sharded on two ranks (contiguous shards, one all-gather of per-IR rows) == the single-process array
    driver: the same metrics on every rank, and the assembled predictions file holds the same records in the same order.
    backend "gloo": torch.distributed (the CPU harness); "tcp": the torch-free rendezvous hub as the transport — the path a
    GPU run takes when the ranks agree that RCCL is not usable (distributed.init_transport)."""
    import plumbing_util as pu
    from memvul_amd import model_memory, predict_memory

    fx = pu.make_fixture(n_irs=45)
    root, arch, golden, test_path, w, dims = fx
    out = str(tmp_path / "metrics.json")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    port = 29820 + (os.getpid() % 150) + (3 if backend == "tcp" else 0)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "dist_driver_worker.py"), root, arch, golden, test_path, out, backend]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=420)
    assert r.returncode == 0, r.stderr[-3000:]
    m0, m1 = json.load(open(out + ".rank0")), json.load(open(out + ".rank1"))
    # every rank lands on ITS device although test_config_memory.json says "cuda:0"
    assert (m0.pop("_device_index"), m1.pop("_device_index")) == (0, 1)
    assert m0 == m1
    monkeypatch.setattr(model_memory, "Engine", pu.OracleEngine)
    single_pred = os.path.join(root, "test_results", "single_result.json")
    ms = predict_memory.test_siamese(archive_file=arch, input_file=test_path, input_golden_file=golden, test_config=pu.TEST_CONFIG,
                                     predictions_output_file=single_pred, batch_size=16, cuda_device=0,
                                     engine_options=dict(max_tokens=16 * 256, max_batch=16, max_anchors=16), sweep="arrays")
    for k in ms:
        assert m0[k] == pytest.approx(ms[k], abs=1e-6), k
    assert json.load(open(os.path.join(root, "test_results", "sharded_metric.json"))) == m0
    # rank 0 assembled the single predictions file the reference's second pass (cal_metrics) reads; the parts are gone
    assert not any(os.path.exists(os.path.join(root, "test_results", f"sharded_result.json.part{rk}")) for rk in range(2))
    parts = [rec for line in open(os.path.join(root, "test_results", "sharded_result.json")) for rec in json.loads(line)]
    single = [rec for line in open(single_pred) for rec in json.loads(line)]
    assert [p["Issue_Url"] for p in parts] == [s["Issue_Url"] for s in single] and [p["label"] for p in parts] == [s["label"] for s in single]
    a = np.array([list(p["predict"].values()) for p in parts])
    b = np.array([list(s["predict"].values()) for s in single])
    assert np.abs(a - b).max() < 1e-6  # the oracle pads a shard's batches differently: fp32 rounding only


def test_tcp_fallback_transport_world3(tmp_path):
    """The rendezvous hub as the transport (distributed.init_transport(prefer="tcp"); what every rank falls back to TOGETHER when
    RCCL is not usable on all of them): three torch-free processes, ragged row blocks — every rank gets the rank-ordered
    concatenation, the maximum and the statistics."""
    world, port, out = 3, 30100 + (os.getpid() % 500), str(tmp_path / "tcp")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "tcp_worker.py"), str(r), str(world), str(port), out],
                              stderr=subprocess.PIPE, text=True) for r in range(world)]
    for p in procs:
        _, err = p.communicate(timeout=120)
        assert p.returncode == 0, err[-2000:]
    want_rows = np.concatenate([np.arange((r + 2) * 3, dtype=np.float32).reshape(r + 2, 3) + 100 * r for r in range(world)])
    for r in range(world):
        res = json.load(open(f"{out}.rank{r}"))
        assert np.array_equal(np.array(res["rows"], np.float32), want_rows)
        assert res["max"] == world - 0.5
        assert res["scores"] == [0.0, 0.5, 0.25, 0.5, 0.5, 0.5] and res["labels"] == [0, 1, 1, 1, 0, 1]


def test_hub_rejects_strangers_and_bad_ranks():
    """the hub used to accept any connection and to index by whatever rank it was told.  The handshake now carries
    the rank and a run token: a stray connection, a wrong token and an out-of-range rank are dropped while the hub keeps
    waiting for the real rank."""
    import socket
    import threading

    port = 30700 + (os.getpid() % 300)
    box = {}
    os.environ["MEMVUL_RUN_TOKEN"] = "hub-test"
    try:
        t = threading.Thread(target=lambda: box.setdefault("hub", mvdist._Hub(0, 2, "127.0.0.1", port, timeout_s=30.0)))
        t.start()
        tok = mvdist.run_token()

        def knock(payload):
            for _ in range(200):
                try:
                    c = socket.create_connection(("127.0.0.1", port), timeout=2.0)
                    break
                except OSError:
                    import time
                    time.sleep(0.02)
            c.sendall(payload)
            return c

        knock(b"GET / HTTP/1.0\r\n\r\n______")                     # a stranger
        knock((1).to_bytes(4, "little") + b"x" * 16)                   # right rank, wrong token
        knock((7).to_bytes(4, "little") + tok)                         # token ok, rank out of range
        good = mvdist._Hub(1, 2, "127.0.0.1", port, timeout_s=30.0)   # the real rank 1
        t.join(timeout=30)
        assert "hub" in box and len(box["hub"].peers) == 1
        out = {}
        t2 = threading.Thread(target=lambda: out.setdefault("r0", box["hub"].comm_allgather(np.array([5], np.int32))))
        t2.start()
        r1 = good.comm_allgather(np.array([9], np.int32))
        t2.join(timeout=10)
        assert r1.ravel().tolist() == [5, 9] and out["r0"].ravel().tolist() == [5, 9]
        good.comm_destroy(); box["hub"].comm_destroy()
    finally:
        os.environ.pop("MEMVUL_RUN_TOKEN", None)


def test_transport_agreement_cleans_up_when_a_peer_vanishes():
    """A rank that dies between the handshake and the agreement must not leave rank 0 with an open hub or a half-made
    communicator: init_transport raises, the module is back at "no transport", the engine's communicator is destroyed."""
    import threading

    port = 31050 + (os.getpid() % 300)
    os.environ["MEMVUL_RUN_TOKEN"] = "vanish-test"

    class FakeEngine:  # RCCL "usable" here, so rank 0 enters the agreement's first all-gather
        destroyed = 0

        def comm_prepare(self):
            pass

        def comm_destroy(self):
            FakeEngine.destroyed += 1

    def peer():
        h = mvdist._Hub(1, 2, "127.0.0.1", port, timeout_s=30.0)
        h.comm_destroy()  # gone before the agreement starts

    try:
        t = threading.Thread(target=peer)
        t.start()
        with pytest.raises((ConnectionError, OSError)):
            mvdist.init_transport(FakeEngine(), rank=0, world=2, prefer="rccl", addr="127.0.0.1", port=port)
        t.join(timeout=30)
        assert mvdist._comm is None and mvdist.transport_note() == "none" and FakeEngine.destroyed >= 1
    finally:
        os.environ.pop("MEMVUL_RUN_TOKEN", None)
        mvdist.shutdown()


def test_bench_two_ranks_control_flow_and_json_contract(tmp_path):
    """the REAL bench.py under `torch.distributed.run --nproc-per-node 2` with a numpy stand-in engine
    (MEMVUL_BENCH_STUB_ENGINE; no GPU, not a measurement): rendezvous, agreed transport, barrier + max-over-ranks timing, the
    corpus-shard leg with its ONE all-gather, and the N > 1 fields of the one JSON line rank 0 prints."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", MEMVUL_BENCH_STUB_ENGINE="1")
    port = 30400 + (os.getpid() % 200)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--batch", "32",
           "--seq-len", "64", "--anchors", "8", "--anchor-len", "64", "--layers", "1", "--shard-irs", "700"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]              # ONE line, from rank 0
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in d, k
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["warmup"] == 2 and d["scaling"] == "weak" and d["higher_is_better"] is True
    # the stand-in engine never yields a `value`: null, no roofline / kernels; its rate sits under a name of its own
    assert d["value"] is None and "roofline" not in d and "kernels" not in d and d["e2e_mfma_frac"] is None
    assert d["stub_rate_not_a_measurement"] == pytest.approx(2 * 6 * 32 / (d["ms_per_step"] * 6e-3), rel=1e-3)   # whole-job aggregate over both ranks
    assert d["data"] == "stub" and "NO GPU" in d["config"]["note"]
    assert d["config"]["comm_world"] == 2 and d["config"]["rccl_ranks"] == 0 and d["config"]["compute"] == "precise"
    assert "sweep_ms" in d["corpus_shard"] and "results_d2h_ms" in d["corpus_shard"]
    assert d["config"]["stats_transport"].startswith("tcp hub") and d["config"]["global_batch"] == 64
    assert d["stats_table_sum"] == 2 * 8 * 32 * 40       # both ranks' (score, label) rows reached rank 0: 40 thresholds x rows
    sh = d["corpus_shard"]
    assert sh["irs_per_rank"] == 700 and sh["irs_total"] == 1400 and sh["allgather_bytes_per_rank"] == 5600
    assert 0 < sh["scaling_vs_sum_of_ranks"] <= 1.0 + 1e-6 and sh["fixed_overhead_frac"] < 0.5
    assert "cpu_baseline" not in d and "precise" not in d  # rank-0, N = 1 only


def _stub_bench(extra, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(OMP_NUM_THREADS="1", MEMVUL_BENCH_STUB_ENGINE="1", **(env_extra or {}))
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "1", "--batch", "32", "--seq-len", "64",
           "--anchors", "8", "--anchor-len", "64", "--layers", "1", "--shard-irs", "300"] + extra
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)


def test_bench_launches_its_own_ranks_when_no_launcher_is_present():
    """`python bench.py --gpus 2` with NO WORLD_SIZE in the environment must run TWO ranks (it used to run one
    and print n_gpus 1): the process spawns them itself and rank 0's one line says n_gpus 2 over a 2-rank transport."""
    r = _stub_bench(["--gpus", "2"])
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["comm_world"] == 2 and d["config"]["global_batch"] == 64
    assert d["config"]["launcher"].startswith("bench.py self_launch")
    assert d["corpus_shard"]["irs_total"] == 600 and d["stats_table_sum"] == 2 * 5 * 32 * 40


def test_bench_world_8_self_launch_with_the_corpus_shard_leg():
    """the form the driver runs on an 8-GPU node — `bench.py --gpus 8`, here launcher-less on the stand-in engine over the hub
    transport — eight ranks, ONE line from rank 0, and the corpus-shard leg's ONE all-gather delivering the eight shards' (score, label) statistics
    in rank order, bit-equal to a single process that scores the same eight shards one after the other."""
    import hashlib

    import numpy as np

    from bench_stub_engine import Engine
    from memvul_amd import synth

    n = 150
    r = _stub_bench(["--gpus", "8", "--shard-irs", str(n)], timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["config"]["comm_world"] == 8 and d["config"]["global_batch"] == 8 * 32
    assert d["config"]["launcher"].startswith("bench.py self_launch") and d["config"]["stats_transport"].startswith("tcp hub")
    sh = d["corpus_shard"]
    assert sh["irs_per_rank"] == n and sh["irs_total"] == 8 * n and sh["allgather_bytes_per_rank"] == 8 * n
    assert d["stats_table_sum"] == 8 * 5 * 32 * 40
    # one process, the same eight shards in rank order (bench.corpus_shard_leg: ids by seed SEED + 5000 + rank, labels by SEED + 9000 + rank)
    dims = synth.BertDims(layers=1)
    scores, labels = [], []
    for rank in range(8):
        ids, lens = synth.make_ids(n, 64, dims.vocab_size, seed=synth.SEED + 5000 + rank)
        e = Engine(0)
        e.corpus_upload(ids, lens)
        e.corpus_run(0, n, 32)
        scores.append(e.corpus_results(0, n)[0][:, 0])
        labels.append(synth.make_labels(n, seed=synth.SEED + 9000 + rank))
    s_all, l_all = np.concatenate(scores).astype(np.float32), np.concatenate(labels).astype(np.uint8)
    assert sh["positives_gathered"] == int(l_all.sum())
    assert sh["stats_sha256"] == hashlib.sha256(s_all.tobytes() + l_all.tobytes()).hexdigest()


def test_bench_single_rank_line_on_the_stand_in_engine():
    """the N = 1 form on the stand-in engine (it used to crash in the matcher leg: no `topk`)."""
    r = _stub_bench([])
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 1 and d["value"] is None and d["data"] == "stub" and d["matcher"]["anchors"] == 1000


def test_bench_refuses_a_world_that_is_not_what_gpus_says():
    """--gpus 2 under a launcher that set WORLD_SIZE=1 (or any other size) is an error, not a one-rank run."""
    r = _stub_bench(["--gpus", "2"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_bench_self_launch_reports_a_failed_rank():
    """One rank failing ends the job with a non-zero code and no line (the stand-in fails rank 1 at construction)."""
    r = _stub_bench(["--gpus", "2"], {"MEMVUL_BENCH_STUB_FAIL_RANK": "1", "MEMVUL_HUB_TIMEOUT_S": "20"}, timeout=120)
    assert r.returncode != 0, r.stdout[-500:]
    assert "rank 1 of 2 exited with code" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_bench_refuses_more_gpus_than_are_visible():
    """The real engine's device count decides (no GPU in the CPU container -> 0 visible): a message, a non-zero code, nothing run."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MEMVUL_BENCH_STUB_ENGINE", "MEMVUL_BENCH_ONE_GPU_SMOKE")}
    import memvul_amd.binding as binding
    try:
        have = binding.device_count()
    except Exception:
        pytest.skip("libmemvul_hip.so not built")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(have + 2), "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "GPU(s) visible" in r.stderr and not r.stdout.strip()
