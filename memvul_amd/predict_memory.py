"""The inference driver of the hot path (reference: predict_memory.py:49-197): ``test_siamese``,
``model_measure``, ``cal_metrics`` — same names, arguments and file formats — plus the AllenNLP
``evaluate`` loop it calls (predict_memory.py:103-110).

Differences that matter for throughput: scores stay ndarrays until serialisation; the
predictions file is still one JSON list per batch, ``{"Issue_Url","label","predict":{cwe: P(same)}}``
(model_memory.py:186-189), because ``cal_metrics`` (l.159-171) consumes exactly that.
"""
from __future__ import annotations

import copy
import json
import shutil
import logging
import os
from typing import Any, Dict, List, Optional

import numpy as np

from .archive import load_archive
from .data import DataLoader

logger = logging.getLogger(__name__)

DATA_PATH = os.environ.get("MEMVUL_DATA_PATH", "xxx")  # predict_memory.py:200


def evaluate(model, data_loader, cuda_device: int = -1, batch_weight_key: str = None, output_file: str = None,
             predictions_output_file: str = None, record_workers: Optional[int] = None) -> Dict[str, Any]:
    """``allennlp.training.util.evaluate``: loop the batches through ``model(**batch)``, stream one JSON line of human-readable predictions per
    batch, return (and optionally dump) the final metrics.  Same calls in the same order on the model, same bytes in the file — as a three-stage
    pipeline: a thread collates batch k + 1 (pad-to-longest: pure Python) while another runs ``model(**batch k)`` (the engine call releases the GIL) and
    the caller's thread writes the records of batch k - 1 through ``records.RecordWriter`` (the byte-equal array formatter of
    ``json.dumps(make_output_human_readable(...))``; ``record_workers`` / $MEMVUL_RECORD_WORKERS processes format, 0 = in this thread).  Serially the
    form ran at the SUM of its stages (collation + GPU + 0.65 us per printed double): profiles/r06_*_e2e_dropin.txt."""
    import queue
    import threading

    model.eval()
    if record_workers is None:
        record_workers = int(os.environ.get("MEMVUL_RECORD_WORKERS", "0"))
    q_in: "queue.Queue" = queue.Queue(maxsize=2)
    q_out: "queue.Queue" = queue.Queue(maxsize=2)
    err: List[BaseException] = []
    stop = threading.Event()

    def put(q, item):  # never blocks for ever on a consumer that has died
        while not stop.is_set():
            try:
                q.put(item, timeout=0.2)
                return
            except queue.Full:
                continue

    def collator():
        try:
            for batch in data_loader:
                if stop.is_set():
                    break
                put(q_in, batch)
        except BaseException as e:
            err.append(e)
            stop.set()
        finally:
            put(q_in, None)

    def scorer():
        # a model with forward_begin / forward_end (ModelMemory): batch k + 1 is handed to the engine BEFORE batch k is collected, so the GPU does not wait for
        # this thread's Python — or for the interpreter lock it shares with the two others — between batches; collected in order: the same calls in the same
        # order on the metric accumulators.  Any other model: model(**batch), as AllenNLP's evaluate calls it.
        two_halves = hasattr(model, "forward_begin") and hasattr(model, "forward_end")
        pending = None
        try:
            while not stop.is_set():
                try:
                    batch = q_in.get(timeout=0.2)
                except queue.Empty:
                    continue
                if batch is None:
                    break
                if two_halves:
                    nxt = model.forward_begin(**batch)
                    out = model.forward_end(pending) if pending is not None else None
                    pending = nxt
                    if out is None:
                        continue
                else:
                    out = model(**batch)
                if predictions_output_file:
                    put(q_out, out)
            if pending is not None and not stop.is_set():
                last, pending = pending, None
                out = model.forward_end(last)
                if predictions_output_file:
                    put(q_out, out)
        except BaseException as e:
            err.append(e)
            stop.set()
        finally:
            if pending is not None:  # an error on the way: leave no batch in flight inside the engine
                try:
                    model.forward_end(pending)
                except BaseException:
                    pass
            put(q_out, None)

    th = [threading.Thread(target=collator, name="memvul-collate", daemon=True), threading.Thread(target=scorer, name="memvul-score", daemon=True)]
    for t in th:
        t.start()
    writer, plain = None, None
    try:
        while True:
            try:
                out = q_out.get(timeout=0.2)
            except queue.Empty:
                if stop.is_set() and not any(t.is_alive() for t in th):
                    break
                continue
            if out is None:
                break
            if not hasattr(model, "_golden_labels"):  # another registered model (model_single): its own make_output_human_readable, as AllenNLP's evaluate calls it
                if plain is None:
                    plain = open(predictions_output_file, "w")
                plain.write(json.dumps(model.make_output_human_readable(out)) + "\n")
                continue
            if "meta" not in out or out["meta"][0]["type"] not in ["test", "unlabel"]:
                continue  # (a golden batch has no records: make_output_human_readable returns the empty dict)
            if writer is None:
                from .records import RecordWriter

                writer = RecordWriter(predictions_output_file, model._golden_labels, workers=record_workers)
            meta = out["meta"]
            p_same = out["p_same"] if "p_same" in out else np.asarray(out["probs"])[:, :, model._same_idx]
            writer.submit([m["instance"][0]["Issue_Url"] for m in meta], [m["instance"][0]["label"] for m in meta], p_same)
    except BaseException:
        stop.set()
        raise
    finally:
        stop.set() if err else None
        for t in th:
            t.join()
        if writer is not None:
            writer.close()
        elif plain is not None:
            plain.close()
        elif predictions_output_file and not err:
            open(predictions_output_file, "w").close()
    if err:
        raise err[0]
    final_metrics = model.get_metrics(reset=True)
    if output_file:
        with open(output_file, "w") as f:
            json.dump(_jsonable(final_metrics), f, indent=4)
    return final_metrics


def evaluate_sweep(model, data_loader, output_file: str = None, predictions_output_file: str = None) -> Dict[str, Any]:
    """``evaluate`` with the batches of the loader scored in ONE resident length-bucketed sweep (ModelMemory.sweep):
    identical files and metrics, no per-batch host round trip."""
    model.eval()
    instances = list(data_loader.iter_instances())
    if predictions_output_file and instances and hasattr(model, "sweep_scores"):
        # the records from the arrays (records.RecordWriter: the bytes of json.dumps(make_output_human_readable(...)), tests/test_plumbing.py) instead of
        # one dict per issue report: 1.4 + 0.4 s of a 40 k-report file's 6.7 s (profiles/r06_*_e2e_dropin.txt)
        from .records import RecordWriter

        meta, p_same = model.sweep_scores(instances, data_loader.batch_size)
        bs = data_loader.batch_size
        with RecordWriter(predictions_output_file, model._golden_labels) as rw:
            for s0 in range(0, len(meta), bs):
                m = meta[s0:s0 + bs]
                rw.submit([x["instance"][0]["Issue_Url"] for x in m], [x["instance"][0]["label"] for x in m], p_same[s0:s0 + bs])
    else:
        per_batch = model.sweep(instances, data_loader.batch_size)
        if predictions_output_file:
            with open(predictions_output_file, "w") as pf:
                for recs in per_batch:
                    pf.write(json.dumps(recs) + "\n")
    final_metrics = model.get_metrics(reset=True)
    if output_file:
        with open(output_file, "w") as f:
            json.dump(_jsonable(final_metrics), f, indent=4)
    return final_metrics


def evaluate_arrays(model, arrays: Dict[str, Any], batch_size: int = 512, output_file: str = None,
                    predictions_output_file: str = None, chunk_batches: int = 32, record_workers: Optional[int] = None) -> Dict[str, Any]:
    """``evaluate_sweep`` on the array form of the evaluation set (ReaderMemory.read_arrays), pipelined: the set is scored
    in chunks of ``chunk_batches`` batches, and while the engine sweeps chunk k (the GIL is released inside the library)
    a writer thread formats and writes the JSON-lines of chunk k-1 — one line per ``batch_size`` issue reports, the
    bytes ``evaluate`` writes (``record_workers`` / $MEMVUL_RECORD_WORKERS processes format them; 0 = in the thread).
    Without a predictions file no per-IR Python object is created at all."""
    import queue
    import threading

    model.eval()
    if record_workers is None:
        record_workers = int(os.environ.get("MEMVUL_RECORD_WORKERS", "0"))
    # ``arrays``: the whole set as one dict (ReaderMemory.read_arrays) or a stream of such dicts (ReaderMemory.iter_arrays: each a multiple of batch_size
    # samples but the last; the next one is being tokenised while this one is scored)
    parts = [arrays] if isinstance(arrays, dict) else arrays
    step = max(1, chunk_batches) * batch_size
    q: "queue.Queue" = queue.Queue(maxsize=2)
    err: List[BaseException] = []

    def writer():
        try:
            from .records import RecordWriter

            with RecordWriter(predictions_output_file, model._golden_labels, workers=record_workers) as rw:
                while True:
                    item = q.get()
                    if item is None:
                        return
                    part, s0, p_same = item
                    for b0 in range(0, len(p_same), batch_size):
                        a, b = s0 + b0, s0 + min(b0 + batch_size, len(p_same))
                        rw.submit(part["urls"][a:b], part["labels"][a:b], p_same[b0:b0 + batch_size])
        except BaseException as e:  # surfaced on the caller's thread below
            err.append(e)
            while q.get() is not None:
                pass

    th = None
    if predictions_output_file:
        th = threading.Thread(target=writer, name="memvul-records", daemon=True)
        th.start()
    try:
        for part in parts:
            n = len(part["lens"])
            for s0 in range(0, n, step):
                _, _, p_same = model.sweep_arrays(part, s0, min(n, s0 + step), batch_size, with_probs=th is not None)
                if th is not None:
                    q.put((part, s0, p_same))
    finally:
        if th is not None:
            q.put(None)
            th.join()
    if err:
        raise err[0]
    final_metrics = model.get_metrics(reset=True)
    if output_file:
        with open(output_file, "w") as f:
            json.dump(_jsonable(final_metrics), f, indent=4)
    return final_metrics


def _jsonable(x):
    if isinstance(x, dict):
        return {k: _jsonable(v) for k, v in x.items()}
    if isinstance(x, (np.floating, np.integer)):
        return x.item()
    if isinstance(x, np.ndarray):
        return x.tolist()
    return x


def test_siamese(archive_file, input_file, input_golden_file, test_config=None, weights_file=None, output_file=None,
                 predictions_output_file=None, batch_size=64, cuda_device=0, seed=2021, package="memvul_amd",
                 batch_weight_key="", file_friendly_logging=False, engine_options=None, sweep=False) -> Dict[str, Any]:
    """predict_memory.py:49-114.  ``sweep=True``: score the evaluation set in one resident length-bucketed sweep
    (same outputs; see evaluate_sweep).  ``sweep="arrays"``: the same through the array form of the reader
    (ReaderMemory.read_arrays -> evaluate_arrays: batched tokenisation, no Instances, records written by a thread
    while the engine runs)."""
    overrides = test_config or ""
    archive = load_archive(archive_file, weights_file=weights_file, cuda_device=cuda_device, overrides=overrides,
                           engine_options=engine_options)
    config = archive.config
    model = archive.model
    dataset_reader = archive.dataset_reader                        # loads the test samples
    dataset_reader_validation = archive.validation_dataset_reader  # loads the golden anchors
    model.eval()

    logger.info("Reading golden data from %s", input_golden_file)
    golden_samples = list(dataset_reader_validation.read(input_golden_file))
    model._golden_instances_embeddings = None
    model._golden_instances_labels = None
    model.forward_on_instances(golden_samples[:128])
    if len(golden_samples) > 128:
        model.forward_on_instances(golden_samples[128:])

    logger.info("Reading evaluation data from %s", input_file)
    if sweep == "arrays":
        bs = batch_size or int((config.get("validation_data_loader") or config.get("data_loader") or {}).get("batch_size", 512))
        tw = int(os.environ.get("MEMVUL_TOKENIZER_WORKERS", "0"))
        if tw == 0 and hasattr(dataset_reader, "iter_arrays") and hasattr(getattr(dataset_reader, "_tokenizer", None), "batch_ids"):  # streamed: chunk k + 1 is tokenised while chunk k is scored
            arrays = dataset_reader.iter_arrays(input_file, chunk=32 * bs)
        else:
            arrays = dataset_reader.read_arrays(input_file, workers=tw)
        metrics = evaluate_arrays(model, arrays, bs, output_file=output_file, predictions_output_file=predictions_output_file)
        logger.info("Finished evaluating.")
        return metrics
    data_loader_params = dict(config.get("validation_data_loader") or config.get("data_loader") or {})
    if batch_size:
        data_loader_params["batch_size"] = batch_size
    data_loader = DataLoader.from_params(params=data_loader_params, reader=dataset_reader, data_path=input_file)
    data_loader.index_with(model.vocab)
    if sweep:
        metrics = evaluate_sweep(model, data_loader, output_file=output_file, predictions_output_file=predictions_output_file)
    else:
        metrics = evaluate(model, data_loader, cuda_device, batch_weight_key, output_file=output_file,
                           predictions_output_file=predictions_output_file)
    logger.info("Finished evaluating.")
    return metrics


test_siamese.__test__ = False  # not a pytest test


def test_siamese_sharded(archive_file, input_file, input_golden_file, test_config=None, weights_file=None, output_file=None,
                         predictions_output_file=None, batch_size=512, seed=2021, engine_options=None, backend=None) -> Dict[str, Any]:
    """The multi-GPU form of ``test_siamese`` (one process per GPU under torchrun): every rank loads the
    archive on its own GPU and builds the anchor bank redundantly, scores its contiguous shard of the evaluation set
    (array form, ``read_arrays(shard=...)`` -> ``sweep_arrays``) and writes its own JSON-lines part
    (``<predictions_output_file>.part<rank>``); ONE all-gather of the per-IR ``(p_0, p_1, is_positive)`` rows then lets
    every rank compute the metrics on the whole set — the numbers a single process gives on the same per-IR results.
    Rank 0 writes ``output_file``."""
    from . import distributed as mvdist

    rank, local_rank, world = mvdist.env_world()
    if backend not in (None, "rccl", "tcp", "gloo"):
        raise ValueError(f"backend {backend!r}: expected None / 'rccl' (RCCL bound in the library, agreed over the rendezvous hub), 'tcp' "
                         "(the hub itself) or 'gloo' (torch.distributed, the CPU test harness)")
    use_torch = world > 1 and backend == "gloo"
    if use_torch:
        mvdist.init_process_group("gloo")
    # the reference takes the model's device from the config (predict_memory.py:210, test_config_memory.json "cuda:0");
    # here every rank must land on ITS GPU whatever the config says
    overrides = json.loads(test_config) if isinstance(test_config, str) and test_config else copy.deepcopy(test_config or {})
    overrides.setdefault("model", {})["device"] = f"cuda:{local_rank}"
    archive = load_archive(archive_file, weights_file=weights_file, cuda_device=local_rank, overrides=overrides,
                           engine_options=engine_options)
    model = archive.model
    model.eval()
    try:
        if world > 1 and not use_torch:
            # the default on GPUs: RCCL bound inside libmemvul_hip.so, collective on the engine's stream, no torch.distributed; the
            # ranks agree over the rendezvous hub whether RCCL carries the run or the hub does (distributed.init_transport)
            note = mvdist.init_transport(None if backend == "tcp" else model.engine, rank, world, prefer="tcp" if backend == "tcp" else "rccl")
            logger.info("statistics transport: %s", note)
        return _sharded_body(archive, model, input_file, input_golden_file, output_file, predictions_output_file, batch_size, rank, world)
    finally:
        if not use_torch:
            mvdist.shutdown()  # a later sharded call in this process starts from scratch (no stale communicator / sockets)


def _sharded_body(archive, model, input_file, input_golden_file, output_file, predictions_output_file, batch_size, rank, world):
    from . import distributed as mvdist

    golden_samples = list(archive.validation_dataset_reader.read(input_golden_file))
    model._golden_instances_embeddings = None
    model._golden_instances_labels = None
    for s0 in range(0, len(golden_samples), 128):
        model.forward_on_instances(golden_samples[s0:s0 + 128])

    arrays = archive.dataset_reader.read_arrays(input_file, workers=int(os.environ.get("MEMVUL_TOKENIZER_WORKERS", "0")),
                                                shard=(rank, world))
    n = len(arrays["lens"])
    part = f"{predictions_output_file}.part{rank}" if predictions_output_file and world > 1 else predictions_output_file
    best = np.zeros((n, 2), np.float32)
    step = 32 * batch_size
    writer = None
    if part:
        from .records import RecordWriter

        writer = RecordWriter(part, model._golden_labels, workers=int(os.environ.get("MEMVUL_RECORD_WORKERS", "0")))
    try:
        for s0 in range(0, n, step):
            s1 = min(n, s0 + step)
            b, _, p_same = model.sweep_arrays(arrays, s0, s1, batch_size, with_probs=writer is not None)
            best[s0:s1] = b
            if writer is not None:
                for b0 in range(s0, s1, batch_size):
                    b1 = min(s1, b0 + batch_size)
                    writer.submit(arrays["urls"][b0:b1], arrays["labels"][b0:b1], p_same[b0 - s0:b1 - s0])
    finally:
        if writer is not None:
            writer.close()
    rows = mvdist.all_gather_rows(np.concatenate([best, np.asarray(arrays["same"], np.float32)[:, None]], 1))
    # metrics of the whole set from the gathered per-IR rows, through the same accumulators a single process feeds
    model._siamese_metric.reset()
    model._counts.reset()
    same = rows[:, 2] > 0.5
    diff_idx = model.vocab.get_token_index("diff", namespace=model._label_namespace)
    model._counts(rows[:, :2], np.where(same, model._same_idx, diff_idx).astype(np.int64))
    model._siamese_metric.add_arrays(same.astype(np.uint8), rows[:, model._same_idx])
    metrics = model.get_metrics(reset=True)
    if output_file and rank == 0:
        with open(output_file, "w") as f:
            json.dump(_jsonable(metrics), f, indent=4)
    mvdist.barrier()  # every part is closed
    if predictions_output_file and world > 1 and rank == 0:
        # shards are contiguous: the parts in rank order are the single-process predictions file (what cal_metrics,
        # predict_memory.py:159-165, reads); the parts are removed once they are in it
        with open(predictions_output_file, "wb") as out:
            for r in range(world):
                with open(f"{predictions_output_file}.part{r}", "rb") as f:
                    shutil.copyfileobj(f, out)
        for r in range(world):
            os.remove(f"{predictions_output_file}.part{r}")
    mvdist.barrier()
    return metrics


test_siamese_sharded.__test__ = False  # not a pytest test



def model_measure(test_label, pred, pred_score, sample_id=None):
    """predict_memory.py:117-156 on arrays: confusion counts, P/R/F1, ROC-AUC, AP."""
    from sklearn import metrics

    y = np.asarray(test_label).astype(np.int64)
    p = np.asarray(pred).astype(np.int64)
    TP = int(np.sum((y == 1) & (p == 1)))
    FN = int(np.sum((y == 1) & (p != 1)))
    TN = int(np.sum((y == 0) & (p == 0)))
    FP = int(np.sum((y == 0) & (p != 0)))
    pd = prec = f_measure = 0
    if TP + FN != 0:
        pd = TP / (TP + FN)
    if TP + FP != 0:
        prec = TP / (TP + FP)
    if pd + prec != 0:
        f_measure = 2 * pd * prec / (pd + prec)
    fpr, tpr, _ = metrics.roc_curve(y, pred_score, pos_label=1)
    auc = metrics.auc(fpr, tpr)
    ap = metrics.average_precision_score(y, pred_score, pos_label=1)
    result = {"TP": TP, "FN": FN, "TN": TN, "FP": FP, "pd&recall": pd, "prec": prec, "f1": f_measure, "ap": ap, "auc": auc}
    return result, fpr, tpr


def measure_arrays(best_same: np.ndarray, labels01: np.ndarray, thres: float = 0.5) -> Dict[str, Any]:
    """``cal_metrics`` without the JSON round trip: per-IR score = max_g P(same) = the best-anchor
    P(same) (predict_memory.py:170-171), positive iff ``score >= thres`` (l.174-177)."""
    score = np.asarray(best_same, dtype=np.float64)
    pred = (score >= thres).astype(np.int64)
    m, _, _ = model_measure(labels01, pred, score)
    m["thres"] = thres
    return m


def cal_metrics(file, thres=0.5, data_path: Optional[str] = None):
    """predict_memory.py:159-197: second pass over ``{data_path}/test_results/{file}.json``."""
    data_path = DATA_PATH if data_path is None else data_path
    merged_results = []
    with open(f"{data_path}/test_results/{file}.json", "r") as f:
        for line in f:  # results of multiple batches are segmented by \n
            merged_results.extend(json.loads(line))
    score = np.array([np.max(list(s["predict"].values())) for s in merged_results], np.float64)
    label = np.array([0 if s["label"] == "neg" else 1 for s in merged_results], np.int64)
    pred = (score >= thres).astype(np.int64)
    metrics, fpr, tpr = model_measure(label, pred, score, [s["Issue_Url"] for s in merged_results])
    fn = file.split("_")[:-1]
    fn.append("metric_all")
    fn = "_".join(fn)
    metrics["thres"] = thres
    with open(f"{data_path}/test_results/{fn}.json", "w") as f:
        json.dump(_jsonable(metrics), f, indent=4)
    return metrics
