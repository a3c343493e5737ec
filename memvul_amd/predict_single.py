"""The inference driver of MemVul-m (reference: predict_single.py:49-131): ``test`` and ``cal_metrics`` with the
reference's names, arguments and file formats; the evaluation loop is the same ``evaluate`` predict_memory uses."""
from __future__ import annotations

import json
import logging
import os
from typing import Any, Dict, Optional

import numpy as np

from .archive import load_archive
from .data import DataLoader
from .predict_memory import _jsonable, evaluate, model_measure

logger = logging.getLogger(__name__)

DATA_PATH = os.environ.get("MEMVUL_DATA_PATH", "xxx")  # predict_single.py:134


def test(archive_file, input_file, test_config=None, weights_file=None, output_file=None, predictions_output_file=None,
         batch_size=64, cuda_device=0, seed=2021, package="memvul_amd", batch_weight_key="", file_friendly_logging=False,
         engine_options=None) -> Dict[str, Any]:
    """predict_single.py:49-108."""
    archive = load_archive(archive_file, weights_file=weights_file, cuda_device=cuda_device, overrides=test_config or "",
                           engine_options=engine_options)
    config, model = archive.config, archive.model
    model.eval()
    dataset_reader = archive.dataset_reader
    logger.info("Reading evaluation data from %s", input_file)
    data_loader_params = dict(config.get("validation_data_loader") or config.get("data_loader") or {})
    if batch_size:
        data_loader_params["batch_size"] = batch_size
    data_loader = DataLoader.from_params(params=data_loader_params, reader=dataset_reader, data_path=input_file)
    data_loader.index_with(model.vocab)
    metrics = evaluate(model, data_loader, cuda_device, batch_weight_key, output_file=output_file,
                       predictions_output_file=predictions_output_file)
    logger.info("Finished evaluating.")
    return metrics


test.__test__ = False  # not a pytest test


def cal_metrics(file, data_path: Optional[str] = None):
    """predict_single.py:111-131: second pass over ``{data_path}/test_results/{file}.json``."""
    data_path = DATA_PATH if data_path is None else data_path
    merged_results = []
    with open(f"{data_path}/test_results/{file}.json", "r") as f:
        for line in f:
            merged_results.extend(json.loads(line))
    label_convert = {"pos": 1, "neg": 0}
    pred = np.array([label_convert[r["predict"]] for r in merged_results], np.int64)
    label = np.array([label_convert[r["label"]] for r in merged_results], np.int64)
    pred_score = np.array([r["prob"] for r in merged_results], np.float64)
    metrics, fpr, tpr = model_measure(label, pred, pred_score, [r["Issue_Url"] for r in merged_results])
    fn = file.split("_")[:-1]
    fn.append("metric_all")
    with open(f"{data_path}/test_results/{'_'.join(fn)}.json", "w") as f:
        json.dump(_jsonable(metrics), f, indent=4)
    return metrics
