"""allennlp/training/util.py (subset): evaluate() — the hot loop predict_memory.py:103 drives."""
import json
import logging
from typing import Any, Dict

import torch

from allennlp.common.checks import check_for_gpu
from allennlp.common.tqdm import Tqdm
from allennlp.common.util import dump_metrics, int_to_device, sanitize
from allennlp.nn import util as nn_util

logger = logging.getLogger(__name__)


def get_batch_size(batch) -> int:
    if isinstance(batch, torch.Tensor):
        return batch.size(0)
    if isinstance(batch, dict):
        return get_batch_size(next(iter(batch.values())))
    return 0


def evaluate(model, data_loader, cuda_device: int = -1, batch_weight_key: str = None, output_file: str = None,
             predictions_output_file: str = None) -> Dict[str, Any]:
    check_for_gpu(cuda_device)
    data_loader.set_target_device(int_to_device(cuda_device))
    predictions_file = None if predictions_output_file is None else open(predictions_output_file, "w")
    with torch.no_grad():
        model.eval()
        iterator = iter(data_loader)
        logger.info("Iterating over dataset")
        generator_tqdm = Tqdm.tqdm(iterator)
        batch_count = 0
        loss_count = 0
        total_loss = 0.0
        total_weight = 0.0
        for batch in generator_tqdm:
            batch_count += 1
            batch = nn_util.move_to_device(batch, cuda_device)
            output_dict = model(**batch)
            loss = output_dict.get("loss")
            metrics = model.get_metrics()
            if loss is not None:
                loss_count += 1
                weight = output_dict[batch_weight_key].item() if batch_weight_key else 1.0
                total_weight += weight
                total_loss += loss.item() * weight
                metrics["loss"] = total_loss / total_weight
            if predictions_file is not None:
                predictions = json.dumps(sanitize(model.make_output_human_readable(output_dict)))
                predictions_file.write(predictions + "\n")
        if predictions_file is not None:
            predictions_file.close()
        final_metrics = model.get_metrics(reset=True)
        if loss_count > 0:
            if loss_count != batch_count:
                raise RuntimeError("The model you are trying to evaluate only sometimes produced a loss!")
            final_metrics["loss"] = total_loss / total_weight
        if output_file is not None:
            dump_metrics(output_file, final_metrics, log=True)
        return final_metrics
