"""allennlp/models/model.py (subset): registry, forward_on_instances, load from an archive directory."""
import logging
import os
import re
from typing import Dict, List

import torch

from allennlp.common import Params, Registrable
from allennlp.data import Batch, Instance, Vocabulary
from allennlp.nn import util

logger = logging.getLogger(__name__)
_DEFAULT_WEIGHTS = "best.th"


class Model(torch.nn.Module, Registrable):
    _warn_for_unseparable_batches = set()

    def __init__(self, vocab: Vocabulary, regularizer=None, serialization_dir=None) -> None:
        super().__init__()
        self.vocab = vocab
        self._regularizer = regularizer
        self.serialization_dir = serialization_dir

    def get_metrics(self, reset: bool = False) -> Dict[str, float]:
        return {}

    def make_output_human_readable(self, output_dict):
        return output_dict

    def _get_prediction_device(self) -> int:
        devices = {p.device.index if p.device.type != "cpu" else -1 for p in self.parameters()}
        return devices.pop() if len(devices) == 1 else -1

    def forward_on_instance(self, instance: Instance):
        return self.forward_on_instances([instance])[0]

    def forward_on_instances(self, instances: List[Instance]) -> List[Dict]:
        batch_size = len(instances)
        with torch.no_grad():
            cuda_device = self._get_prediction_device()
            dataset = Batch(instances)
            dataset.index_instances(self.vocab)
            model_input = util.move_to_device(dataset.as_tensor_dict(), cuda_device)
            outputs = self.make_output_human_readable(self(**model_input))
            instance_separated_output = [{} for _ in dataset.instances]
            for name, output in list(outputs.items()):
                if isinstance(output, torch.Tensor):
                    if output.dim() == 0:
                        output = output.unsqueeze(0)
                    if output.size(0) != batch_size:
                        continue
                    output = output.detach().cpu().numpy()
                elif len(output) != batch_size:
                    continue
                for instance_output, batch_element in zip(instance_separated_output, output):
                    instance_output[name] = batch_element
            return instance_separated_output

    def extend_embedder_vocab(self, *a, **kw):
        pass

    @classmethod
    def _load(cls, config: Params, serialization_dir: str, weights_file: str = None, cuda_device: int = -1) -> "Model":
        weights_file = weights_file or os.path.join(serialization_dir, _DEFAULT_WEIGHTS)
        vocab_dir = os.path.join(serialization_dir, "vocabulary")
        vocab_params = config.get("vocabulary", Params({}))
        vocab = Vocabulary.from_files(vocab_dir, vocab_params.get("padding_token") or "@@PADDING@@",
                                      vocab_params.get("oov_token") or "@@UNKNOWN@@")
        model_params = config.get("model")
        model = Model.from_params(vocab=vocab, params=model_params, serialization_dir=serialization_dir)
        if cuda_device >= 0:
            model.cuda(cuda_device)
        else:
            model.cpu()
        model.extend_embedder_vocab()
        model_state = torch.load(weights_file, map_location="cpu" if cuda_device < 0 else f"cuda:{cuda_device}")
        missing_keys, unexpected_keys = model.load_state_dict(model_state, strict=False)

        def filter_out_authorized_missing_keys(module, prefix=""):
            nonlocal missing_keys
            for pat in getattr(module.__class__, "authorized_missing_keys", None) or []:
                missing_keys = [k for k in missing_keys if not (k.startswith(prefix) and re.search(pat, k[len(prefix):]))]
            for name, child in module._modules.items():
                if child is not None:
                    filter_out_authorized_missing_keys(child, prefix + name + ".")

        filter_out_authorized_missing_keys(model)
        if unexpected_keys or missing_keys:
            raise RuntimeError(f"Error loading state dict for {model.__class__.__name__}\n\tMissing keys: {missing_keys}\n\t"
                               f"Unexpected keys: {unexpected_keys}")
        return model

    @classmethod
    def load(cls, config: Params, serialization_dir: str, weights_file: str = None, cuda_device: int = -1) -> "Model":
        model_type = config["model"] if isinstance(config["model"], str) else config["model"]["type"]
        model_class = cls.by_name(model_type)
        return model_class._load(config, serialization_dir, weights_file, cuda_device)
