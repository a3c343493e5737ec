"""allennlp/data/data_loaders (subset): the default ("multiprocess") loader with num_workers = 0 — instances in reader
order, consecutive batches of batch_size (shuffle=False on the reference's configs), pad-to-longest per batch."""
from typing import Dict

import torch

from allennlp.common import Registrable
from allennlp.data.instance import Batch

TensorDict = Dict[str, torch.Tensor]


class DataLoader(Registrable):
    default_implementation = "multiprocess"


@DataLoader.register("multiprocess")
class MultiProcessDataLoader(DataLoader):
    def __init__(self, reader, data_path: str, batch_size: int = None, drop_last: bool = False, shuffle: bool = False,
                 batch_sampler=None, batches_per_epoch: int = None, num_workers: int = 0, max_instances_in_memory: int = None,
                 start_method: str = "fork", cuda_device=None) -> None:
        assert batch_sampler is None and num_workers == 0
        self.reader, self.data_path = reader, data_path
        self.batch_size, self.drop_last, self.shuffle = batch_size, drop_last, shuffle
        self._instances = None
        self._vocab = None
        self.cuda_device = None

    def index_with(self, vocab):
        self._vocab = vocab

    def set_target_device(self, device):
        self.cuda_device = device

    def iter_instances(self):
        if self._instances is None:
            self._instances = list(self.reader.read(self.data_path))
        for ins in self._instances:
            if self._vocab is not None:
                ins.index_fields(self._vocab)
            yield ins

    def __iter__(self):
        instances = list(self.iter_instances())
        if self.shuffle:
            import random

            random.shuffle(instances)
        for s in range(0, len(instances), self.batch_size):
            chunk = instances[s:s + self.batch_size]
            if self.drop_last and len(chunk) < self.batch_size:
                break
            yield Batch(chunk).as_tensor_dict()

    def __len__(self):
        n = len(list(self.iter_instances()))
        return n // self.batch_size if self.drop_last else -(-n // self.batch_size)
