from allennlp.common import Registrable


class DatasetReader(Registrable):
    """allennlp/data/dataset_readers/dataset_reader.py (subset): read() = the instances _read() yields, in order."""

    def __init__(self, max_instances=None, manual_distributed_sharding=False, manual_multiprocess_sharding=False,
                 serialization_dir=None) -> None:
        self.max_instances = max_instances

    def read(self, file_path):
        n = 0
        for ins in self._read(str(file_path)):
            if self.max_instances is not None and n >= self.max_instances:
                break
            n += 1
            yield ins

    def _read(self, file_path):
        raise NotImplementedError

    def text_to_instance(self, *inputs):
        raise NotImplementedError
