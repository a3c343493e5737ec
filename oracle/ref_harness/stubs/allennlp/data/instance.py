"""allennlp/data/instance.py + batch.py (subset): pad-to-longest collation."""
from collections import defaultdict
from typing import Dict, List


class Instance:
    def __init__(self, fields) -> None:
        self.fields = fields
        self.indexed = False

    def __getitem__(self, key):
        return self.fields[key]

    def index_fields(self, vocab):
        if not self.indexed:
            for f in self.fields.values():
                f.index(vocab)
            self.indexed = True

    def get_padding_lengths(self):
        return {name: f.get_padding_lengths() for name, f in self.fields.items()}

    def as_tensor_dict(self, padding_lengths=None):
        padding_lengths = padding_lengths or self.get_padding_lengths()
        return {name: f.as_tensor(padding_lengths[name]) for name, f in self.fields.items()}


class Batch:
    def __init__(self, instances) -> None:
        self.instances: List[Instance] = list(instances)

    def index_instances(self, vocab):
        for ins in self.instances:
            ins.index_fields(vocab)

    def get_padding_lengths(self) -> Dict[str, Dict[str, int]]:
        out: Dict[str, Dict[str, int]] = defaultdict(dict)
        for ins in self.instances:
            for fname, lens in ins.get_padding_lengths().items():
                for k, v in lens.items():
                    out[fname][k] = max(out[fname].get(k, 0), v)
        for ins in self.instances:
            for fname in ins.fields:
                out.setdefault(fname, {})
        return out

    def as_tensor_dict(self):
        lens = self.get_padding_lengths()
        per_field = defaultdict(list)
        for ins in self.instances:
            for fname, t in ins.as_tensor_dict(lens).items():
                per_field[fname].append(t)
        first = self.instances[0]
        return {fname: first.fields[fname].batch_tensors(ts) for fname, ts in per_field.items()}


def allennlp_collate(instances):
    return Batch(instances).as_tensor_dict()
