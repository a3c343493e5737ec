"""allennlp/data/fields (subset): TextField, LabelField, MetadataField; ListField / SequenceLabelField are import
surface only."""
from typing import Any, Dict, List

import torch


class Field:
    def index(self, vocab):
        pass

    def get_padding_lengths(self) -> Dict[str, int]:
        return {}

    def as_tensor(self, padding_lengths):
        raise NotImplementedError

    def batch_tensors(self, tensor_list):
        return torch.stack(tensor_list)


class TextField(Field):
    def __init__(self, tokens, token_indexers=None) -> None:
        self.tokens = tokens
        self._token_indexers = token_indexers
        self._indexed_tokens = None

    @property
    def token_indexers(self):
        return self._token_indexers

    def index(self, vocab):
        self._indexed_tokens = {name: ix.tokens_to_indices(self.tokens, vocab) for name, ix in self._token_indexers.items()}

    def get_padding_lengths(self) -> Dict[str, int]:
        out = {}
        for name, ix in self._token_indexers.items():
            for key, n in ix.get_padding_lengths(self._indexed_tokens[name]).items():
                out[f"{name}___{key}"] = n
        return out

    def as_tensor(self, padding_lengths):
        tensors = {}
        for name, ix in self._token_indexers.items():
            lens = {k.split("___", 1)[1]: v for k, v in padding_lengths.items() if k.startswith(name + "___")}
            tensors[name] = ix.as_padded_tensor_dict(self._indexed_tokens[name], lens)
        return tensors

    def batch_tensors(self, tensor_list):
        out = {}
        for name in tensor_list[0]:
            out[name] = {key: torch.stack([t[name][key] for t in tensor_list]) for key in tensor_list[0][name]}
        return out

    def __len__(self):
        return len(self.tokens)


class LabelField(Field):
    def __init__(self, label, label_namespace: str = "labels", skip_indexing: bool = False) -> None:
        self.label = label
        self._label_namespace = label_namespace
        self._label_id = label if skip_indexing else None

    def index(self, vocab):
        if self._label_id is None:
            self._label_id = vocab.get_token_index(self.label, self._label_namespace)

    def as_tensor(self, padding_lengths):
        return torch.tensor(self._label_id, dtype=torch.long)


class MetadataField(Field):
    def __init__(self, metadata: Any) -> None:
        self.metadata = metadata

    def as_tensor(self, padding_lengths):
        return self.metadata

    def batch_tensors(self, tensor_list: List[Any]):
        return tensor_list


class ListField(Field):
    def __init__(self, field_list):
        self.field_list = field_list


class SequenceLabelField(Field):
    def __init__(self, labels, sequence_field, label_namespace="labels"):
        self.labels = labels
