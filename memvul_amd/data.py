"""Minimal AllenNLP data surface used by the hot path: ``Token``, ``TextField``, ``LabelField``,
``MetadataField``, ``Instance``, pad-to-longest collation into ``TextFieldTensors`` and a sequential
``DataLoader`` (reference call sites: reader_memory.py:195-246, predict_memory.py:92-101).

Tensors are numpy arrays (the engine takes host int32 buffers through the C ABI; torch is not needed):
``{"tokens": {"token_ids": int64[B,S], "mask": bool[B,S], "type_ids": int64[B,S]}}`` — the same keys and
dtypes AllenNLP's ``PretrainedTransformerIndexer`` produces.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Dict, Iterable, Iterator, List, Optional

import numpy as np

from .registry import HAVE_ALLENNLP

if HAVE_ALLENNLP:
    # AllenNLP is importable: the reader emits AllenNLP's OWN Instances / Fields, so AllenNLP's DataLoader, Batch and
    # `evaluate` (predict_memory.py:92-110) can index and collate them (registry.py; tests/test_reference_driver_over_plugin.py)
    from allennlp.data import Instance, Token  # type: ignore
    from allennlp.data.fields import Field, LabelField, MetadataField, TextField  # type: ignore
else:
    @dataclass
    class Token:
        text: str = None
        text_id: Optional[int] = None
        type_id: Optional[int] = None

    class Field:
        pass

    class TextField(Field):
        def __init__(self, tokens: List[Token], token_indexers: Optional[Dict[str, Any]] = None) -> None:
            self.tokens = tokens
            self._token_indexers = token_indexers

        def __len__(self):
            return len(self.tokens)

    class LabelField(Field):
        def __init__(self, label: str, label_namespace: str = "labels") -> None:
            self.label = label
            self._label_namespace = label_namespace

    class MetadataField(Field):
        def __init__(self, metadata: Any) -> None:
            self.metadata = metadata

    class Instance:
        def __init__(self, fields: Dict[str, Field]) -> None:
            self.fields = fields

        def __getitem__(self, k):
            return self.fields[k]


def collate(instances: List[Instance], vocab=None) -> Dict[str, Any]:
    """``allennlp_collate``: every TextField padded with 0 to the longest in the batch, labels indexed
    through the ``labels`` namespace, metadata passed through as a list.  (Same arrays from AllenNLP's own Instances and
    from the stand-in's: only ``tokens`` / ``text_id`` / ``type_id`` / ``label`` / ``metadata`` are read.)"""
    out: Dict[str, Any] = {}
    names = list(instances[0].fields.keys())
    for name in names:
        f0 = instances[0].fields[name]
        if isinstance(f0, TextField):
            L = max(len(ins.fields[name]) for ins in instances)
            B = len(instances)
            ids = np.zeros((B, L), np.int64)
            typ = np.zeros((B, L), np.int64)
            mask = np.zeros((B, L), bool)
            lens = np.zeros(B, np.int32)
            single = True
            for b, ins in enumerate(instances):
                toks = ins.fields[name].tokens
                n = len(toks)
                row = getattr(toks, "ids", None)
                if row is not None:  # tokenizer.TokenRow: the id array itself (single-segment: type ids 0)
                    ids[b, :n] = row
                else:
                    ids[b, :n] = [t.text_id for t in toks]
                    tt = [t.type_id or 0 for t in toks]
                    typ[b, :n] = tt
                    single = single and not any(tt)
                mask[b, :n] = True
                lens[b] = n
            # "_collated": what this function knows by construction (a prefix mask of these lengths, zero padding, single segment or not) — ModelMemory._ids_lens
            # takes it (with the int32 copy of the ids the engine wants) instead of re-deriving it from the arrays on the scoring thread (ten numpy passes over [B, L], each a GIL hand-over: 11 ms per batch of
            # 512 next to two other Python threads, profiles/r06_*_e2e_dropin.txt)
            out[name] = {"tokens": {"token_ids": ids, "mask": mask, "type_ids": typ, "_collated": (ids.astype(np.int32), lens, single)}}
        elif isinstance(f0, LabelField):
            if vocab is None:
                raise ValueError("a Vocabulary is needed to index LabelFields (DataLoader.index_with)")
            out[name] = np.array([vocab.get_token_index(ins.fields[name].label, ins.fields[name]._label_namespace) for ins in instances], np.int64)
        elif isinstance(f0, MetadataField):
            out[name] = [ins.fields[name].metadata for ins in instances]
    return out


class DataLoader:
    """Sequential loader (``"shuffle": false`` in every inference config, test_config_memory.json:22-25)."""

    def __init__(self, reader=None, data_path: str = None, batch_size: int = 32, shuffle: bool = False, instances=None, **_ignored) -> None:
        if shuffle:
            raise ValueError("the inference path never shuffles (test_config_memory.json:24)")
        self.reader, self.data_path, self.batch_size = reader, data_path, int(batch_size)
        self._instances = instances
        self._vocab = None

    @classmethod
    def from_params(cls, params: Dict[str, Any], reader=None, data_path: str = None):
        return cls(reader=reader, data_path=data_path, **dict(params))

    def index_with(self, vocab) -> None:
        self._vocab = vocab

    def iter_instances(self) -> Iterable[Instance]:
        if self._instances is None:
            self._instances = list(self.reader.read(self.data_path))
        return self._instances

    def __iter__(self) -> Iterator[Dict[str, Any]]:
        if self._instances is None and self.reader is not None:
            # first pass: batches are collated as the reader produces the Instances (ReaderMemory streams its tokenisation), the list is kept for later passes
            acc, cur = [], []
            for ins in self.reader.read(self.data_path):
                acc.append(ins)
                cur.append(ins)
                if len(cur) == self.batch_size:
                    yield collate(cur, self._vocab)
                    cur = []
            if cur:
                yield collate(cur, self._vocab)
            self._instances = acc
            return
        ins = self.iter_instances()
        for s in range(0, len(ins), self.batch_size):
            yield collate(ins[s : s + self.batch_size], self._vocab)

    def __len__(self):
        n = len(self.iter_instances())
        return (n + self.batch_size - 1) // self.batch_size
