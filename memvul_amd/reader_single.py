"""``reader_single`` — the DatasetReader of MemVul-m (reference: MemVul/reader_single.py:31-139).

Kept from the reference, by line: constructor arguments (l.34-40); ``read_dataset`` tokenises
``"{Issue_Title}. {Issue_Body}"`` (l.63), labels pos/neg from ``str(s[target]) == "1"`` (l.65-66), groups the
records by label in first-appearance order and caches per path (l.53-56,72); ``_read`` dispatches on the path
substring — ``"test_"`` -> type "unlabel" (l.90-95), ``"validation_"`` -> type "test" (l.97-101) — and emits the
records in concatenation order of the label groups (l.80-82; NOT reversed, unlike reader_memory);
``text_to_instance`` (l.125-139): fields ``sample`` (TextField), ``label`` in namespace ``class_labels`` and
``metadata = {"type", "instance": {"Issue_Url", "label"}}`` (a dict, not a list).
Out of scope: the shuffled / negative-sampled training branch (l.103-121).
"""
from __future__ import annotations

import json
import logging
from typing import Dict

from .data import Instance, LabelField, MetadataField, TextField
from .registry import DatasetReader, TokenIndexer, Tokenizer
from . import tokenizer as _tok  # noqa: F401  (registers "pretrained_transformer")

logger = logging.getLogger(__name__)


@DatasetReader.register("reader_single")
class ReaderSingle(DatasetReader):
    def __init__(self,
                 tokenizer: Tokenizer = None,
                 token_indexers: Dict[str, TokenIndexer] = None,
                 sample_neg: float = None,
                 train_iter: int = None,
                 cache_directory: str = None,
                 target: str = "Security_Issue_Full") -> None:
        super().__init__()
        self._token_indexers = token_indexers
        self._tokenizer = tokenizer
        self._target = target
        self._train_iter = train_iter or 1
        select_neg = sample_neg or 0.1
        self._select_neg = [select_neg, 1 - select_neg]
        self._dataset = dict()

    def read_dataset(self, file_path):
        if self._dataset.get(file_path):
            return self._dataset[file_path]
        with open(file_path, "r", encoding="utf-8") as f:
            samples = json.load(f)
        dataset = dict()
        for s in samples:
            s["description"] = self._tokenizer.tokenize(f"{s['Issue_Title']}. {s['Issue_Body']}")
            label = "pos" if str(s[self._target]) == "1" else "neg"
            s[self._target] = label
            dataset.setdefault(label, list()).append(s)
        self._dataset[file_path] = dataset
        return dataset

    def _read(self, file_path):
        dataset = self.read_dataset(file_path)
        all_data = []
        for ll in list(dataset.values()):
            all_data.extend(ll)
        logger.info({k: len(v) for k, v in dataset.items()})
        if "test_" in file_path:
            for sample in all_data:
                yield self.text_to_instance(sample, type_="unlabel")
        elif "validation_" in file_path:
            for sample in all_data:
                yield self.text_to_instance(sample, type_="test")
        else:
            raise NotImplementedError("the shuffled, negative-sampled training branch (reader_single.py:103-121) is outside "
                                      "the inference path; file names select the branch by substring: 'test_', 'validation_'")

    def text_to_instance(self, ins, type_="train") -> Instance:
        fields = dict()
        fields["sample"] = TextField(ins["description"], self._token_indexers)
        fields["label"] = LabelField(ins[self._target], label_namespace="class_labels")
        fields["metadata"] = MetadataField({"type": type_, "instance": {"Issue_Url": ins["Issue_Url"], "label": ins[self._target]}})
        return Instance(fields)
