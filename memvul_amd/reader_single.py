"""``reader_single`` — the DatasetReader of MemVul-m behind the reference's plugin API (MemVul/reader_single.py:31-139).

What a caller of the reference class can rely on is kept — the registered name, the constructor arguments (l.34-40), the
three public methods and what they return:

* ``read_dataset(path)``  ->  ``{"pos" | "neg": [record, ...]}`` in first-appearance order of the labels, each record
  carrying its tokenised ``"{Issue_Title}. {Issue_Body}"`` under ``"description"`` and its label under ``target`` (l.53-72);
  parsed files are cached per path;
* ``_read(path)``: the path's substring selects the evaluation branch — ``"test_"`` emits type ``"unlabel"`` (l.90-95),
  ``"validation_"`` type ``"test"`` (l.97-101) — and the records come out group after group in concatenation order
  (l.80-82; NOT reversed, unlike reader_memory);
* ``text_to_instance(record, type_)``: fields ``sample`` (TextField), ``label`` in namespace ``class_labels`` and
  ``metadata = {"type", "instance": {"Issue_Url", "label"}}`` (a dict, not a list; l.125-139).

The implementation is this package's own: one table of evaluation branches and the whole file tokenised in ONE batched
call when the tokenizer offers ``batch_tokenize`` (the WordPiece path spends its time per call).
Out of scope: the shuffled / negative-sampled training branch (l.103-121).
"""
from __future__ import annotations

import json
import logging
from itertools import chain
from typing import Dict, Iterator, List, Optional

from . import tokenizer as _tok  # noqa: F401  (registers "pretrained_transformer")
from .data import Instance, LabelField, MetadataField, TextField
from .registry import DatasetReader, TokenIndexer, Tokenizer

logger = logging.getLogger(__name__)

#: path substring -> ``metadata["type"]`` of the instances that file yields (reader_single.py:90, 97)
EVALUATION_BRANCHES = (("test_", "unlabel"), ("validation_", "test"))


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
        self._tokenizer, self._token_indexers, self._target = tokenizer, token_indexers, target
        # kept for config compatibility; only the (out-of-scope) training branch consumes them
        self._train_iter = train_iter or 1
        p_keep = sample_neg or 0.1
        self._select_neg = [p_keep, 1 - p_keep]
        self._dataset: Dict[str, Dict[str, List[dict]]] = {}

    # ---- parsing -------------------------------------------------------------------------------------------------
    def _tokenize_all(self, texts: List[str]):
        batch = getattr(self._tokenizer, "batch_tokenize", None)
        return batch(texts) if batch is not None else [self._tokenizer.tokenize(t) for t in texts]

    def read_dataset(self, file_path) -> Dict[str, List[dict]]:
        cached = self._dataset.get(file_path)
        if cached:
            return cached
        with open(file_path, "r", encoding="utf-8") as f:
            records = json.load(f)
        tokens = self._tokenize_all([f"{r['Issue_Title']}. {r['Issue_Body']}" for r in records])
        groups: Dict[str, List[dict]] = {}
        for rec, toks in zip(records, tokens):
            rec["description"] = toks
            rec[self._target] = "pos" if str(rec[self._target]) == "1" else "neg"
            groups.setdefault(rec[self._target], []).append(rec)
        self._dataset[file_path] = groups
        return groups

    @staticmethod
    def branch_of(file_path) -> Optional[str]:
        return next((type_ for key, type_ in EVALUATION_BRANCHES if key in file_path), None)

    # ---- the AllenNLP reader protocol ----------------------------------------------------------------------------
    def _read(self, file_path) -> Iterator[Instance]:
        type_ = self.branch_of(file_path)
        if type_ is None:
            raise NotImplementedError("the shuffled, negative-sampled training branch (reader_single.py:103-121) is outside "
                                      "the inference path; file names select the branch by substring: 'test_', 'validation_'")
        groups = self.read_dataset(file_path)
        logger.info({label: len(recs) for label, recs in groups.items()})
        for rec in chain.from_iterable(groups.values()):
            yield self.text_to_instance(rec, type_=type_)

    def text_to_instance(self, ins, type_="train") -> Instance:
        label = ins[self._target]
        return Instance({
            "sample": TextField(ins["description"], self._token_indexers),
            "label": LabelField(label, label_namespace="class_labels"),
            "metadata": MetadataField({"type": type_, "instance": {"Issue_Url": ins["Issue_Url"], "label": label}}),
        })
