"""``pretrained_transformer`` tokenizer / indexer stand-ins (AllenNLP names used by the reference
configs: test_config_memory.json:5-16, config_memory.json:15-27).

Behaviour kept from AllenNLP's ``PretrainedTransformerTokenizer``: lower-cased WordPiece of
``bert-base-uncased`` with ``[CLS] ... [SEP]`` added and truncation to ``max_length`` INCLUDING the two
special tokens (256 for issue reports, 512 for anchors).  The real WordPiece vocabulary is used when a
``vocab.txt`` is reachable (``model_name`` is a directory holding one, or ``$MEMVUL_BERT_VOCAB``); it is
not on disk in this image and there is no network, so otherwise a deterministic hashing word tokenizer
with the same id range / special ids is used — enough for plumbing and throughput runs on synthetic
text, not for real accuracy numbers.
"""
from __future__ import annotations

import logging
import os
import re
import zlib
from typing import Dict, List, Optional

from .data import Token
from .registry import Tokenizer, TokenIndexer, register_builtin

logger = logging.getLogger(__name__)
CLS_ID, SEP_ID, PAD_ID, UNK_ID = 101, 102, 0, 100
_WORD_RE = re.compile(r"[a-z0-9]+|[^\sa-z0-9]")


def _find_vocab(model_name: str) -> Optional[str]:
    cands = [os.environ.get("MEMVUL_BERT_VOCAB")]
    if model_name and os.path.isdir(model_name):
        cands.append(os.path.join(model_name, "vocab.txt"))
    for c in cands:
        if c and os.path.isfile(c):
            return c
    return None


class TokenRow:
    """The tokens of one text as a sequence over its id array: ``len``, iteration and indexing give ``Token`` objects (built when asked for: with
    the WordPiece backend their ``text`` is looked up then), ``ids`` gives the int32 array the collation reads directly — a file of 40 k issue
    reports is 8 M tokens, and one Python object per token costs more than the tokenisation itself."""

    __slots__ = ("ids", "_hf")

    def __init__(self, ids, hf=None):
        self.ids, self._hf = ids, hf

    def __len__(self):
        return len(self.ids)

    def _tok(self, i: int) -> Token:
        return Token(self._hf.convert_ids_to_tokens(i) if self._hf is not None else str(i), i, 0)

    def __getitem__(self, k):
        if isinstance(k, slice):
            return TokenRow(self.ids[k], self._hf)
        return self._tok(int(self.ids[k]))

    def __iter__(self):
        if self._hf is not None:
            ids = [int(i) for i in self.ids]
            return (Token(t, i, 0) for t, i in zip(self._hf.convert_ids_to_tokens(ids), ids))
        return (Token(str(int(i)), int(i), 0) for i in self.ids)

    def __eq__(self, other):
        try:
            return len(self) == len(other) and all((a.text, a.text_id, a.type_id) == (b.text, b.text_id, b.type_id) for a, b in zip(self, other))
        except (TypeError, AttributeError):
            return NotImplemented


@register_builtin(Tokenizer, "pretrained_transformer")
class PretrainedTransformerTokenizer(Tokenizer):
    def __init__(self, model_name: str = "bert-base-uncased", add_special_tokens: bool = True, max_length: Optional[int] = None,
                 tokenizer_kwargs: Optional[Dict] = None, vocab_size: int = 30522) -> None:
        self.model_name, self._add_special, self._max_length = model_name, add_special_tokens, max_length
        self.vocab_size = vocab_size
        self._hf = None
        vocab = _find_vocab(model_name)
        if vocab is None:
            # a cached / local HuggingFace copy of `model_name` (what AllenNLP's tokenizer loads, reader_memory.py:88)
            try:
                from transformers import BertTokenizerFast

                hf = BertTokenizerFast.from_pretrained(model_name, local_files_only=True, **(tokenizer_kwargs or {}))
                # transformers 5.x hands back a 5-token default vocabulary instead of raising when nothing is cached
                if hf.vocab_size >= 1000 and hf.cls_token_id is not None and hf.sep_token_id is not None:
                    self._hf = hf
                    self.vocab_size = hf.vocab_size
            except Exception:
                self._hf = None
            if self._hf is None:
                if os.environ.get("MEMVUL_ALLOW_HASH_TOKENIZER") != "1":
                    raise RuntimeError(
                        f"no WordPiece vocabulary for {model_name!r}: point $MEMVUL_BERT_VOCAB at its vocab.txt (or make model_name "
                        "a directory holding one, or have it in the local HuggingFace cache).  The CRC32 hashing stand-in feeds "
                        "meaningless ids to trained weights; it is only for synthetic plumbing / throughput runs and must be "
                        "asked for with MEMVUL_ALLOW_HASH_TOKENIZER=1.")
                logger.warning("MEMVUL_ALLOW_HASH_TOKENIZER=1: %r is tokenised by the CRC32 hashing stand-in (synthetic runs only)", model_name)
        if vocab is not None:
            import inspect

            from transformers import BertTokenizerFast

            with open(vocab, "r", encoding="utf-8") as f:
                table = {line.rstrip("\n"): i for i, line in enumerate(f)}
            # transformers 4.x (the reference pins 4.1.0) takes ``vocab_file``; 5.x takes ``vocab`` and silently IGNORES
            # ``vocab_file`` (a 5-token default vocabulary comes back) — so pick by signature and check the result
            if "vocab" in inspect.signature(BertTokenizerFast.__init__).parameters:
                self._hf = BertTokenizerFast(vocab=table, do_lower_case=True, **(tokenizer_kwargs or {}))
            else:
                self._hf = BertTokenizerFast(vocab_file=vocab, do_lower_case=True, **(tokenizer_kwargs or {}))
            if self._hf.vocab_size != len(table) or self._hf.cls_token_id != table.get("[CLS]") or self._hf.sep_token_id != table.get("[SEP]"):
                raise RuntimeError(f"WordPiece vocabulary {vocab} was not taken over by BertTokenizerFast "
                                   f"({self._hf.vocab_size} tokens loaded, {len(table)} in the file)")
            self.vocab_size = self._hf.vocab_size

    def _hash_ids(self, text: str) -> List[int]:
        lo = 1000 if self.vocab_size > 2000 else 3
        return [lo + zlib.crc32(w.encode("utf-8")) % (self.vocab_size - lo) for w in _WORD_RE.findall(text.lower())]

    def _hash_encode(self, text: str) -> List[int]:
        ids = self._hash_ids(text)
        if self._add_special:
            if self._max_length is not None:
                ids = ids[: max(0, self._max_length - 2)]
            return [CLS_ID] + ids + [SEP_ID]
        return ids[: self._max_length] if self._max_length is not None else ids

    def batch_ids(self, texts: List[str], workers: int = 0):
        """Array form of ``tokenize`` for a whole file: ``(ids int32 [N, L] zero-padded, lens int32 [N])`` with exactly the
        ids ``tokenize`` gives text by text.  The WordPiece path is one batched call into the Rust tokenizer (parallel
        inside); the hashing stand-in can fan out over ``workers`` forked processes."""
        import numpy as np

        bt = getattr(self._hf, "backend_tokenizer", None) if self._hf is not None else None
        if bt is not None and len(texts):
            # the Rust tokenizer itself, not the transformers wrapper around it: the wrapper turns every encoding into dicts of Python lists under the interpreter
            # lock (1.3 s of a 1.8 s call for 8 k reports; the same ids: tests/test_plumbing.py) — here the ids go from the encodings into ONE flat int32 array and from
            # there into the padded matrix, no Python object per token or per text
            import itertools

            if self._max_length is not None:
                bt.enable_truncation(max_length=self._max_length)  # (longest_first, right, stride 0: what truncation=True asks of the wrapper)
            else:
                bt.no_truncation()
            bt.no_padding()
            encs = bt.encode_batch(list(texts), add_special_tokens=self._add_special)
            lens = np.fromiter((len(e) for e in encs), dtype=np.int32, count=len(encs))
            flat = np.fromiter(itertools.chain.from_iterable(e.ids for e in encs), dtype=np.int32, count=int(lens.sum()))
            L = int(lens.max())
            ids = np.zeros((len(encs), L), np.int32)
            ids[np.arange(L)[None, :] < lens[:, None]] = flat
            return ids, lens
        if self._hf is not None:
            rows = self._hf(list(texts), add_special_tokens=self._add_special, truncation=self._max_length is not None,
                            max_length=self._max_length, return_attention_mask=False, return_token_type_ids=False)["input_ids"]
        elif workers and workers > 1 and len(texts) >= 4 * workers:
            import multiprocessing as mp

            step = (len(texts) + workers - 1) // workers
            with mp.get_context("fork").Pool(workers) as pool:
                parts = pool.map(self._hash_encode_many, [texts[i:i + step] for i in range(0, len(texts), step)])
            rows = [r for part in parts for r in part]
        else:
            rows = self._hash_encode_many(texts)
        lens = np.fromiter((len(r) for r in rows), dtype=np.int32, count=len(rows))
        ids = np.zeros((len(rows), int(lens.max()) if len(rows) else 0), np.int32)
        for i, r in enumerate(rows):
            ids[i, :len(r)] = r
        return ids, lens

    def _hash_encode_many(self, texts: List[str]) -> List[List[int]]:
        return [self._hash_encode(t) for t in texts]

    def batch_token_rows(self, texts: List[str]) -> List["TokenRow"]:
        """``batch_tokenize`` without the Token objects: one TokenRow per text (same tokens, same ids, materialised on demand)."""
        ids, lens = self.batch_ids(texts)
        return [TokenRow(ids[i, :lens[i]], self._hf) for i in range(len(texts))]

    def batch_tokenize(self, texts: List[str]) -> List[List[Token]]:
        """``[tokenize(t) for t in texts]`` with ONE call into the tokenizer backend (same tokens, same ids)."""
        ids, lens = self.batch_ids(texts)
        rows = [ids[i, :lens[i]].tolist() for i in range(len(texts))]
        if self._hf is not None:
            return [[Token(t, i, 0) for t, i in zip(self._hf.convert_ids_to_tokens(r), r)] for r in rows]
        return [[Token(str(i), i, 0) for i in r] for r in rows]

    def tokenize(self, text: str) -> List[Token]:
        if self._hf is not None:
            enc = self._hf(text, add_special_tokens=self._add_special, truncation=self._max_length is not None,
                           max_length=self._max_length, return_attention_mask=False, return_token_type_ids=False)
            ids = enc["input_ids"]
            texts = self._hf.convert_ids_to_tokens(ids)
            return [Token(t, i, 0) for t, i in zip(texts, ids)]
        return [Token(str(i), i, 0) for i in self._hash_encode(text)]


@register_builtin(TokenIndexer, "pretrained_transformer")
class PretrainedTransformerIndexer(TokenIndexer):
    """Tokens already carry their wordpiece ids; indexing is the identity (``namespace`` is only where
    AllenNLP would mirror the HF vocabulary)."""

    def __init__(self, model_name: str = "bert-base-uncased", namespace: str = "tags", max_length: Optional[int] = None, **_kw) -> None:
        self.model_name, self._namespace, self._max_length = model_name, namespace, max_length

    def tokens_to_indices(self, tokens: List[Token], vocabulary=None) -> Dict[str, List]:
        return {"token_ids": [t.text_id for t in tokens], "mask": [True] * len(tokens), "type_ids": [t.type_id or 0 for t in tokens]}
