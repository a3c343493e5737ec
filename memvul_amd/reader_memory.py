"""``reader_memory`` — the DatasetReader of the hot path (reference: MemVul/reader_memory.py:35-245).

Kept from the reference, by line:
  * constructor arguments and defaults (l.38-45); ``sample_neg is None`` -> the "golden anchors only"
    reader of ``custom_validation`` / ``validation_dataset_reader`` which never opens CVE_dict.json
    (l.58-60); otherwise ``CVE_dict.json`` and the anchor file are loaded (l.62-68);
  * ``read_dataset``: golden files are ``{cwe_id: description}`` (l.73-79); issue files are a list of
    records tokenised as ``"{Issue_Title}. {Issue_Body}"`` (l.88), labelled pos/neg from
    ``str(s[target]) == "1"`` (l.91), positives keyed by the CWE id of their CVE (l.93-108) and dropped when
    that id is ``None`` (l.103-105); parsed files are cached per path (l.81-82,111);
  * ``_read``: branch by path substring — ``"golden_"`` (l.138), ``"test_"`` -> type "unlabel" (l.146),
    ``"validation_"`` -> type "test" (l.155); test/validation instances are emitted in REVERSED
    concatenation order, i.e. positives first (l.150-152);
  * ``text_to_instance`` for golden / test / unlabel (l.195-246): fields ``sample1``, ``label`` (same/diff)
    and ``metadata = {"type", "instance": [{"label", "Issue_Url"}]}``.
Out of scope: the online pair-sampling training branch (l.164-192, 205-229).
The CVE-description normalisation (``replace_tokens_simple``, l.96-99) only feeds training pairs and is
skipped: at test time a positive needs nothing but its CVE's ``CWE_ID``.
"""
from __future__ import annotations

import json
import logging
import os
from typing import Dict, Optional

import numpy as np

from .data import Instance, LabelField, MetadataField, TextField
from .registry import HAVE_ALLENNLP, DatasetReader, TokenIndexer, Tokenizer
from . import tokenizer as _tok  # noqa: F401  (registers "pretrained_transformer")

logger = logging.getLogger(__name__)

# The reference hard-codes ``data_path = "xxx"`` (l.62); here it is an environment/module setting.
DATA_PATH = os.environ.get("MEMVUL_DATA_PATH", "")


@DatasetReader.register("reader_memory")
class ReaderMemory(DatasetReader):
    def __init__(self,
                 tokenizer: Tokenizer = None,
                 same_diff_ratio: Dict[str, int] = None,
                 target: str = "Security_Issue_Full",
                 anchor_path: str = "CWE_anchor_golden_project.json",
                 sample_neg: float = None,
                 train_iter: int = None,
                 token_indexers: Dict[str, TokenIndexer] = None) -> None:
        super().__init__()
        self._token_indexers = token_indexers
        self._tokenizer = tokenizer
        self._same_diff_ratio = same_diff_ratio or {"diff": 6, "same": 2}
        self._choice_neg = [True, False]
        select_neg = sample_neg or 0.1
        self._train_iter = train_iter or 1
        self._select_neg = [select_neg, 1 - select_neg]
        self._target = target
        self._dataset = dict()
        self._cve_info = None
        self._anchor = None
        if sample_neg is None:
            return
        cve_path = os.path.join(DATA_PATH, "CVE_dict.json") if DATA_PATH else "CVE_dict.json"
        if os.path.exists(cve_path):
            with open(cve_path, "r") as f:
                self._cve_info = json.load(f)
        if os.path.exists(anchor_path):
            with open(anchor_path, "r") as f:
                self._anchor = json.load(f)
            for k, v in self._anchor.items():
                self._anchor[k] = self._tokenizer.tokenize(v)

    STREAM_FIRST = 1024  # samples of the first batched call of a stream (two batches of 512: the pipeline fills while the next, full-size chunk is tokenised)
    STREAM_CHUNK = 4096  # samples tokenised per batched call when the Instances are streamed (_read)

    def read_dataset(self, file_path, defer_tokens: bool = False):
        """``defer_tokens`` (the streaming form of _read): return the grouped samples WITHOUT their "description" tokens; _read tokenises them chunk by chunk."""
        if "golden" in file_path:
            dataset = dict()
            with open(file_path, "r", encoding="utf-8") as f:
                anchors = json.load(f)
            for cwe_id, description in anchors.items():
                dataset[cwe_id] = [{self._target: cwe_id, "description": self._tokenizer.tokenize(description)}]
            return dataset

        if self._dataset.get(file_path):
            return self._dataset[file_path]

        dataset = self._grouped_samples(file_path)
        flat = [s for group in dataset.values() for s in group]
        rows_of = getattr(self._tokenizer, "batch_token_rows", None)
        if defer_tokens and rows_of is not None and not HAVE_ALLENNLP:
            self._dataset[file_path] = dataset
            return dataset
        if rows_of is not None and not HAVE_ALLENNLP:
            # ONE batched call into the tokenizer backend for the whole file, tokens as id arrays (tokenizer.TokenRow: Token objects only when somebody
            # asks for them) — text by text with a list of Token objects per text this stage ran at 1.2 k issue reports/s (profiles/r06_*_e2e_dropin.txt)
            for s, row in zip(flat, rows_of([self._text_of(s) for s in flat])):
                s["description"] = row
        else:  # AllenNLP's own tokenizer / TextField (registry.HAVE_ALLENNLP): real Token lists
            for s in flat:
                s["description"] = self._tokenizer.tokenize(self._text_of(s))
        self._dataset[file_path] = dataset
        return dataset

    @staticmethod
    def _text_of(s) -> str:
        return f"{s['Issue_Title']}. {s['Issue_Body']}"

    def _grouped_samples(self, file_path):
        """reader_memory.py:82-111 without the tokenisation: samples grouped under "neg" / their CWE id, in file order."""
        with open(file_path, "r", encoding="utf-8") as f:
            samples = json.load(f)
        dataset = {"neg": list()}
        for s in samples:
            label = "pos" if str(s[self._target]) == "1" else "neg"
            s[self._target] = label
            if label == "pos":
                if "CWE_ID" not in s or self._cve_info is not None:
                    if self._cve_info is None:
                        raise FileNotFoundError("CVE_dict.json is needed to map a positive issue report to its CWE id "
                                                "(reader_memory.py:62-64); set MEMVUL_DATA_PATH")
                    s["CWE_ID"] = self._cve_info[s["CVE_ID"]]["CWE_ID"]
                label = s["CWE_ID"]
                if label is None:
                    continue  # "2 dirty data" (l.103-105)
                if label not in dataset:
                    dataset[label] = list()
            dataset[label].append(s)
        return dataset

    def read_arrays(self, file_path, workers: int = 0, shard=None):
        """The ``test_`` / ``validation_`` branch of ``_read`` as arrays instead of Instances (same samples, same order:
        positives first, reader_memory.py:150-152): ``ids int32 [N, L]`` zero-padded, ``lens int32 [N]``,
        ``same bool [N]`` (label "same" = a positive), ``labels`` (CWE id or "neg", what the records carry) and
        ``urls``.  Feeds ``ModelMemory.sweep_arrays``; tokenisation is one batched call (``Tokenizer.batch_ids``).
        ``shard=(rank, world)``: only this rank's contiguous slice of that order (distributed.shard_range) is tokenised
        and returned; ``n_total`` / ``first`` place it in the whole set."""
        if "test_" in file_path:
            type_ = "unlabel"
        elif "validation_" in file_path and "golden" not in file_path:
            type_ = "test"
        else:
            raise NotImplementedError("read_arrays serves the 'test_' / 'validation_' branches (reader_memory.py:146-157)")
        dataset = self._grouped_samples(file_path)
        all_data = [s for group in dataset.values() for s in group]
        all_data.reverse()
        n_total, first = len(all_data), 0
        if shard is not None:
            from .distributed import shard_range

            first, count = shard_range(n_total, int(shard[0]), int(shard[1]))
            all_data = all_data[first:first + count]
        texts = [self._text_of(s) for s in all_data]
        if hasattr(self._tokenizer, "batch_ids"):
            ids, lens = self._tokenizer.batch_ids(texts, workers=workers)
        else:  # AllenNLP's own PretrainedTransformerTokenizer (registry.HAVE_ALLENNLP): text by text, same ids
            rows = [[t.text_id for t in self._tokenizer.tokenize(x)] for x in texts]
            lens = np.fromiter((len(r) for r in rows), dtype=np.int32, count=len(rows))
            ids = np.zeros((len(rows), int(lens.max()) if len(rows) else 0), np.int32)
            for i, r in enumerate(rows):
                ids[i, :len(r)] = r
        same = np.fromiter((s[self._target] == "pos" for s in all_data), dtype=bool, count=len(all_data))
        labels = [s["CWE_ID"] if s[self._target] == "pos" else s[self._target] for s in all_data]
        return {"type": type_, "ids": ids, "lens": lens, "same": same, "labels": labels,
                "urls": [s["Issue_Url"] for s in all_data], "n_total": n_total, "first": first}

    def iter_arrays(self, file_path, chunk: int = 16384):
        """``read_arrays`` as a stream of array dicts of ``chunk`` consecutive samples each (same order, same keys; ``first`` = the chunk's offset in the
        whole set), the NEXT chunk's batched tokenisation running on a helper thread while the caller scores this one (predict_memory.evaluate_arrays):
        the whole-job rate of the array form is then bound by the slower of the two stages, not by their sum."""
        from concurrent.futures import ThreadPoolExecutor

        if "test_" in file_path:
            type_ = "unlabel"
        elif "validation_" in file_path and "golden" not in file_path:
            type_ = "test"
        else:
            raise NotImplementedError("iter_arrays serves the 'test_' / 'validation_' branches (reader_memory.py:146-157)")
        dataset = self._grouped_samples(file_path)
        all_data = [s for group in dataset.values() for s in group]
        all_data.reverse()
        n_total = len(all_data)

        # chunk boundaries: the first chunks are small and double (chunk / 8, / 4, / 2, then chunk) — the scorer waits for the FIRST chunk's tokenisation only,
        # so a 16 k-sample first chunk cost the array form a second of idle GPU per file; every chunk stays a multiple of chunk / 8 (= of the batch size when
        # chunk is 32 batches: evaluate_arrays writes one JSON line per batch_size samples)
        bounds, step = [0], (chunk // 8 if chunk % 8 == 0 and chunk >= 8 else chunk)
        while bounds[-1] < n_total:
            bounds.append(min(n_total, bounds[-1] + step))
            step = min(chunk, step * 2)

        def make(i):
            first = bounds[i]
            part = all_data[first:bounds[i + 1]]
            ids, lens = self._tokenizer.batch_ids([self._text_of(s) for s in part])
            return {"type": type_, "ids": ids, "lens": lens, "same": np.fromiter((s[self._target] == "pos" for s in part), dtype=bool, count=len(part)),
                    "labels": [s["CWE_ID"] if s[self._target] == "pos" else s[self._target] for s in part],
                    "urls": [s["Issue_Url"] for s in part], "n_total": n_total, "first": first}

        with ThreadPoolExecutor(1) as ex:
            fut = ex.submit(make, 0) if n_total else None
            for i in range(len(bounds) - 1):
                cur = fut.result()
                fut = ex.submit(make, i + 1) if i + 2 < len(bounds) else None
                yield cur

    def _stream(self, samples, type_):
        """Instances of ``samples`` in order, tokenised STREAM_CHUNK texts per batched call with the NEXT chunk's call running on a helper thread (the
        WordPiece backend releases the GIL) while this chunk's Instances are consumed — so that a consumer that scores batches as they arrive
        (predict_memory.evaluate) overlaps the tokenisation of a 40 k-report file (2.4 s) with the GPU instead of waiting for it."""
        from concurrent.futures import ThreadPoolExecutor

        rows_of = self._tokenizer.batch_token_rows
        first = min(len(samples), self.STREAM_FIRST)  # a small first chunk: the consumer's first batch (and the GPU) starts a quarter of a second earlier
        chunks = [samples[:first]] + [samples[i:i + self.STREAM_CHUNK] for i in range(first, len(samples), self.STREAM_CHUNK)]
        chunks = [c for c in chunks if len(c)]

        def tok(chunk):
            todo = [s for s in chunk if "description" not in s]
            for s, row in zip(todo, rows_of([self._text_of(s) for s in todo])):
                s["description"] = row
            return chunk

        with ThreadPoolExecutor(1) as ex:
            fut = ex.submit(tok, chunks[0]) if chunks else None
            for k in range(len(chunks)):
                chunk = fut.result()
                fut = ex.submit(tok, chunks[k + 1]) if k + 1 < len(chunks) else None
                for sample in chunk:
                    yield self.text_to_instance((sample, sample), type_=type_)

    def _read(self, file_path):
        streaming = (("test_" in file_path or "validation_" in file_path) and "golden" not in file_path
                     and hasattr(self._tokenizer, "batch_token_rows") and not HAVE_ALLENNLP)
        dataset = self.read_dataset(file_path, defer_tokens=streaming)
        all_data = list()
        for ll in list(dataset.values()):
            all_data.extend(ll)
        dist = {"pos": sum(len(v) for k, v in dataset.items() if k != "neg"), "neg": len(dataset["neg"]) if "neg" in dataset else None}
        logger.info(dist)

        if "golden_" in file_path:
            logger.info("Begin loading golden instances------")
            for sample in all_data:
                yield self.text_to_instance((sample, sample), type_="golden")
            logger.info(f"Num of golden instances is {len(all_data)}")
        elif "test_" in file_path:
            logger.info("Begin predict------")
            if streaming:
                yield from self._stream(all_data[::-1], "unlabel")
            else:
                for sample in reversed(all_data):  # positives first, then the negatives
                    yield self.text_to_instance((sample, sample), type_="unlabel")
            logger.info(f"Predict sample num is {len(all_data)}")
        elif "validation_" in file_path:
            logger.info("Begin testing------")
            if streaming:
                yield from self._stream(all_data[::-1], "test")
            else:
                for sample in reversed(all_data):
                    yield self.text_to_instance((sample, sample), type_="test")
            logger.info(f"Test sample num is {len(all_data)}")
        else:
            raise NotImplementedError(
                "the online pair-sampling training branch (reader_memory.py:164-192) is outside the inference hot "
                "path; file names select the branch by substring: 'golden_', 'test_', 'validation_'")

    def text_to_instance(self, p, type_="train") -> Instance:
        fields = dict()
        ins1, ins2 = p
        fields["sample1"] = TextField(ins1["description"], self._token_indexers)
        if type_ == "train":
            raise NotImplementedError("training pairs (reader_memory.py:205-229) are outside the inference hot path")
        if type_ in ["test", "unlabel"]:
            # pos == same (only CIRs make matched pairs), neg == diff
            fields["label"] = LabelField("same" if ins1[self._target] == "pos" else "diff")
        meta_ins1 = {"label": ins1[self._target]}
        if type_ in ["test", "unlabel"]:
            if ins1[self._target] == "pos":
                meta_ins1["label"] = ins1["CWE_ID"]
            meta_ins1["Issue_Url"] = ins1["Issue_Url"]
        fields["metadata"] = MetadataField({"type": type_, "instance": [meta_ins1]})
        return Instance(fields)
