"""``model_memory`` — the Model of the hot path (reference: MemVul/model_memory.py:39-224), with the
tensor work done by libmemvul_hip.so through the C ABI.

Same constructor arguments, method names and output-dict keys as the reference:
  * ``__init__(vocab, text_field_embedder, PTM, dropout, label_namespace, device, use_header, temperature,
    initializer, regularizer)`` (l.41-51);
  * ``forward(sample1, sample2, label, metadata)`` (l.118-167): ``metadata[0]["type"] == "golden"`` ->
    ``forward_gold_instances`` and ``{}`` (l.126-128); ``"test"``/``"unlabel"`` -> ``{"meta", "probs"}``
    with ``probs`` the full ``[B,G,2]`` softmax (an ndarray instead of ``p.tolist()``, l.143), metric
    updates on the best-anchor row (l.144-147,162-166);
  * ``forward_gold_instances`` (l.105-115), ``forward_on_instances`` (AllenNLP ``Model`` API used at
    predict_memory.py:81-83), ``make_output_human_readable`` (l.169-191), ``get_metrics`` (l.194-217),
    ``get_output_dim`` (l.220-224);
  * ``_golden_instances_embeddings`` / ``_golden_instances_labels`` stay assignable from outside
    (callbacks.py:48-49 resets them to ``None`` before rebuilding the bank).
Out of scope: the training branch (l.149-160: pair loss with temperature) — it raises.

There is no CPU implementation here: without the HIP library / a GPU, constructing the engine raises.
"""
from __future__ import annotations

import json
import logging
import os
from typing import Any, Dict, List, Optional

import numpy as np

from .binding import Engine, compute_dtype_of
from .custom_metric import SiameseMeasureV1
from .data import Instance, collate
from .registry import HAVE_ALLENNLP, Model, TextFieldEmbedder, TokenEmbedder, Vocabulary, register_builtin

logger = logging.getLogger(__name__)

PFX_BERT = "_text_field_embedder.token_embedder_tokens.transformer_model."


class _LoadResult(tuple):
    """``(missing_keys, unexpected_keys)`` of ``torch.nn.Module.load_state_dict``."""

    def __new__(cls, missing, unexpected):
        return super().__new__(cls, (list(missing), list(unexpected)))

    missing_keys = property(lambda self: self[0])
    unexpected_keys = property(lambda self: self[1])


def _np(x):
    if x is None:
        return None
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    return np.asarray(x)


@TokenEmbedder.register("custom_pretrained_transformer")
class CustomPretrainedTransformerEmbedder(TokenEmbedder):
    """Configuration holder for the reference's embedder (custom_PTM_embedder.py:22-147): the BERT
    weights arrive with the archive's ``weights.th`` (keys ``..transformer_model.*``); the directory
    ``pretrained_model_path`` (l.99) is only consulted for ``config.json`` when present."""

    def __init__(self, model_name: str = None, *, max_length: int = None, sub_module: str = None, train_parameters: bool = True,
                 eval_mode: bool = False, last_layer_only: bool = True, override_weights_file: Optional[str] = None,
                 override_weights_strip_prefix: Optional[str] = None, gradient_checkpointing: Optional[bool] = None,
                 tokenizer_kwargs: Optional[Dict[str, Any]] = None, transformer_kwargs: Optional[Dict[str, Any]] = None,
                 pretrained_model_path: str = "out_wwm/") -> None:
        super().__init__()  # AllenNLP's TokenEmbedder is a torch.nn.Module (BasicTextFieldEmbedder registers it as a sub-module)
        if max_length is not None:
            raise NotImplementedError("segment folding (custom_PTM_embedder.py:244-381) is dead code in every reference config")
        if not last_layer_only:
            raise NotImplementedError("ScalarMix over hidden states is not used by the reference configs")
        self.model_name, self.pretrained_model_path = model_name, pretrained_model_path
        self.output_dim = 768

    def get_output_dim(self):
        return self.output_dim


@register_builtin(TokenEmbedder, "pretrained_transformer")
class _PlainPretrainedTransformerEmbedder(CustomPretrainedTransformerEmbedder):
    """config_no_pretrain.json uses the stock embedder type; same inference path."""

    def __init__(self, model_name: str = None, **kw) -> None:
        kw.pop("pretrained_model_path", None)
        super().__init__(model_name, **kw)


@register_builtin(TextFieldEmbedder, "basic")
class BasicTextFieldEmbedder(TextFieldEmbedder):
    def __init__(self, token_embedders: Dict[str, TokenEmbedder]) -> None:
        super().__init__()
        self.token_embedders = token_embedders

    def get_output_dim(self):
        return sum(e.get_output_dim() for e in self.token_embedders.values())


class _ClassificationCounts:
    """CategoricalAccuracy + FBetaMeasure(beta=1, average in {"weighted", None}) of model_memory.py:80-84,
    from one running confusion matrix."""

    def __init__(self, num_class: int) -> None:
        self.n = num_class
        self.reset()

    def reset(self):
        self.cm = np.zeros((self.n, self.n), np.int64)  # [gold, pred]

    def __call__(self, predictions: np.ndarray, gold_labels: np.ndarray):
        pred = np.argmax(predictions, axis=-1)
        np.add.at(self.cm, (gold_labels.astype(np.int64), pred), 1)

    def accuracy(self) -> float:
        t = self.cm.sum()
        return float(np.trace(self.cm) / t) if t else 0.0

    def prf(self):
        tp = np.diag(self.cm).astype(np.float64)
        pred_sum, true_sum = self.cm.sum(0).astype(np.float64), self.cm.sum(1).astype(np.float64)
        with np.errstate(divide="ignore", invalid="ignore"):
            p = np.where(pred_sum > 0, tp / pred_sum, 0.0)
            r = np.where(true_sum > 0, tp / true_sum, 0.0)
            f = np.where(p + r > 0, 2 * p * r / (p + r), 0.0)
        return p, r, f, true_sum

    def weighted(self):
        p, r, f, w = self.prf()
        tot = w.sum()
        if tot == 0:
            return 0.0, 0.0, 0.0
        return float((p * w).sum() / tot), float((r * w).sum() / tot), float((f * w).sum() / tot)


@Model.register("model_memory")
class ModelMemory(Model):
    def __init__(self,
                 vocab: Vocabulary,
                 text_field_embedder: TextFieldEmbedder,
                 PTM: str = "bert-base-uncased",
                 dropout: float = 0.1,
                 label_namespace: str = "labels",
                 device: str = "cpu",
                 use_header: bool = True,
                 temperature: float = 1,
                 initializer: Any = None,
                 regularizer: Any = None,
                 engine_options: Optional[Dict[str, Any]] = None) -> None:
        super().__init__(vocab, regularizer)
        self.device = device
        self._device_index = int(str(device).split(":")[1]) if ":" in str(device) else 0
        self._use_header = use_header
        self._label_namespace = label_namespace
        self._idx2token_label = self.vocab.get_index_to_token_vocabulary(namespace=label_namespace)
        self._same_idx = vocab.get_token_index("same", namespace=label_namespace)
        self._text_field_embedder = text_field_embedder
        self._num_class = self.vocab.get_vocab_size(self._label_namespace)
        self._temperature = temperature
        self._golden_labels: Optional[List[str]] = None
        self._counts = _ClassificationCounts(self._num_class)
        self._siamese_metric = SiameseMeasureV1(self._same_idx)
        self._engine_options = dict(engine_options or {})
        self._engine: Optional[Engine] = None

    # ---- weights --------------------------------------------------------------------------------
    def load_state_dict(self, state_dict: Dict[str, Any], strict: bool = True):
        """``model.load_state_dict(torch.load(weights.th))`` of AllenNLP's ``load_archive``: creates the
        engine on ``self.device`` sized from the tensors and uploads them (mv_load_tensor /
        mv_finalize_weights)."""
        sd = {k: _np(v) for k, v in state_dict.items()}
        layers = 0
        while (PFX_BERT + f"encoder.layer.{layers}.attention.self.query.weight") in sd:
            layers += 1
        vocab_size = sd[PFX_BERT + "embeddings.word_embeddings.weight"].shape[0]
        max_pos = min(512, sd[PFX_BERT + "embeddings.position_embeddings.weight"].shape[0])
        type_vocab = sd[PFX_BERT + "embeddings.token_type_embeddings.weight"].shape[0]
        opts = dict(max_tokens=128 * 512, max_batch=512, max_anchors=1024)
        opts.update(self._engine_options)
        # use_header (model_memory.py:69-73): with the header the matcher runs on its 512-d output; without it the model has
        # no _projector_single and `_projector` is Linear(3 * 768, 2) on the pooler output -> mv_config.proj_dim = 768
        has_header = "_projector_single._linear_layers.0.weight" in sd
        if has_header != bool(self._use_header):
            raise ValueError(f"use_header={self._use_header} but the state dict {'has' if has_header else 'lacks'} "
                             "_projector_single (model_memory.py:69-71)")
        opts["proj_dim"] = 512 if self._use_header else 768
        # compute dtype: MV_F16X8 ("precise", the DEFAULT: fp16 sweep + one fp8 correction sweep per GEMM — the mode that holds the
        # reference's 1e-3 on trained-like logits; include/memvul_hip.h) or MV_F16 ("fast": explicit opt-in, 3.0-5.6e-3 there);
        # engine_options["compute_dtype"] or $MEMVUL_COMPUTE = precise | f16x8 | f16 | fast.  An unknown name raises here
        # (ValueError), not inside ctypes.
        cd = compute_dtype_of(opts.pop("compute_dtype", None))
        if self._engine is not None:
            self._engine.close()
        self._engine = Engine(self._device_index, vocab_size=vocab_size, layers=layers, max_pos=max_pos, type_vocab=type_vocab,
                              same_idx=self._same_idx, **opts)
        self._engine.load_state_dict(sd, cd)
        # torch's contract (AllenNLP's Model._load unpacks it): every tensor the engine needs was found by mv_load_tensor /
        # mv_finalize_weights (they fail otherwise); keys it has no use for — the BertModel's own pooler copy, position_ids — are
        # not "unexpected" to a model that declares no parameters of its own
        return _LoadResult([], [])

    @property
    def engine(self) -> Engine:
        if self._engine is None:
            raise RuntimeError("ModelMemory has no weights yet: call load_state_dict (or load_archive) first")
        return self._engine

    # ---- anchor memory attributes the callers poke (callbacks.py:48-49) --------------------------
    @property
    def _golden_instances_embeddings(self):
        if self._engine is None or self._engine.n_anchors == 0:
            return None
        return self._engine.anchor_get()

    @_golden_instances_embeddings.setter
    def _golden_instances_embeddings(self, value):
        if value is None:
            if self._engine is not None:
                self._engine.anchor_reset()
        else:
            self.engine.anchor_set(_np(value))

    @property
    def _golden_instances_labels(self):
        return self._golden_labels

    @_golden_instances_labels.setter
    def _golden_instances_labels(self, value):
        self._golden_labels = None if value is None else list(value)

    # ---- the reference's methods ------------------------------------------------------------------
    @staticmethod
    def _ids_lens(sample: Dict[str, Dict[str, Any]]):
        t = sample["tokens"]
        known = t.get("_collated")
        if known is not None:  # memvul_amd.data.collate: lengths, prefix mask, zero padding and the segment check by construction
            ids32, lens, single = known
            if not single:
                raise ValueError("non-zero token-type ids: the hot path is single-segment (custom_PTM_embedder.py:199-202)")
            return ids32, lens
        ids = _np(t["token_ids"]).astype(np.int32)
        mask = np.asarray(_np(t["mask"]), bool)
        tid = t.get("type_ids")
        if tid is not None and _np(tid).any():
            raise ValueError("non-zero token-type ids: the hot path is single-segment (custom_PTM_embedder.py:199-202)")
        lens = mask.sum(1).astype(np.int32)
        # a prefix mask: the LAST set position of a row is its count - 1 (then every position before it is set too)
        last = mask.shape[1] - 1 - np.argmax(mask[:, ::-1], axis=1) if mask.shape[1] else np.zeros(len(lens), np.int64)
        if not np.array_equal(np.where(lens > 0, last, -1), lens.astype(np.int64) - 1):
            raise ValueError("mask must be a prefix mask (pad-to-longest collation)")
        np.multiply(ids, mask, out=ids)
        return ids, lens

    def _instance_forward(self, sample, use_header: bool = False) -> np.ndarray:
        # (the reference passes self._use_header at every call site, l.112, 133; the engine was created for that choice)
        assert bool(use_header) == bool(self._use_header), "the engine was built for use_header=%s" % self._use_header
        ids, lens = self._ids_lens(sample)
        return self.engine.encode(ids, lens)

    def forward_gold_instances(self, sample, metadata):
        ids, lens = self._ids_lens(sample)
        self.engine.anchor_append(ids, lens)
        labels = [_["instance"][0]["label"] for _ in metadata]
        if self._golden_labels is None:
            self._golden_labels = labels
        else:
            self._golden_labels.extend(labels)

    def forward(self, sample1=None, sample2=None, label=None, metadata: List[Dict[str, Any]] = None) -> Dict[str, Any]:
        output_dict: Dict[str, Any] = dict()
        if metadata and metadata[0]["type"] == "golden":
            self.forward_gold_instances(sample1, metadata)
            return output_dict
        if metadata:
            output_dict["meta"] = metadata
        if not (metadata and metadata[0]["type"] in ["test", "unlabel"]):
            raise NotImplementedError("the pair-training branch (model_memory.py:149-160) is outside the inference hot path")
        ids, lens = self._ids_lens(sample1)
        # (rows grouped by their own padded length: a pad-to-longest batch of unsorted reports is mostly padding — Engine.forward_by_length)
        out = self.engine.forward_by_length(ids, lens, want_logits=False, want_probs=True)
        output_dict["probs"] = out["probs"]           # [B,G,2]; the reference stores p.tolist()
        output_dict["best_anchor"] = out["best_idx"]  # extra (not in the reference): g* per issue report
        probs = out["best"]                           # [B,2] = p[b, g*]
        if label is not None:
            self._counts(probs, _np(label))
        self._siamese_metric(probs, metadata)
        return output_dict

    __call__ = forward

    def forward_begin(self, sample1=None, sample2=None, label=None, metadata: List[Dict[str, Any]] = None):
        """``forward`` in two halves for a caller that overlaps batches (predict_memory.evaluate): the engine takes the batch here without waiting for it
        (Engine.forward_by_length_begin) ..."""
        if not (metadata and metadata[0]["type"] in ["test", "unlabel"]) or not hasattr(self.engine, "forward_by_length_begin"):
            return ("done", self.forward(sample1, sample2, label, metadata))
        ids, lens = self._ids_lens(sample1)
        return ("pending", self.engine.forward_by_length_begin(ids, lens, want_logits=False, want_probs=True), label, metadata)

    def forward_end(self, pending) -> Dict[str, Any]:
        """... and is collected here, where the metric accumulators are updated: ``forward_end(forward_begin(**batch))`` is ``forward(**batch)`` (model_memory.py:118-167),
        and batches collected in the order they were begun update the metrics in the reference's order."""
        if pending[0] == "done":
            return pending[1]
        _, ticket, label, metadata = pending
        out = self.engine.forward_by_length_end(ticket)
        output_dict: Dict[str, Any] = {"meta": metadata, "probs": out["probs"], "best_anchor": out["best_idx"]}
        probs = out["best"]
        if label is not None:
            self._counts(probs, _np(label))
        self._siamese_metric(probs, metadata)
        return output_dict

    def forward_on_instances(self, instances: List[Instance]) -> List[Dict[str, Any]]:
        """AllenNLP ``Model.forward_on_instances``: collate (pad to the longest of the chunk) + forward."""
        batch = collate(instances, self.vocab)
        out = self.make_output_human_readable(self.forward(**batch))
        return out if isinstance(out, list) else [dict() for _ in instances]

    def sweep(self, instances: List[Instance], batch_size: int = 512) -> List[List[Dict[str, Any]]]:
        """The whole evaluation set in one resident, length-bucketed sweep (Engine.bucketed_sweep) instead of one
        ``forward`` per pad-to-longest batch (predict_memory.py:97-110): same per-IR arithmetic, same metric updates,
        same human-readable records in the same order and the same per-batch grouping — but every batch runs at its
        own longest member's length, two batches are in flight on the GPU, and token ids / results cross PCIe once."""
        if not instances:
            return []
        metadata, p_same = self.sweep_scores(instances, batch_size)
        out = []
        for s0 in range(0, len(instances), batch_size):
            out.append(self.make_output_human_readable({"meta": metadata[s0:s0 + batch_size], "p_same": p_same[s0:s0 + batch_size]}))
        return out

    def sweep_scores(self, instances: List[Instance], batch_size: int = 512):
        """``sweep`` up to the records: ``(metadata, P(same) [n, G])`` with the metric accumulators updated — what a caller that writes the records itself
        (predict_memory.evaluate_sweep, through records.RecordWriter: the same bytes without one dict per issue report) needs."""
        batch = collate(instances, self.vocab)
        metadata = batch["metadata"]
        if metadata[0]["type"] not in ["test", "unlabel"]:
            raise NotImplementedError("sweep() serves the test / unlabel branch (model_memory.py:133-147)")
        ids, lens = self._ids_lens(batch["sample1"])
        best, best_idx, p_same = self.engine.bucketed_sweep(ids, lens, batch_size, with_probs=True)
        if batch.get("label") is not None:
            self._counts(best, _np(batch["label"]))
        self._siamese_metric(best, metadata)
        return metadata, p_same

    def sweep_arrays(self, arrays: Dict[str, Any], first: int = 0, last: Optional[int] = None, batch_size: int = 512,
                     with_probs: bool = False):
        """``sweep`` on the array form of the evaluation set (ReaderMemory.read_arrays): rows ``first:last`` in one resident
        length-bucketed sweep, metric accumulators updated exactly as ``forward`` / ``sweep`` do (model_memory.py:133-147,
        162-167).  Returns ``(best [n, 2], best_idx [n], p_same [n, G] or None)``; no Instances, no per-IR Python objects."""
        if arrays["type"] not in ["test", "unlabel"]:
            raise NotImplementedError("sweep_arrays() serves the test / unlabel branch (model_memory.py:133-147)")
        last = len(arrays["lens"]) if last is None else last
        lens = np.ascontiguousarray(arrays["lens"][first:last], np.int32)
        if len(lens) == 0:
            return np.zeros((0, 2), np.float32), np.zeros((0,), np.int32), None
        ids = arrays["ids"][first:last, :int(lens.max())]
        best, best_idx, p_same = self.engine.bucketed_sweep(ids, lens, batch_size, with_probs=with_probs)
        same = np.asarray(arrays["same"][first:last], bool)
        diff_idx = self.vocab.get_token_index("diff", namespace=self._label_namespace)
        self._counts(best, np.where(same, self._same_idx, diff_idx).astype(np.int64))
        self._siamese_metric.add_arrays(same.astype(np.uint8), best[:, self._same_idx])
        return best, best_idx, p_same

    def format_records_json(self, labels: List[str], urls: List[str], p_same: np.ndarray) -> str:
        """``json.dumps(make_output_human_readable(...))`` of one batch, byte for byte, built from arrays: the line
        predict_memory.py:111 writes per batch ({"Issue_Url", "label", "predict": {cwe: P(same)}} per issue report)."""
        from .records import format_batch, record_layout

        cols, fmt, names = record_layout(self._golden_labels)
        return format_batch(fmt, names, urls, labels, np.asarray(p_same)[:, cols].astype(np.float64))

    def make_output_human_readable(self, output_dict: Dict[str, Any]):
        if "meta" not in output_dict or output_dict["meta"][0]["type"] not in ["test", "unlabel"]:
            return output_dict
        labels = self._golden_labels
        ps = (np.asarray(output_dict["p_same"]) if "p_same" in output_dict                  # [B,G] (resident sweep)
              else np.asarray(output_dict["probs"])[:, :, self._same_idx])
        # vote_num[golden_name] = p[idx_same] in anchor order: a later duplicate label overwrites (l.181-183)
        order = {name: i for i, name in enumerate(labels)}  # last occurrence wins
        names = list(order.keys())
        cols = np.fromiter(order.values(), dtype=np.int64, count=len(order))
        sel = ps[:, cols].astype(np.float64)
        out2file = []
        for i, meta in enumerate(output_dict["meta"]):
            out2file.append({"Issue_Url": meta["instance"][0]["Issue_Url"], "label": meta["instance"][0]["label"],
                             "predict": dict(zip(names, sel[i].tolist()))})
        return out2file

    def get_metrics(self, reset: bool = False) -> Dict[str, float]:
        metrics = dict()
        metrics["accuracy"] = self._counts.accuracy()
        metrics["precision"], metrics["recall"], metrics["f1-score"] = self._counts.weighted()
        p, r, f, _ = self._counts.prf()
        for i in range(self._num_class):
            metrics[f"{self._idx2token_label[i]}_precision"] = float(p[i])
            metrics[f"{self._idx2token_label[i]}_recall"] = float(r[i])
            metrics[f"{self._idx2token_label[i]}_f1-score"] = float(f[i])
        if reset:
            s = self._siamese_metric.get_metric(reset)
            metrics["s_precision"], metrics["s_recall"], metrics["s_f1-score"] = s["precision"], s["recall"], s["f1"]
            metrics["s_thres"], metrics["s_auc"], metrics["s_ave_precision_score"] = s["thres"], s["auc"], s["ave_precision_score"]
            self._counts.reset()
        return metrics

    def get_output_dim(self, use_header=False):
        return 512 if use_header else self._text_field_embedder.get_output_dim()
