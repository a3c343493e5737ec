"""``model_single`` — MemVul-m, the memory-less variant (reference: MemVul/model_single.py:36-128): the same BERT
issue encoder, BertPooler and 512-d ReLU header as model_memory, followed by a bias-free ``Linear(512, num_class)``
and a softmax instead of the anchor matcher.  The encoder / pooler / header run in libmemvul_hip.so (``mv_encode``,
the K1-K8 kernels of the hot path); the 512 x num_class classifier and the softmax are two numpy lines on the
``[B,512]`` embedding that comes back.

Same constructor arguments (l.38-46), ``forward(sample, label, metadata)`` -> ``{"meta", "probs", "loss"}`` (l.77-100),
``make_output_human_readable`` records ``{"Issue_Url","label","predict","prob"}`` (l.102-112) and ``get_metrics``
keys (l.114-128) as the reference.  State-dict keys (AllenNLP archive of the reference model):
``_projector.0._linear_layers.0.{weight,bias}`` (header), ``_projector.1.weight`` (classifier),
``_bert_pooler.pooler.dense.*`` and the ``_text_field_embedder...transformer_model.*`` BERT tensors.
"""
from __future__ import annotations

import logging
from typing import Any, Dict, List, Optional

import numpy as np

from .binding import Engine, compute_dtype_of
from .model_memory import PFX_BERT, _ClassificationCounts, _np
from .registry import Model, TextFieldEmbedder, Vocabulary

logger = logging.getLogger(__name__)

KEY_HEAD_W, KEY_HEAD_B, KEY_CLS_W = "_projector.0._linear_layers.0.weight", "_projector.0._linear_layers.0.bias", "_projector.1.weight"


@Model.register("model_single")
class ModelSingle(Model):
    def __init__(self,
                 vocab: Vocabulary,
                 text_field_embedder: TextFieldEmbedder,
                 PTM: str = "bert-base-uncased",
                 dropout: float = 0.1,
                 label_namespace: str = "class_labels",
                 device: str = "cpu",
                 initializer: Any = None,
                 regularizer: Any = None,
                 engine_options: Optional[Dict[str, Any]] = None) -> None:
        super().__init__(vocab, regularizer)
        self.device = device
        self._device_index = int(str(device).split(":")[1]) if ":" in str(device) else 0
        self._label_namespace = label_namespace
        self._idx2token_label = vocab.get_index_to_token_vocabulary(namespace=label_namespace)
        self._idx_pos = vocab.get_token_index("pos", namespace=label_namespace)
        self._text_field_embedder = text_field_embedder
        self._num_class = self.vocab.get_vocab_size(self._label_namespace)
        self._counts = _ClassificationCounts(self._num_class)
        self._engine_options = dict(engine_options or {})
        self._engine: Optional[Engine] = None
        self._cls_w: Optional[np.ndarray] = None

    def load_state_dict(self, state_dict: Dict[str, Any], strict: bool = True):
        sd = {k: _np(v) for k, v in state_dict.items()}
        layers = 0
        while (PFX_BERT + f"encoder.layer.{layers}.attention.self.query.weight") in sd:
            layers += 1
        self._cls_w = np.ascontiguousarray(sd[KEY_CLS_W], np.float32)  # [num_class, 512], bias-free (l.66)
        # the engine's encoder takes the header under model_memory's key; its anchor matcher is unused here
        eng_sd = {k: v for k, v in sd.items() if k.startswith(PFX_BERT) or k.startswith("_bert_pooler.")}
        eng_sd["_projector_single._linear_layers.0.weight"] = sd[KEY_HEAD_W]
        eng_sd["_projector_single._linear_layers.0.bias"] = sd[KEY_HEAD_B]
        eng_sd["_projector.weight"] = np.zeros((2, 3 * 512), np.float32)
        opts = dict(max_tokens=128 * 512, max_batch=512, max_anchors=1)
        opts.update(self._engine_options)
        opts_compute = opts.pop("compute_dtype", None)  # None: binding.default_compute() — precise unless $MEMVUL_COMPUTE says otherwise
        if self._engine is not None:
            self._engine.close()
        self._engine = Engine(self._device_index, vocab_size=sd[PFX_BERT + "embeddings.word_embeddings.weight"].shape[0], layers=layers,
                              max_pos=min(512, sd[PFX_BERT + "embeddings.position_embeddings.weight"].shape[0]),
                              type_vocab=sd[PFX_BERT + "embeddings.token_type_embeddings.weight"].shape[0], **opts)
        self._engine.load_state_dict(eng_sd, compute_dtype_of(opts_compute))
        return self

    @property
    def engine(self) -> Engine:
        if self._engine is None:
            raise RuntimeError("ModelSingle has no weights yet: call load_state_dict (or load_archive) first")
        return self._engine

    def classify(self, u: np.ndarray):
        """``softmax(Linear(512, num_class, bias=False)(u))`` (l.66, 92-94) -> (probs [B, num_class], logits)."""
        logits = u.astype(np.float32) @ self._cls_w.T
        e = np.exp(logits - logits.max(axis=-1, keepdims=True))
        return (e / e.sum(axis=-1, keepdims=True)).astype(np.float32), logits

    def forward(self, sample=None, label=None, metadata: List[Dict[str, Any]] = None) -> Dict[str, Any]:
        from .model_memory import ModelMemory

        output_dict: Dict[str, Any] = dict()
        if metadata:
            output_dict["meta"] = metadata
        ids, lens = ModelMemory._ids_lens(sample)
        probs, logits = self.classify(self.engine.encode(ids, lens))
        output_dict["probs"] = probs
        if label is not None:
            lab = _np(label).astype(np.int64)
            z = logits - logits.max(axis=-1, keepdims=True)
            logp = z - np.log(np.exp(z).sum(axis=-1, keepdims=True))
            output_dict["loss"] = float(-logp[np.arange(len(lab)), lab].mean())  # CrossEntropyLoss (l.97)
            self._counts(probs, lab)
        return output_dict

    __call__ = forward

    def make_output_human_readable(self, output_dict: Dict[str, Any]):
        probs = np.asarray(output_dict["probs"])
        idx = np.argmax(probs, axis=1)
        out2file = list()
        for i, k in enumerate(idx):
            meta = output_dict["meta"][i]["instance"]
            out2file.append({"Issue_Url": meta["Issue_Url"], "label": meta["label"], "predict": self._idx2token_label[int(k)],
                             "prob": float(probs[i][self._idx_pos])})
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
            self._counts.reset()
        return metrics
