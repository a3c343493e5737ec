"""``custom_validation`` — the per-epoch anchor refresh of the reference's trainer (MemVul/callbacks.py:25-53): the
other caller of the golden-anchor path.  After every epoch the bank is dropped and rebuilt from the anchor file in
chunks of 128 (l.48-53) through exactly the attributes / methods the hot path exposes
(``_golden_instances_embeddings = None`` -> ``mv_anchor_reset``; ``forward_on_instances`` -> ``mv_anchor_append``), so a
(non-accelerated) training loop can validate against the fast path.  The trainer itself is out of scope."""
from __future__ import annotations

import logging
from typing import Any, Dict, Optional

from .reader_memory import ReaderMemory
from .registry import Registrable
from .tokenizer import PretrainedTransformerIndexer, PretrainedTransformerTokenizer

logger = logging.getLogger(__name__)


class TrainerCallback(Registrable):
    """Stand-in for ``allennlp.training.TrainerCallback`` (only ``on_epoch`` is used here)."""

    def __init__(self, serialization_dir: Optional[str] = None) -> None:
        self.serialization_dir = serialization_dir


@TrainerCallback.register("custom_validation")
class CustomValidation(TrainerCallback):
    def __init__(self, anchor_path: str = "CWE_anchor_golden_project.json", data_reader=None, data_loader=None,
                 serialization_dir: Optional[str] = None) -> None:  # the reference's order (callbacks.py:27-31); data_loader unused there too
        super().__init__(serialization_dir)
        PTM = "bert-base-uncased"
        reader = data_reader or ReaderMemory(tokenizer=PretrainedTransformerTokenizer(PTM, add_special_tokens=True, max_length=512),
                                             token_indexers={"tokens": PretrainedTransformerIndexer(PTM, namespace="tags")})
        self._anchors = list(reader.read(anchor_path))

    def on_epoch(self, trainer, metrics: Dict[str, Any] = None, epoch: int = 0, is_primary: bool = True, **kwargs) -> None:
        model = trainer.model
        model.eval()
        model._golden_instances_embeddings = None  # reset
        model._golden_instances_labels = None  # reset
        logger.info("updating golden embeddings")
        model.forward_on_instances(self._anchors[:128])
        if len(self._anchors) > 128:
            model.forward_on_instances(self._anchors[128:])
