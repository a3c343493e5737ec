from dataclasses import dataclass
from typing import Optional


@dataclass
class Token:
    """allennlp/data/tokenizers/token_class.py (the fields the transformer tokenizer fills)."""

    text: Optional[str] = None
    idx: Optional[int] = None
    idx_end: Optional[int] = None
    lemma_: Optional[str] = None
    pos_: Optional[str] = None
    tag_: Optional[str] = None
    dep_: Optional[str] = None
    ent_type_: Optional[str] = None
    text_id: Optional[int] = None
    type_id: Optional[int] = None

    def __str__(self):
        return self.text
