"""allennlp/data/tokenizers/pretrained_transformer_tokenizer.py (single-sentence path): HF fast tokenizer,
add_special_tokens, truncation to max_length INCLUDING the special tokens; Tokens carry text_id / type_id."""
from typing import Any, Dict, List, Optional

from allennlp.common import cached_transformers

from .token_class import Token
from .tokenizer import Tokenizer


@Tokenizer.register("pretrained_transformer")
class PretrainedTransformerTokenizer(Tokenizer):
    def __init__(self, model_name: str, add_special_tokens: bool = True, max_length: Optional[int] = None,
                 stride: int = 0, tokenizer_kwargs: Optional[Dict[str, Any]] = None) -> None:
        self._model_name = model_name
        self.tokenizer = cached_transformers.get_tokenizer(model_name, **(tokenizer_kwargs or {}))
        self._add_special_tokens = add_special_tokens
        self._max_length = max_length
        self._stride = stride
        self._tokenizer_lowercases = True  # bert-base-uncased
        cls_tok, sep_tok = self.tokenizer.cls_token, self.tokenizer.sep_token
        self.single_sequence_start_tokens = [Token(cls_tok, text_id=self.tokenizer.cls_token_id, type_id=0)]
        self.single_sequence_end_tokens = [Token(sep_tok, text_id=self.tokenizer.sep_token_id, type_id=0)]
        self.sequence_pair_start_tokens = self.single_sequence_start_tokens
        self.sequence_pair_mid_tokens = self.single_sequence_end_tokens
        self.sequence_pair_end_tokens = self.single_sequence_end_tokens

    def tokenize(self, text: str) -> List[Token]:
        max_length = self._max_length
        if max_length is not None and not self._add_special_tokens:
            max_length += len(self.single_sequence_start_tokens) + len(self.single_sequence_end_tokens)
        encode = getattr(self.tokenizer, "encode_plus", None) or self.tokenizer  # transformers 5.x dropped encode_plus: __call__ is the same API
        enc = encode(
            text=text, add_special_tokens=True, max_length=max_length, stride=self._stride,
            truncation=True if max_length is not None else False, return_tensors=None, return_offsets_mapping=True,
            return_attention_mask=False, return_token_type_ids=True, return_special_tokens_mask=True)
        ids, types, special, offs = enc["input_ids"], enc["token_type_ids"], enc["special_tokens_mask"], enc["offset_mapping"]
        tokens = []
        for tid, ty, sp, off in zip(ids, types, special, offs):
            if not self._add_special_tokens and sp == 1:
                continue
            start, end = (None, None) if off is None or off[0] >= off[1] else off
            tokens.append(Token(text=self.tokenizer.convert_ids_to_tokens(tid, skip_special_tokens=False), text_id=tid,
                                type_id=ty, idx=start, idx_end=end))
        return tokens
