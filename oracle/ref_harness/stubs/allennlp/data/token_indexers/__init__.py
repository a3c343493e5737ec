"""allennlp/data/token_indexers: TokenIndexer base, SingleIdTokenIndexer (import surface only) and the
PretrainedTransformerIndexer (ids come from the tokenizer's Tokens; padding value 0 / False)."""
from typing import Dict, List

import torch

from allennlp.common import Registrable

IndexedTokenList = Dict[str, List]


class TokenIndexer(Registrable):
    default_implementation = "single_id"

    def __init__(self, token_min_padding_length: int = 0) -> None:
        self._token_min_padding_length = token_min_padding_length


@TokenIndexer.register("single_id")
class SingleIdTokenIndexer(TokenIndexer):
    def __init__(self, namespace: str = "tokens", **kw) -> None:
        super().__init__()
        self.namespace = namespace


@TokenIndexer.register("pretrained_transformer")
class PretrainedTransformerIndexer(TokenIndexer):
    def __init__(self, model_name: str, namespace: str = "tags", max_length: int = None, tokenizer_kwargs=None, **kwargs) -> None:
        super().__init__(**kwargs)
        self._namespace = namespace
        self._model_name = model_name
        self._max_length = max_length  # None on the reference's configs: no segment folding

    def count_vocab_items(self, token, counter):
        pass

    def tokens_to_indices(self, tokens, vocabulary) -> IndexedTokenList:
        indices = [t.text_id for t in tokens]
        type_ids = [t.type_id if t.type_id is not None else 0 for t in tokens]
        return {"token_ids": indices, "mask": [True] * len(indices), "type_ids": type_ids}

    def get_empty_token_list(self) -> IndexedTokenList:
        return {"token_ids": [], "mask": [], "type_ids": []}

    def as_padded_tensor_dict(self, tokens: IndexedTokenList, padding_lengths: Dict[str, int]) -> Dict[str, torch.Tensor]:
        out = {}
        for key, val in tokens.items():
            n = padding_lengths[key]
            if key == "mask":
                out[key] = torch.BoolTensor(list(val[:n]) + [False] * (n - len(val)))
            else:
                out[key] = torch.LongTensor(list(val[:n]) + [0] * (n - len(val)))
        return out

    def get_padding_lengths(self, indexed_tokens: IndexedTokenList) -> Dict[str, int]:
        return {k: len(v) for k, v in indexed_tokens.items()}
