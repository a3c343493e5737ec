"""allennlp/modules/text_field_embedders: BasicTextFieldEmbedder — one TokenEmbedder per indexer key, registered as
`token_embedder_<key>`; forward passes each indexer's tensors to the embedder as keyword arguments."""
import inspect
from typing import Dict

import torch

from allennlp.common import Registrable
from allennlp.modules.token_embedders.token_embedder import TokenEmbedder


class TextFieldEmbedder(torch.nn.Module, Registrable):
    default_implementation = "basic"

    def get_output_dim(self) -> int:
        raise NotImplementedError


@TextFieldEmbedder.register("basic")
class BasicTextFieldEmbedder(TextFieldEmbedder):
    def __init__(self, token_embedders: Dict[str, TokenEmbedder]) -> None:
        super().__init__()
        self._token_embedders = token_embedders
        for key, embedder in token_embedders.items():
            self.add_module("token_embedder_%s" % key, embedder)
        self._ordered_embedder_keys = sorted(self._token_embedders.keys())

    def get_output_dim(self) -> int:
        return sum(e.get_output_dim() for e in self._token_embedders.values())

    def forward(self, text_field_input, num_wrapping_dims: int = 0, **kwargs) -> torch.Tensor:
        if sorted(self._token_embedders.keys()) != sorted(text_field_input.keys()):
            raise ValueError("Mismatched token keys: %s and %s" % (self._token_embedders.keys(), text_field_input.keys()))
        assert num_wrapping_dims == 0
        embedded = []
        for key in self._ordered_embedder_keys:
            embedder = getattr(self, "token_embedder_{}".format(key))
            forward_params = inspect.signature(embedder.forward).parameters
            forward_params_values = {p: kwargs[p] for p in forward_params if p in kwargs}
            missing = {p for p in forward_params if p not in kwargs}
            tensors = text_field_input[key]
            if len(tensors) == 1 and len(missing) == 1:
                token_vectors = embedder(list(tensors.values())[0], **forward_params_values)
            else:
                token_vectors = embedder(**tensors, **forward_params_values)
            if token_vectors is not None:
                embedded.append(token_vectors)
        return torch.cat(embedded, dim=-1)
