import torch

from allennlp.common import Registrable


class TokenEmbedder(torch.nn.Module, Registrable):
    default_implementation = "embedding"

    def get_output_dim(self) -> int:
        raise NotImplementedError
