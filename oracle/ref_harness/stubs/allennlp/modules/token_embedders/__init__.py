import torch

from .token_embedder import TokenEmbedder  # noqa: F401


@TokenEmbedder.register("embedding")
class Embedding(TokenEmbedder):
    def __init__(self, embedding_dim: int, num_embeddings: int = None, **kw) -> None:
        super().__init__()
        self.weight = torch.nn.Parameter(torch.zeros(num_embeddings or 1, embedding_dim))

    def get_output_dim(self):
        return self.weight.size(1)


class PretrainedTransformerEmbedder(TokenEmbedder):
    """Import surface only: the reference registers its own copy (custom_PTM_embedder.py:22)."""
