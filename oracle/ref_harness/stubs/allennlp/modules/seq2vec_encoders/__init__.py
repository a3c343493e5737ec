"""allennlp/modules/seq2vec_encoders: BertPooler = the pretrained model's own pooler (dense + tanh on token 0), deep
copied, followed by dropout (identity in eval); CnnEncoder / BagOfEmbeddingsEncoder are import surface only."""
import torch

from allennlp.common import cached_transformers


class BertPooler(torch.nn.Module):
    def __init__(self, pretrained_model, *, override_weights_file=None, override_weights_strip_prefix=None,
                 requires_grad: bool = True, dropout: float = 0.0, transformer_kwargs=None) -> None:
        super().__init__()
        if isinstance(pretrained_model, str):
            model = cached_transformers.get(pretrained_model, False, **(transformer_kwargs or {}))
        else:
            model = pretrained_model
        import copy

        self._dropout = torch.nn.Dropout(p=dropout)
        self.pooler = copy.deepcopy(model.pooler)
        for p in self.pooler.parameters():
            p.requires_grad = requires_grad
        self._embedding_dim = model.config.hidden_size

    def get_input_dim(self):
        return self._embedding_dim

    def get_output_dim(self):
        return self._embedding_dim

    def forward(self, tokens: torch.Tensor, mask: torch.BoolTensor = None, num_wrapping_dims: int = 0):
        pooled = self.pooler(tokens)
        return self._dropout(pooled)


class CnnEncoder:
    pass


class BagOfEmbeddingsEncoder:
    pass
