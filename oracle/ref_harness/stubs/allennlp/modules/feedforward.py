"""allennlp/modules/feedforward.py: Linear -> activation -> dropout per layer; parameters live in `_linear_layers`."""
import torch


class FeedForward(torch.nn.Module):
    def __init__(self, input_dim: int, num_layers: int, hidden_dims, activations, dropout=0.0) -> None:
        super().__init__()
        if not isinstance(hidden_dims, list):
            hidden_dims = [hidden_dims] * num_layers
        if not isinstance(activations, list):
            activations = [activations] * num_layers
        if not isinstance(dropout, list):
            dropout = [dropout] * num_layers
        assert len(hidden_dims) == len(activations) == len(dropout) == num_layers
        self._activations = torch.nn.ModuleList(activations)
        input_dims = [input_dim] + hidden_dims[:-1]
        self._linear_layers = torch.nn.ModuleList([torch.nn.Linear(i, o) for i, o in zip(input_dims, hidden_dims)])
        self._dropout = torch.nn.ModuleList([torch.nn.Dropout(p=v) for v in dropout])
        self._output_dim = hidden_dims[-1]
        self.input_dim = input_dim

    def get_output_dim(self):
        return self._output_dim

    def get_input_dim(self):
        return self.input_dim

    def forward(self, inputs: torch.Tensor) -> torch.Tensor:
        output = inputs
        for layer, activation, dropout in zip(self._linear_layers, self._activations, self._dropout):
            output = dropout(activation(layer(output)))
        return output
