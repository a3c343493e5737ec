from .feedforward import FeedForward  # noqa: F401
from .text_field_embedders import TextFieldEmbedder  # noqa: F401
from .token_embedders import Embedding, TokenEmbedder  # noqa: F401
