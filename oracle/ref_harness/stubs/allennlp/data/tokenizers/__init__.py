from .token_class import Token  # noqa: F401
from .tokenizer import Tokenizer  # noqa: F401
from .pretrained_transformer_tokenizer import PretrainedTransformerTokenizer  # noqa: F401
from .spacy_tokenizer import SpacyTokenizer  # noqa: F401


@Tokenizer.register("whitespace")
class WhitespaceTokenizer(Tokenizer):
    def tokenize(self, text):
        return [Token(t) for t in text.split()]
