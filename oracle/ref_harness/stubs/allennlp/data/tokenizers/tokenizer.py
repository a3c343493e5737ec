from typing import List

from allennlp.common import Registrable

from .token_class import Token


class Tokenizer(Registrable):
    default_implementation = "spacy"

    def tokenize(self, text: str) -> List[Token]:
        raise NotImplementedError

    def batch_tokenize(self, texts: List[str]) -> List[List[Token]]:
        return [self.tokenize(t) for t in texts]
