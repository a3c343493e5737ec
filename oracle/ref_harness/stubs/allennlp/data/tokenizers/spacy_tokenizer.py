from .tokenizer import Tokenizer


@Tokenizer.register("spacy")
class SpacyTokenizer(Tokenizer):
    def __init__(self, *a, **kw):
        raise RuntimeError("allennlp stub: spacy is not available here (SpacyTokenizer is off the predict_memory.py path)")
