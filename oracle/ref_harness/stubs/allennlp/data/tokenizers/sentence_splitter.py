class SpacySentenceSplitter:
    def __init__(self, *a, **kw):
        raise RuntimeError("allennlp stub: spacy is not available here")
