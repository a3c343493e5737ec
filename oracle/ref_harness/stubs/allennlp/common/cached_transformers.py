"""allennlp/common/cached_transformers.py: model / tokenizer cache keyed by name.  There is no hub access here, so a
model name (e.g. "bert-base-uncased") resolves through the environment variable ALLENNLP_STUB_MODEL_DIR, which the
harness points at a directory holding a synthetic checkpoint + vocab.txt of the same architecture."""
import os

_models = {}
_tokenizers = {}


def _resolve(name: str) -> str:
    if os.path.isdir(name):
        return name
    d = os.environ.get("ALLENNLP_STUB_MODEL_DIR")
    if not d or not os.path.isdir(d):
        raise RuntimeError(f"allennlp stub: cannot resolve pretrained model {name!r} offline (set ALLENNLP_STUB_MODEL_DIR)")
    return d


def get(model_name: str, make_copy: bool, override_weights_file=None, override_weights_strip_prefix=None, **kwargs):
    import copy

    from transformers import AutoModel

    key = _resolve(model_name)
    if key not in _models:
        _models[key] = AutoModel.from_pretrained(key, **kwargs)
    return copy.deepcopy(_models[key]) if make_copy else _models[key]


def get_tokenizer(model_name: str, **kwargs):
    from transformers import BertTokenizerFast

    key = _resolve(model_name)
    if key not in _tokenizers:
        vocab_file = os.path.join(key, "vocab.txt")
        with open(vocab_file, "r", encoding="utf-8") as f:
            table = {line.rstrip("\n"): i for i, line in enumerate(f)}
        import inspect

        if "vocab" in inspect.signature(BertTokenizerFast.__init__).parameters:  # transformers 5.x
            tok = BertTokenizerFast(vocab=table, do_lower_case=True, **kwargs)
        else:  # transformers 4.x (the reference pins 4.1.0)
            tok = BertTokenizerFast(vocab_file=vocab_file, do_lower_case=True, **kwargs)
        assert tok.vocab_size == len(table), (tok.vocab_size, len(table))
        _tokenizers[key] = tok
    return _tokenizers[key]
