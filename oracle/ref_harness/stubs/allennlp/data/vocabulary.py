"""allennlp/data/vocabulary.py (subset): namespaces loaded from an archive's vocabulary/ directory."""
import os
from collections import defaultdict
from typing import Dict

from allennlp.common import Registrable

DEFAULT_NON_PADDED_NAMESPACES = ("*tags", "*labels")
DEFAULT_PADDING_TOKEN = "@@PADDING@@"
DEFAULT_OOV_TOKEN = "@@UNKNOWN@@"
NAMESPACE_PADDING_FILE = "non_padded_namespaces.txt"


def namespace_match(pattern: str, namespace: str) -> bool:
    if pattern[0] == "*" and namespace.endswith(pattern[1:]):
        return True
    return pattern == namespace


class Vocabulary(Registrable):
    default_implementation = "from_instances"

    def __init__(self, non_padded_namespaces=DEFAULT_NON_PADDED_NAMESPACES, padding_token=DEFAULT_PADDING_TOKEN,
                 oov_token=DEFAULT_OOV_TOKEN, **_kw) -> None:
        self._padding_token, self._oov_token = padding_token, oov_token
        self._non_padded_namespaces = set(non_padded_namespaces)
        self._token_to_index: Dict[str, Dict[str, int]] = defaultdict(dict)
        self._index_to_token: Dict[str, Dict[int, str]] = defaultdict(dict)

    def _is_padded(self, namespace: str) -> bool:
        return not any(namespace_match(p, namespace) for p in self._non_padded_namespaces)

    def _ensure(self, namespace: str):
        if namespace not in self._token_to_index and self._is_padded(namespace):
            self._token_to_index[namespace] = {self._padding_token: 0, self._oov_token: 1}
            self._index_to_token[namespace] = {0: self._padding_token, 1: self._oov_token}

    @classmethod
    def from_files(cls, directory: str, padding_token=DEFAULT_PADDING_TOKEN, oov_token=DEFAULT_OOV_TOKEN) -> "Vocabulary":
        with open(os.path.join(directory, NAMESPACE_PADDING_FILE), "r") as f:
            non_padded = [line.strip() for line in f if line.strip()]
        vocab = cls(non_padded_namespaces=non_padded, padding_token=padding_token, oov_token=oov_token)
        for fn in sorted(os.listdir(directory)):
            if fn == NAMESPACE_PADDING_FILE or fn.startswith("."):
                continue
            namespace = fn.replace(".txt", "")
            padded = vocab._is_padded(namespace)
            with open(os.path.join(directory, fn), "r", encoding="utf-8") as f:
                lines = f.read().split("\n")
            if lines and lines[-1] == "":
                lines = lines[:-1]
            start = 1 if padded else 0  # padded namespaces: index 0 is the padding token, the file starts at the OOV token
            t2i, i2t = ({padding_token: 0}, {0: padding_token}) if padded else ({}, {})
            for i, tok in enumerate(lines):
                t2i[tok] = i + start
                i2t[i + start] = tok
            vocab._token_to_index[namespace], vocab._index_to_token[namespace] = t2i, i2t
        return vocab

    def add_token_to_namespace(self, token: str, namespace: str = "tokens") -> int:
        self._ensure(namespace)
        t2i = self._token_to_index[namespace]
        if token not in t2i:
            idx = len(t2i)
            t2i[token] = idx
            self._index_to_token[namespace][idx] = token
        return t2i[token]

    def get_token_index(self, token: str, namespace: str = "tokens") -> int:
        t2i = self._token_to_index[namespace]
        if token in t2i:
            return t2i[token]
        if self._oov_token in t2i:
            return t2i[self._oov_token]
        raise KeyError(f"'{token}' not found in vocab namespace '{namespace}'")

    def get_token_from_index(self, index: int, namespace: str = "tokens") -> str:
        return self._index_to_token[namespace][index]

    def get_index_to_token_vocabulary(self, namespace: str = "tokens") -> Dict[int, str]:
        return self._index_to_token[namespace]

    def get_token_to_index_vocabulary(self, namespace: str = "tokens") -> Dict[str, int]:
        return self._token_to_index[namespace]

    def get_vocab_size(self, namespace: str = "tokens") -> int:
        return len(self._token_to_index[namespace])


Vocabulary.register("from_instances")(Vocabulary)
Vocabulary.register("from_files", constructor="from_files")(Vocabulary)
