"""The AllenNLP registry surface the reference plugs into (``DatasetReader.register("reader_memory")``
reader_memory.py:35, ``Model.register("model_memory")`` model_memory.py:39,
``TokenEmbedder.register("custom_pretrained_transformer")`` custom_PTM_embedder.py:22,
``Metric.register("siamese_measure_v1")`` custom_metric.py:55).

When AllenNLP is importable the real base classes are re-exported, so the classes of this package
register into AllenNLP's own registry and ``load_archive``/``evaluate`` find them by the names the
reference's configs use: the reference's UNMODIFIED driver runs over this plugin with
``test_siamese(..., package="memvul_amd")`` (predict_memory.py:49,59) — executed by
tests/test_reference_driver_over_plugin.py against the AllenNLP stand-in of oracle/ref_harness/stubs.
Names AllenNLP itself owns (the ``pretrained_transformer`` tokenizer / indexer / embedder, the ``basic``
text-field embedder) are then served by AllenNLP's own classes (``register_builtin``).  AllenNLP is absent
from this image, so otherwise a small stand-in with the same ``register`` / ``by_name`` / ``from_params``
behaviour is provided: construction by ``"type"`` key, nested construction of annotated/registered
sub-objects, unknown keys rejected.
"""
from __future__ import annotations

import inspect
from typing import Any, Callable, Dict, Optional, Type

try:
    from allennlp.common import Registrable  # type: ignore
    from allennlp.data import DatasetReader, Vocabulary  # type: ignore
    from allennlp.data.token_indexers import TokenIndexer  # type: ignore
    from allennlp.data.tokenizers import Tokenizer  # type: ignore
    from allennlp.models import Model  # type: ignore
    from allennlp.modules import TextFieldEmbedder, TokenEmbedder  # type: ignore
    from allennlp.training.metrics import Metric  # type: ignore

    HAVE_ALLENNLP = True
except Exception:
    HAVE_ALLENNLP = False

    class ConfigurationError(Exception):
        pass

    class Registrable:
        """Stand-in for ``allennlp.common.Registrable``."""

        _registry: Dict[type, Dict[str, type]] = {}
        default_implementation: Optional[str] = None

        @classmethod
        def register(cls, name: str, exist_ok: bool = False) -> Callable[[type], type]:
            reg = Registrable._registry.setdefault(cls, {})

            def add(sub: type) -> type:
                if name in reg and not exist_ok and reg[name] is not sub:
                    raise ConfigurationError(f"{name} already registered for {cls.__name__}")
                reg[name] = sub
                return sub

            return add

        @classmethod
        def by_name(cls, name: str) -> type:
            reg = Registrable._registry.get(cls, {})
            if name not in reg:
                raise ConfigurationError(f"{name} is not a registered name for {cls.__name__}; known: {sorted(reg)}")
            return reg[name]

        @classmethod
        def list_available(cls):
            return sorted(Registrable._registry.get(cls, {}))

        @classmethod
        def from_params(cls, params: Dict[str, Any], **extras):
            """Build from a (JSON-like) dict: ``{"type": name, **kwargs}``.  Constructor parameters whose
            annotation is a Registrable subclass (or Dict[str, Registrable]) are built recursively."""
            params = dict(params or {})
            sub = cls
            if "type" in params:
                sub = cls.by_name(params.pop("type"))
            elif cls in Registrable._registry and cls.default_implementation:
                sub = cls.by_name(cls.default_implementation)
            sig = inspect.signature(sub.__init__)
            try:  # resolve string annotations (`from __future__ import annotations`)
                import typing

                hints = typing.get_type_hints(sub.__init__)
            except Exception:
                hints = {}
            kwargs: Dict[str, Any] = {}
            accepts_kwargs = any(p.kind == p.VAR_KEYWORD for p in sig.parameters.values())
            for pname, p in sig.parameters.items():
                if pname == "self" or p.kind in (p.VAR_KEYWORD, p.VAR_POSITIONAL):
                    continue
                if pname in params:
                    kwargs[pname] = _construct(hints.get(pname, p.annotation), params.pop(pname), extras)
                elif pname in extras:
                    kwargs[pname] = extras[pname]
            if params and not accepts_kwargs:
                raise ConfigurationError(f"Extra parameters passed to {sub.__name__}: {sorted(params)}")
            kwargs.update(params if accepts_kwargs else {})
            return sub(**kwargs)

    def _construct(annotation, value, extras):
        origin = getattr(annotation, "__origin__", None)
        args = getattr(annotation, "__args__", ())
        if inspect.isclass(annotation) and issubclass(annotation, Registrable) and isinstance(value, dict):
            return annotation.from_params(value, **extras)
        if origin in (dict, Dict) and len(args) == 2 and inspect.isclass(args[1]) and issubclass(args[1], Registrable):
            return {k: (args[1].from_params(v, **extras) if isinstance(v, dict) else v) for k, v in value.items()}
        if origin is not None and type(None) in args:  # Optional[X]
            inner = [a for a in args if a is not type(None)]
            if len(inner) == 1 and value is not None:
                return _construct(inner[0], value, extras)
        return value

    class DatasetReader(Registrable):
        def __init__(self, **kwargs) -> None:
            pass

        def read(self, file_path):
            return self._read(file_path)

        def _read(self, file_path):  # pragma: no cover
            raise NotImplementedError

    class Tokenizer(Registrable):
        default_implementation = "pretrained_transformer"

    class TokenIndexer(Registrable):
        default_implementation = "pretrained_transformer"

    class TokenEmbedder(Registrable):
        pass

    class TextFieldEmbedder(Registrable):
        default_implementation = "basic"

    class Metric(Registrable):
        def get_metric(self, reset: bool):  # pragma: no cover
            raise NotImplementedError

        def reset(self) -> None:  # pragma: no cover
            pass

    class Model(Registrable):
        def __init__(self, vocab=None, regularizer=None) -> None:
            self.vocab = vocab
            self.training = False

        def eval(self):
            self.training = False
            return self

        def train(self, mode: bool = True):
            self.training = mode
            return self

    class Vocabulary:
        """Just enough of ``allennlp.data.Vocabulary`` for the ``labels`` namespace the model reads
        (model_memory.py:58-61,67)."""

        def __init__(self, namespaces: Optional[Dict[str, list]] = None) -> None:
            self._t2i: Dict[str, Dict[str, int]] = {}
            self._i2t: Dict[str, Dict[int, str]] = {}
            for ns, toks in (namespaces or {}).items():
                for t in toks:
                    self.add_token_to_namespace(t, ns)

        def add_token_to_namespace(self, token: str, namespace: str = "tokens") -> int:
            t2i = self._t2i.setdefault(namespace, {})
            i2t = self._i2t.setdefault(namespace, {})
            if token not in t2i:
                t2i[token] = len(t2i)
                i2t[t2i[token]] = token
            return t2i[token]

        def get_token_index(self, token: str, namespace: str = "tokens") -> int:
            return self._t2i[namespace][token]

        def get_index_to_token_vocabulary(self, namespace: str = "tokens") -> Dict[int, str]:
            return dict(self._i2t.get(namespace, {}))

        def get_vocab_size(self, namespace: str = "tokens") -> int:
            return len(self._t2i.get(namespace, {}))

        @classmethod
        def from_files(cls, directory: str):
            """AllenNLP archive layout: ``vocabulary/<namespace>.txt`` one token per line
            (``non_padded_namespaces.txt`` is metadata)."""
            import os

            v = cls()
            for fn in sorted(os.listdir(directory)):
                if fn.endswith(".txt") and fn != "non_padded_namespaces.txt":
                    ns = fn[:-4]
                    with open(os.path.join(directory, fn), encoding="utf-8") as f:
                        for line in f:
                            v.add_token_to_namespace(line.rstrip("\n"), ns)
            return v


def register_builtin(base, name: str):
    """Class decorator: register under a name that AllenNLP ITSELF owns (its ``pretrained_transformer`` tokenizer /
    indexer / embedder, its ``basic`` text-field embedder).  Stand-in mode: a plain ``base.register(name)``; with AllenNLP
    importable: a no-op — AllenNLP's own implementation serves the name (re-registering it is a ConfigurationError there),
    and this package's class stays importable for direct use."""
    if HAVE_ALLENNLP:
        return lambda cls: cls
    return base.register(name)
