"""allennlp/common/params.py (subset): a dict wrapper with pop semantics and override merging."""
import copy
import json
from typing import Any, Dict

from .checks import ConfigurationError


def with_fallback(preferred: Dict[str, Any], fallback: Dict[str, Any]) -> Dict[str, Any]:
    """Deep merge, `preferred` wins (allennlp.common.params.with_fallback)."""
    out = dict(fallback)
    for k, v in preferred.items():
        if isinstance(v, dict) and isinstance(out.get(k), dict):
            out[k] = with_fallback(v, out[k])
        else:
            out[k] = copy.deepcopy(v)
    return out


class Params:
    DEFAULT = object()

    def __init__(self, params: Dict[str, Any], history: str = ""):
        self.params = params if params is not None else {}
        self.history = history

    @classmethod
    def from_file(cls, path: str, params_overrides="", ext_vars=None) -> "Params":
        with open(path, "r") as f:
            file_dict = json.load(f)  # an archive's config.json is evaluated JSON
        if isinstance(params_overrides, str):
            params_overrides = json.loads(params_overrides) if params_overrides else {}
        return cls(with_fallback(preferred=params_overrides or {}, fallback=file_dict))

    def _wrap(self, v, key):
        return Params(v, self.history + key + ".") if isinstance(v, dict) else v

    def pop(self, key: str, default: Any = DEFAULT, keep_as_dict: bool = False) -> Any:
        if default is self.DEFAULT:
            if key not in self.params:
                raise ConfigurationError(f'key "{key}" is required at location "{self.history}"')
            v = self.params.pop(key)
        else:
            v = self.params.pop(key, default)
        return v if keep_as_dict else self._wrap(v, key)

    def get(self, key: str, default: Any = None) -> Any:
        return self._wrap(self.params.get(key, default), key)

    def pop_choice(self, key, choices, default_to_first_choice=False):
        default = choices[0] if default_to_first_choice else self.DEFAULT
        v = self.pop(key, default)
        if v not in choices and "." not in str(v):
            raise ConfigurationError(f"{v} not in acceptable choices for {self.history}{key}: {choices}")
        return v

    def pop_int(self, key, default=DEFAULT):
        v = self.pop(key, default)
        return None if v is None else int(v)

    def pop_float(self, key, default=DEFAULT):
        v = self.pop(key, default)
        return None if v is None else float(v)

    def pop_bool(self, key, default=DEFAULT):
        v = self.pop(key, default)
        return v if v is None or isinstance(v, bool) else str(v).lower() == "true"

    def duplicate(self) -> "Params":
        return copy.deepcopy(self)

    def as_dict(self, quiet: bool = False):
        return self.params

    def assert_empty(self, class_name: str):
        if self.params:
            raise ConfigurationError(f"Extra parameters passed to {class_name}: {self.params}")

    def __getitem__(self, key):
        return self._wrap(self.params[key], key)

    def __setitem__(self, key, value):
        self.params[key] = value

    def __contains__(self, key):
        return key in self.params

    def __iter__(self):
        return iter(self.params)

    def __len__(self):
        return len(self.params)

    def keys(self):
        return self.params.keys()

    def items(self):
        return self.params.items()
