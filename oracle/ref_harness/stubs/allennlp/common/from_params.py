"""allennlp/common/from_params.py + registrable.py (subset): construct objects from config dicts by inspecting
constructor annotations; registered base classes dispatch on the "type" key."""
import collections.abc
import inspect
import typing
from collections import defaultdict
from typing import Any, Callable, Dict, Optional, Type, TypeVar

from .checks import ConfigurationError
from .params import Params

T = TypeVar("T")


def _takes_kwargs(fn) -> bool:
    return any(p.kind == p.VAR_KEYWORD for p in inspect.signature(fn).parameters.values())


def _unwrap_optional(ann):
    origin = typing.get_origin(ann)
    if origin is typing.Union:
        args = [a for a in typing.get_args(ann) if a is not type(None)]
        if len(args) == 1:
            return args[0]
    return ann


def construct_arg(cls_name, name, ann, value, extras):
    ann = _unwrap_optional(ann)
    origin = typing.get_origin(ann)
    if value is None:
        return None
    if inspect.isclass(ann) and issubclass(ann, FromParams):
        if isinstance(value, (Params, dict, str)):
            p = value if isinstance(value, Params) else Params(value if isinstance(value, dict) else {"type": value})
            sub_extras = {k: v for k, v in extras.items() if _accepts(ann, k)}
            return ann.from_params(p, **sub_extras)
        return value
    if origin in (dict, collections.abc.Mapping) or origin is typing.Dict:
        args = typing.get_args(ann)
        if len(args) == 2 and inspect.isclass(_unwrap_optional(args[1])) and issubclass(_unwrap_optional(args[1]), FromParams):
            items = value.items()
            return {k: construct_arg(cls_name, f"{name}.{k}", args[1], v, extras) for k, v in items}
    if isinstance(value, Params):
        return value.as_dict()
    return value


def _accepts(cls, key) -> bool:
    try:
        sig = inspect.signature(cls.__init__)
    except (TypeError, ValueError):
        return False
    return key in sig.parameters or _takes_kwargs(cls.__init__)


def create_kwargs(constructor, cls, params: Params, **extras) -> Dict[str, Any]:
    kwargs: Dict[str, Any] = {}
    sig = inspect.signature(constructor)
    try:
        hints = typing.get_type_hints(constructor)
    except Exception:
        hints = {}
    for name, p in sig.parameters.items():
        if name == "self" or p.kind in (p.VAR_POSITIONAL, p.VAR_KEYWORD):
            continue
        ann = hints.get(name, p.annotation)
        if name in extras and name not in params:
            kwargs[name] = extras[name]
            continue
        if p.default is inspect.Parameter.empty:
            raw = params.pop(name)
        else:
            raw = params.pop(name, p.default)
            if raw is p.default:
                kwargs[name] = raw
                continue
        kwargs[name] = construct_arg(cls.__name__, name, ann, raw, extras)
    if _takes_kwargs(constructor):
        for k in list(params.keys()):
            kwargs[k] = params.pop(k, keep_as_dict=True)
    params.assert_empty(cls.__name__)
    return kwargs


class FromParams:
    @classmethod
    def from_params(cls: Type[T], params, constructor_to_call=None, constructor_to_inspect=None, **extras) -> T:
        if params is None:
            return None
        if isinstance(params, str):
            params = Params({"type": params})
        if isinstance(params, dict):
            params = Params(params)
        registered = Registrable._registry.get(cls)
        if registered is not None and constructor_to_call is None:  # a registered BASE class: dispatch on "type"
            choices = list(registered.keys())
            default = getattr(cls, "default_implementation", None)
            if default is not None and "type" not in params:
                choice = default
            else:
                choice = params.pop_choice("type", choices)
            subclass, ctor_name = cls.resolve_class_name(choice)
            ctor = getattr(subclass, ctor_name) if ctor_name else None
            if hasattr(subclass, "from_params"):
                sub_extras = {k: v for k, v in extras.items() if _accepts(subclass, k) or ctor is not None}
                return subclass.from_params(params, constructor_to_call=ctor or subclass,
                                            constructor_to_inspect=ctor or subclass.__init__, **sub_extras)
            return subclass(**params.as_dict())
        ctor_inspect = constructor_to_inspect or cls.__init__
        call = constructor_to_call or cls
        if ctor_inspect is object.__init__:
            return call()
        if "type" in params and "type" not in inspect.signature(ctor_inspect).parameters:
            params.params.pop("type")
        kwargs = create_kwargs(ctor_inspect, cls, params, **extras)
        return call(**kwargs)


class Registrable(FromParams):
    _registry: Dict[type, Dict[str, tuple]] = defaultdict(dict)
    default_implementation: Optional[str] = None

    @classmethod
    def register(cls, name: str, constructor: str = None, exist_ok: bool = False):
        registry = Registrable._registry[cls]

        def add(subclass):
            if name in registry and not exist_ok and registry[name][0] is not subclass:
                raise ConfigurationError(f"{name} already registered for {cls.__name__}")
            registry[name] = (subclass, constructor)
            return subclass

        return add

    @classmethod
    def resolve_class_name(cls, name: str):
        reg = Registrable._registry[cls]
        if name in reg:
            return reg[name]
        raise ConfigurationError(f"{name} is not a registered name for {cls.__name__}")

    @classmethod
    def by_name(cls, name: str) -> Callable:
        subclass, ctor = cls.resolve_class_name(name)
        return getattr(subclass, ctor) if ctor else subclass

    @classmethod
    def list_available(cls):
        return list(Registrable._registry[cls].keys())


class Lazy(typing.Generic[T]):
    def __init__(self, constructor):
        self._constructor = constructor

    def construct(self, **kwargs):
        return self._constructor(**kwargs)
