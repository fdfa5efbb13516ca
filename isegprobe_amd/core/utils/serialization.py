"""Constructor-argument capture for checkpoints, API-compatible with the reference's
``@serialize`` / ``load_model`` (core/utils/serialization.py:10-134): the checkpoint holds
``{"state_dict", "config"}`` where config = {"class": dotted path, "params": {name:
{"type", "value", "specified"}}}."""
import inspect
from copy import deepcopy
from functools import wraps
from importlib import import_module


def get_classname(cls):
    return f"{cls.__module__}.{cls.__qualname__}"


def _default_params(cls):
    """Keyword defaults of cls.__init__ and of its bases (most-derived wins)."""
    out = {}
    for klass in cls.mro():
        if klass is object or "__init__" not in vars(klass):
            continue
        for name, p in inspect.signature(klass.__init__).parameters.items():
            if name == "self" or p.kind in (p.VAR_POSITIONAL, p.VAR_KEYWORD) or p.default is p.empty:
                continue
            out.setdefault(name, p.default)
    return out


def serialize(init):
    names = list(inspect.signature(init).parameters)[1:]

    @wraps(init)
    def wrapped(self, *args, **kwargs):
        given = deepcopy(kwargs)
        given.update(dict(zip(names, args)))
        specified = set(given)
        for name, default in _default_params(type(self)).items():
            given.setdefault(name, default)
        params = {}
        for name, value in given.items():
            kind = "builtin"
            if inspect.isclass(value):
                kind, value = "class", get_classname(value)
            params[name] = {"type": kind, "value": value, "specified": name in specified}
        self._config = {"class": get_classname(type(self)), "params": params}
        init(self, *args, **kwargs)

    return wrapped


def get_class_from_str(path):
    """Resolve 'pkg.mod.Class'.  Reference checkpoints name ``core.model...``; when the
    top-level alias is not installed fall back to this package's mirror."""
    module, _, name = path.rpartition(".")
    try:
        return getattr(import_module(module), name)
    except ModuleNotFoundError:
        if module.startswith("core."):
            return getattr(import_module("isegprobe_amd." + module), name)
        raise


def load_model(config, eval_ritm=False, **overrides):
    """Rebuild a model from a checkpoint's config (serialization.py:61-91; ``eval_ritm`` is the reference's second
    positional parameter -- RITM-style evaluation is outside the probed path)."""
    if eval_ritm:
        raise NotImplementedError("eval_ritm=True (RITM-style evaluation) is outside the dense-feature path")
    cls = get_class_from_str(config["class"])
    defaults = _default_params(cls)
    kwargs = {}
    for name, p in config["params"].items():
        value = p["value"]
        if p["type"] == "class":
            value = get_class_from_str(value)
        if name not in defaults and not p["specified"]:
            continue
        if name in defaults and not p["specified"] and defaults[name] == value:
            continue
        kwargs[name] = value
    kwargs.update(overrides)
    return cls(**kwargs)
