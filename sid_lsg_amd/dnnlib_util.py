"""The slice of `dnnlib` the hot path depends on (dnnlib/util.py:39-52, 255-302): EasyDict and the
construct-object-by-dotted-name seam (boundary B4 of SURVEY.md section 8) through which the
reference selects its optimizers and datasets.  Written from the interface description; same names,
argument meaning and error behaviour (ImportError carrying the unresolved name).
"""
import importlib
from typing import Any


class EasyDict(dict):
    """dict with attribute access."""

    def __getattr__(self, name: str) -> Any:
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name: str, value: Any) -> None:
        self[name] = value

    def __delattr__(self, name: str) -> None:
        del self[name]


def get_obj_by_name(name: str) -> Any:
    parts = name.split('.')
    # try the longest importable module prefix first, then walk attributes
    for i in range(len(parts), 0, -1):
        mod_name = '.'.join(parts[:i])
        try:
            obj = importlib.import_module(mod_name)
        except ImportError as e:
            missing = getattr(e, 'name', None) or ''
            if missing and not (mod_name == missing or mod_name.startswith(missing + '.')):
                raise   # the module exists but one of ITS imports failed: surface that error
            continue
        try:
            for p in parts[i:]:
                obj = getattr(obj, p)
            return obj
        except AttributeError:
            continue
    raise ImportError(name)


def call_func_by_name(*args, func_name: str = None, **kwargs) -> Any:
    assert func_name is not None
    fn = get_obj_by_name(func_name)
    assert callable(fn)
    return fn(*args, **kwargs)


def construct_class_by_name(*args, class_name: str = None, **kwargs) -> Any:
    return call_func_by_name(*args, func_name=class_name, **kwargs)


def format_time(seconds) -> str:
    s = int(round(seconds))
    if s < 60:
        return f'{s}s'
    if s < 3600:
        return f'{s // 60}m {s % 60:02d}s'
    if s < 86400:
        return f'{s // 3600}h {(s // 60) % 60:02d}m {s % 60:02d}s'
    return f'{s // 86400}d {(s // 3600) % 24:02d}h {(s // 60) % 60:02d}m'
