"""argv -> dataclass parsing with the semantics the reference gets from smart_arg (not installable here):

  * one flat argv shared by several parameter classes: unknown flags are ignored
    (`__from_argv__(args, error_on_unknown=False)`, gdmix.py:21-22);
  * `--key=value` and `--key value`; booleans spelled True/False (also true/false/1/0);
  * Optional[...] fields default to None; a required field without default raises;
  * `__post_init__` assertions of the dataclass run as usual.
"""
import dataclasses
import typing

_MISSING = dataclasses.MISSING


def _convert(tp, text, name):
    origin = typing.get_origin(tp)
    if origin is typing.Union:
        args = [a for a in typing.get_args(tp) if a is not type(None)]
        if text in ("None", ""):
            return None
        return _convert(args[0], text, name)
    if tp is bool:
        t = text.strip().lower()
        if t in ("true", "1", "yes"):
            return True
        if t in ("false", "0", "no"):
            return False
        raise ValueError(f"--{name}: expected True/False, got {text!r}")
    if tp is int:
        return int(text)
    if tp is float:
        return float(text)
    return text


def split_argv(argv):
    """[--k=v | --k v | --flag] -> dict of last occurrence; non-flag tokens are skipped."""
    out, i = {}, 0
    argv = list(argv)
    while i < len(argv):
        tok = argv[i]
        if isinstance(tok, str) and tok.startswith("--") and len(tok) > 2:
            body = tok[2:]
            if "=" in body:
                k, v = body.split("=", 1)
                out[k] = v
            elif i + 1 < len(argv) and not (isinstance(argv[i + 1], str) and argv[i + 1].startswith("--")):
                out[body] = argv[i + 1]
                i += 1
            else:
                out[body] = "True"
        i += 1
    return out


def from_argv(cls, argv, error_on_unknown=False):
    given = split_argv(argv)
    hints = typing.get_type_hints(cls)
    kwargs = {}
    names = set()
    for f in dataclasses.fields(cls):
        if not f.init:
            continue
        names.add(f.name)
        if f.name in given:
            kwargs[f.name] = _convert(hints[f.name], str(given[f.name]), f.name)
        elif f.default is _MISSING and f.default_factory is _MISSING:
            raise ValueError(f"missing required argument --{f.name} for {cls.__name__}")
    if error_on_unknown:
        unknown = set(given) - names
        if unknown:
            raise ValueError(f"unknown arguments for {cls.__name__}: {sorted(unknown)}")
    return cls(**kwargs)


def to_argv(obj):
    """Inverse: dataclass -> ['--k', 'v', ...] omitting None (what the workflow layer does, local_ops.py:15-23)."""
    out = []
    for f in dataclasses.fields(obj):
        v = getattr(obj, f.name)
        if v is None:
            continue
        out += [f"--{f.name}", str(v)]
    return out
