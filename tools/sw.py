"""Route switches for the tools: the library reads SARPRO_HIP_<NAME> once, when a context is created, and takes changes through
sarpro_hip_ctx_set_attr afterwards -- these helpers change the environment (for contexts created later) AND every open context."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sarpro_amd as S

_names = None


def _live(name):
    global _names
    if _names is None:
        _names = frozenset(S.Context.attr_names())
    short = name[11:] if name.startswith("SARPRO_HIP_") else name
    return short, ([c for c in list(S.Context._live or ()) if getattr(c, "_h", None)] if short in _names else [])


def set(name, value="1"):
    os.environ[name] = str(value)
    short, ctxs = _live(name)
    for c in ctxs:
        c.set_attr(short, str(value))


def pop(name, *_):
    os.environ.pop(name, None)
    short, ctxs = _live(name)
    for c in ctxs:
        c.reset_attr(short)


def update(env):
    for k, v in env.items():
        pop(k) if v is None else set(k, v)
