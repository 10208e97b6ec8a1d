import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ctx():
    """A sarpro_hip context on cuda:0 (GPU tests only).  Fails loudly without a device."""
    import sarpro_amd
    c = sarpro_amd.Context(0)
    yield c
    c.close()


@pytest.fixture(autouse=True)
def _route_switches_follow_the_environment(request, monkeypatch):
    """The library reads its route switches (SARPRO_HIP_<NAME>) from the environment ONCE, when a context is created; afterwards
    they are context attributes (sarpro_hip_ctx_set_attr).  The GPU tests flip switches with monkeypatch.setenv / delenv, often
    while a context is open (the session-wide `ctx` fixture, or a `with Context()` block that compares several routes): inside
    a GPU test those two calls therefore ALSO set / reset the attribute on every open context, and the attributes a test touched
    are put back when it ends.  (tests/test_gpu_attrs.py drives the attribute interface directly.)"""
    if "gpu" not in request.keywords:
        yield
        return
    import sarpro_amd
    names = set(sarpro_amd.Context.attr_names())
    touched = []
    env_set, env_del = monkeypatch.setenv, monkeypatch.delenv

    def mirror(name, value):
        if not name.startswith("SARPRO_HIP_") or name[11:] not in names:
            return
        for c in list(sarpro_amd.Context._live or ()):
            if getattr(c, "_h", None):
                touched.append((c, name[11:], c.get_attr(name[11:])))
                c.set_attr(name[11:], value)

    def setenv(name, value, *a, **k):
        env_set(name, value, *a, **k)
        mirror(name, value)

    def delenv(name, *a, **k):
        env_del(name, *a, **k)
        mirror(name, None)
    monkeypatch.setenv, monkeypatch.delenv = setenv, delenv
    yield
    for c, attr, old in reversed(touched):
        if getattr(c, "_h", None):
            c.set_attr(attr, old)


@pytest.fixture(autouse=True)
def _order_library_behind_torch(request):
    """The `_dev` entry points run on the context's own non-blocking stream, which is not ordered against torch's stream
    (include/sarpro_hip.h, "Stream ordering"): the GPU tests build inputs / outputs with torch, so every device-pointer
    call of the Python mirror first waits for torch's work.  (A product caller orders sarpro_hip_ctx_stream behind its
    producers with an event instead.)"""
    if "gpu" not in request.keywords:
        yield
        return
    import functools

    import torch

    import sarpro_amd
    saved = {}
    for cls in (sarpro_amd.Context, getattr(sarpro_amd.api, "Stripe", None), getattr(sarpro_amd.api, "StripeF32", None)):
        if cls is None:
            continue
        for name, fn in list(vars(cls).items()):
            if callable(fn) and (name.startswith("dev_") or name.startswith("stripe_") or name.startswith("phase")):
                def wrap(f):
                    @functools.wraps(f)
                    def g(*a, **k):
                        torch.cuda.synchronize()
                        return f(*a, **k)
                    return g
                saved[(cls, name)] = fn
                setattr(cls, name, wrap(fn))
    yield
    for (cls, name), fn in saved.items():
        setattr(cls, name, fn)
