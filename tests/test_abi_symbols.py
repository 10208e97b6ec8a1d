"""The C-ABI library loads without a GPU and exports every function include/sarpro_hip.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "sarpro_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sarpro_hip_\w+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from sarpro_amd import _lib
    names = declared_functions()
    assert len(names) >= 40
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert sorted(_lib.SYMBOLS) == names  # the Python binding covers the whole header


def test_no_device_is_a_loud_error_not_a_fallback():
    import sarpro_amd
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(sarpro_amd.SarproHipError) as ei:
        sarpro_amd.Context(0)
    assert ei.value.code == sarpro_amd._lib.ERR_NO_DEVICE


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "sarpro_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "sarpro_oracle" not in text and "import oracle" not in text, f
