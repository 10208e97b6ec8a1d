"""The context attributes (sarpro_hip_ctx_set_attr / _reset_attr / _get_attr): the route switches live on the context, the
environment only supplies their initial values when a context is created."""
import numpy as np
import pytest

import oracle
import sarpro_amd as S
from sarpro_amd import AutoscaleStrategy as St, BitDepth as Bd, SarproHipError, synth

pytestmark = pytest.mark.gpu


def test_attribute_interface(monkeypatch):
    names = S.Context.attr_names()
    assert "NO_SPEC" in names and "SAMPLE_STRIDE" in names and "F32_ZONES" in names and len(names) == len(set(names)) >= 30
    monkeypatch.setenv("SARPRO_HIP_SAMPLE_STRIDE", "9")       # initial values come from the environment at creation ...
    monkeypatch.setenv("SARPRO_HIP_SPEC_FORCE", "mispredict,nospec")
    monkeypatch.setenv("SARPRO_HIP_F32_ZONES", "tiny")
    monkeypatch.setenv("SARPRO_HIP_NO_FUSED", "")              # (a bare variable is a switch that is on)
    with S.Context(0) as c:
        assert c.get_attr("SAMPLE_STRIDE") == 9 and c.get_attr("SPEC_FORCE") == 3 and c.get_attr("F32_ZONES") == 2 and c.get_attr("NO_FUSED") == 1
        assert c.get_attr("NO_SPEC") is None
        for n in ("SAMPLE_STRIDE", "SPEC_FORCE", "F32_ZONES", "NO_FUSED"):
            c.reset_attr(n)
        import os
        os.environ["SARPRO_HIP_NO_SPEC"] = "1"                 # ... and never again: a later change of the environment is not seen
        try:
            assert c.get_attr("NO_SPEC") is None
        finally:
            del os.environ["SARPRO_HIP_NO_SPEC"]
        c.set_attr("SARPRO_HIP_NO_SPEC", 1)                    # the prefix is optional
        assert c.get_attr("NO_SPEC") == 1
        c.set_attr("NO_SPEC", None)
        assert c.get_attr("NO_SPEC") is None
        with pytest.raises(SarproHipError):
            c.set_attr("NO_SUCH_SWITCH", 1)


def test_a_switch_set_on_the_context_selects_the_route():
    """NO_CHAIN on the context: the host-orchestrated phases instead of the device chain -- same raster, other kernels; two contexts
    in one process hold different values."""
    dn = synth.scene_u16(300, 448, 0)
    rc, ref = oracle.pipeline(dn.astype(np.float32), 0, int(St.Robust))
    assert rc == 0
    with S.Context(0, timing=True) as a, S.Context(0, timing=True) as b:
        b.set_attr("NO_CHAIN", 1)
        for c, chained in ((a, True), (b, False), (a, True)):
            u8, _ = c.process_scalar_data_pipeline(dn, Bd.U8, St.Robust)
            names = [n for n, _ in c.last_kernel_times()]
            assert np.array_equal(u8, ref) and ("chain_stats" in names) == chained, names
