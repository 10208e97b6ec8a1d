#!/usr/bin/env python3
"""How much of the speculation margin the f32 blend actually uses: an instrumented build of the library
(tools/build_variant.sh specmeasure "-DSARPRO_SPEC_MEASURE") computes the reference's f64 value of EVERY sample next to the f32
value in the CLAHE apply pass and keeps the largest |y32 - y| per margin class of spec_delta (kernels.hip 4b): interior cells
(margin 1.6e-4), cells that extrapolate along y only (dy < 0: 3.0e-4), along x only (dx < 0: 2.6e-4), the corner (6.2e-4) -- round 5:
one bucket per margin, so that the one-axis margins of round 4 stand on a measurement of their own.  Scenes: the benchmark's 20000^2
generator with several seeds, plus rasters built to stress the bound (saturated and empty tiles, full-range DN).
usage: SARPRO_HIP_LIB=sarpro_amd/lib_specmeasure.so python tools/spec_margin.py [n_seeds]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("SARPRO_HIP_LIB", os.path.join(ROOT, "sarpro_amd", "lib_specmeasure.so"))
os.environ["SARPRO_HIP_NO_FUSED_RGB"] = "1"          # the apply + compose route: the instrumented kernel sees every pixel
os.environ["SARPRO_HIP_SAMPLED_HIST_MIN_PX"] = "0"
import numpy as np
import torch
import sarpro_amd as S
from sarpro_amd import AutoscaleStrategy as St, SyntheticRgbMode as Mode, synth
from sarpro_amd._lib import lib

lib.sarpro_hip_debug_spec_max_err.argtypes = [C.POINTER(C.c_float)]
def read():
    out = (C.c_float * 4)()
    assert lib.sarpro_hip_debug_spec_max_err(out) == 0
    return [float(x) for x in out]

NAMES = ("interior", "dy<0 only", "dx<0 only", "corner")
MARGINS = (1.6e-4, 3.0e-4, 2.6e-4, 6.2e-4)  # kSpecDeltaInner, kSpecDeltaEdgeY, kSpecDeltaEdgeX, kSpecDeltaEdge
def fmt(e):
    return "  ".join(f"{n} {x:.3e}" for n, x in zip(NAMES, e))

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
q = synth.q_tables()
worst = [0.0] * 4
with S.Context(0) as c:
    read()
    rows = cols = 20000; pitch = 20032
    d = [torch.empty((rows, pitch), dtype=torch.int16, device="cuda") for _ in range(2)]
    rgb = torch.zeros((rows, pitch * 3), dtype=torch.uint8, device="cuda")
    for seed in range(n):
        for k in range(2):
            c.dev_synth_scene_u16(synth.SEED_SCENE_A + 17 * seed, k, q, rows, cols, 0, rows, d[k].data_ptr(), pitch)
        c.dev_dualpol_synrgb_u16(d[0].data_ptr(), d[1].data_ptr(), rows, cols, pitch, St.Clahe, Mode.Default, rgb.data_ptr(), pitch)
        e = read(); worst = [max(a, b) for a, b in zip(worst, e)]
        print(f"synthetic scene seed {seed}: 2 x 4e8 samples, max |y32 - y|: {fmt(e)}", flush=True)
    del d, rgb
    g = torch.Generator(device="cuda"); g.manual_seed(3)
    for name, make in (("uniform full-range DN", lambda r, p: torch.randint(1, 65536, (r, p), generator=g, device="cuda", dtype=torch.int32)),
                       ("two-valued tiles (steep CDFs)", lambda r, p: torch.where(torch.rand((r, p), generator=g, device="cuda") < 0.5, 7, 60000).to(torch.int32)),
                       ("exponential speckle", lambda r, p: (torch.rand((r, p), generator=g, device="cuda").log().neg() * 400.0).clamp(0, 65535).to(torch.int32))):
        rows, cols, pitch = 6000, 7000, 7040
        b = []
        for _ in range(2):
            x = make(rows, pitch)
            b.append((x - (x >= 32768).int() * 65536).to(torch.int16))
        rgb = torch.zeros((rows, pitch * 3), dtype=torch.uint8, device="cuda")
        c.dev_dualpol_synrgb_u16(b[0].data_ptr(), b[1].data_ptr(), rows, cols, pitch, St.Clahe, Mode.Default, rgb.data_ptr(), pitch)
        e = read(); worst = [max(a, b) for a, b in zip(worst, e)]
        print(f"{name}: 2 x {rows * cols:.1e} samples, max |y32 - y|: {fmt(e)}", flush=True)
print("worst per margin class: " + "; ".join(f"{n} {w:.3e} = {w / m:.2f} of its margin {m:.1e}" for n, w, m in zip(NAMES, worst, MARGINS)))
