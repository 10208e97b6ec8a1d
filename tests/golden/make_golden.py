#!/usr/bin/env python3
"""Generates tests/golden/raster_core_v1.npz: small inputs + the ORACLE's outputs for them.

The reference (Rust) cannot be built or imported here and holds no golden vectors of its own
(SURVEY.md section 4), so these vectors come from the C restatement in oracle/ -- they pin the
oracle against libm / compiler drift between machines and give the GPU tests a fixture that does
not depend on re-running the oracle.  Re-run only when the oracle itself is corrected:

    python tests/golden/make_golden.py
"""
import os
import platform
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import f32data  # noqa: E402
import oracle  # noqa: E402
from sarpro_amd import synth  # noqa: E402

ROWS, COLS = 96, 112


def main():
    out = {}
    b = [synth.scene_u16(ROWS, COLS, k) for k in (0, 1)]
    flat = synth.scene_u16(ROWS, COLS, 0, q=synth.q_tables(flat=True))
    ratio = f32data.ratio_scene(ROWS, COLS)
    resampled = f32data.resampled_scene(ROWS, COLS)
    out["in_u16_band0"], out["in_u16_band1"], out["in_u16_flat"] = b[0], b[1], flat
    out["in_f32_ratio"], out["in_f32_resampled"] = ratio, resampled
    for name, x in (("band0", b[0].astype(np.float32)), ("band1", b[1].astype(np.float32)),
                    ("flat", flat.astype(np.float32)), ("ratio", ratio), ("resampled", resampled)):
        for strategy in range(7):
            for bd in (0, 1):
                rc, ref, st = oracle.pipeline(x, bd, strategy, want_stats=True)
                assert rc == 0
                out[f"out_{name}_s{strategy}_b{bd}"] = ref
                out[f"stats_{name}_s{strategy}_b{bd}"] = np.array(
                    [float(st.valid_count)] + [getattr(st, n) for n, _ in st._fields_[1:]], np.float64)
    for strategy in range(7):
        rc, rgb, u1, u2 = oracle.dualpol_synrgb(b[0].astype(np.float32), b[1].astype(np.float32), strategy)
        assert rc == 0
        out[f"rgb_s{strategy}"], out[f"rgb_u1_s{strategy}"], out[f"rgb_u2_s{strategy}"] = rgb, u1, u2
    for op in range(5):
        out[f"polop_{op}"] = oracle.polop(op, b[0].astype(np.float32), b[1].astype(np.float32))
    out["tamed_copol"] = oracle.tamed_synrgb_u8(b[0].astype(np.float32), True)
    out["tamed_crosspol"] = oracle.tamed_synrgb_u8(b[1].astype(np.float32), False)
    out["meta"] = np.array([f"glibc {platform.libc_ver()[1]}; {oracle.lib().sarpro_oracle_version().decode()}"])
    path = os.path.join(HERE, "raster_core_v1.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes;", out["meta"][0])


def api_flow(b1, b2, strategy, target, pad, plain):
    """api/mod.rs:404-437 (plain) or save.rs:317-351 (Tamed re-autoscale) on f32 bands -> resize -> pad -> synRGB, through the oracle."""
    us = []
    for k, x in enumerate((b1, b2)):
        if not plain and strategy == 5:
            u = oracle.tamed_synrgb_u8(x, k == 0)
        else:
            rc, u = oracle.pipeline(x, 0, strategy)
            assert rc == 0
        u, _ = oracle.resize_image_data_with_meta(u, target, pad)
        us.append(u)
    return oracle.synrgb(0, strategy, us[0], us[1])


def main_f32_flow():
    """raster_core_v2_f32flow.npz: the reference's DEFAULT flow -- both bands resampled on read (non-integer f32,
    sentinel1.rs:1074-1108), per-band u8, Lanczos3 resize, pad, synRGB -- for every strategy, with and without the Tamed
    re-autoscale of save.rs:324-351, at two target sizes."""
    out = {}
    b1, b2 = f32data.resampled_scene(ROWS, COLS, 0), f32data.resampled_scene(ROWS, COLS, 1)
    out["in_f32_band0"], out["in_f32_band1"] = b1, b2
    for strategy in range(7):
        for plain in (0, 1):
            for ti, (target, pad) in enumerate(((40, True), (None, False))):
                out[f"rgb_s{strategy}_plain{plain}_t{ti}"] = api_flow(b1, b2, strategy, target, pad, bool(plain))
    out["meta"] = np.array([f"glibc {platform.libc_ver()[1]}; {oracle.lib().sarpro_oracle_version().decode()}"])
    path = os.path.join(HERE, "raster_core_v2_f32flow.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes;", out["meta"][0])


if __name__ == "__main__":
    which = sys.argv[1:] or ["f32flow"]  # v1 is only regenerated on request: `make_golden.py v1`
    if "v1" in which:
        main()
    if "f32flow" in which:
        main_f32_flow()
