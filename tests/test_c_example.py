"""The C ABI from C: include/sarpro_hip.h must be a valid strict-C99 header, and examples/grd_to_rgb.c (TIFF files in,
RGB TIFF out, nothing but the shared library) must reproduce the oracle."""
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "sarpro_amd")


def _build(tmp_path):
    exe = str(tmp_path / "grd_to_rgb")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-O2", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "grd_to_rgb.c"), "-L" + LIBDIR, "-lsarpro_hip", "-Wl,-rpath," + LIBDIR, "-o", exe])
    return exe


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_header_is_strict_c99_and_the_example_links(tmp_path):
    exe = _build(tmp_path)
    p = subprocess.run([exe], capture_output=True, text=True)
    assert p.returncode == 2 and "usage" in p.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("strategy", [4, 1])
def test_c_example_reproduces_the_oracle(tmp_path, strategy):
    import oracle
    import sarpro_amd as S
    from sarpro_amd import synth

    exe = _build(tmp_path)
    rows, cols = 333, 520
    b = [synth.scene_u16(rows, cols, k) for k in (0, 1)]
    paths = [str(tmp_path / f"b{k}.tif") for k in (0, 1)]
    for p, a in zip(paths, b):
        w = S.TiffWriter(p, cols, rows, 1, 16)
        w.write_rows(0, a)
        w.finish()
    out = str(tmp_path / "rgb.tif")
    r = subprocess.run([exe, paths[0], paths[1], out, str(strategy)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    t = S.TiffReader(out)
    got = np.stack([t.read_rows(0, rows, sample=s) for s in range(3)], axis=-1).astype(np.uint8)
    rc, ref, _, _ = oracle.dualpol_synrgb(b[0].astype(np.float32), b[1].astype(np.float32), strategy)
    assert rc == 0 and np.array_equal(got, ref)
