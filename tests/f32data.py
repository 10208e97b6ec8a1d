"""f32 test rasters: what the f32 flavour sees in the reference (pol-op results, resampled reads)."""
import numpy as np

from sarpro_amd import synth


def ratio_scene(rows, cols):
    """ratio_arrays(VV, VH) of the synthetic scene (ops.rs:10-19): heavy-tailed positive floats + zeros."""
    a = synth.scene_u16(rows, cols, 0).astype(np.float32)
    b = synth.scene_u16(rows, cols, 1).astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.where(np.abs(b) > np.float32(1e-10), a / b, np.float32(0)).astype(np.float32)


def resampled_scene(rows, cols, band=0, seed=None):
    """Non-integer amplitudes, as a downsample-on-read produces (sentinel1.rs:1074-1108)."""
    a = (synth.scene_u16(rows * 2, cols * 2, band) if seed is None else synth.scene_u16(rows * 2, cols * 2, band, seed=synth.SEED_SCENE_A + seed)).astype(np.float32)
    return (0.25 * (a[0::2, 0::2] + a[1::2, 0::2] + a[0::2, 1::2] + a[1::2, 1::2])).astype(np.float32)


def nasty_scene(rows, cols, seed=0):
    """Signed / tiny / huge / NaN samples: ndiff-like data plus specials."""
    rng = np.random.default_rng(seed)
    x = np.exp(rng.standard_normal((rows, cols)) * 3.0).astype(np.float32)
    x.ravel()[::17] *= -1.0                     # negatives are invalid (dB floor)
    x.ravel()[5::29] = 0.0
    x.ravel()[7::31] = np.nan
    x.ravel()[11::37] = np.float32(1e-5)        # exactly -50 dB: invalid (`>` in pipeline.rs:22)
    x.ravel()[13::41] = np.float32(1.0000001e-5)
    x.ravel()[3::43] = np.float32(3e38)
    x.ravel()[2::47] = np.float32(1e-30)
    return x
