"""sarpro_amd -- MI355X-native (gfx950) per-pixel raster core of bogwi/sarpro.

Product code: HIP kernels + C ABI in csrc/ (libsarpro_hip.so) and this thin host-side
mirror of the reference interface.  No CPU fallback; the oracle under oracle/ is test
infrastructure and is never imported from here.
"""
from .types import AutoscaleStrategy, BitDepth, PolarizationOperation, SyntheticRgbMode  # noqa: F401
from .api import (LocalGroup, Context, SarproHipError, Stripe, StripeF32, host_f32_merge_partials, comm_unique_id, host_clahe_bin_lut_u16,  # noqa: F401
                  host_clahe_cdfs, host_clahe_saturated_levels, host_clahe_shape_ok, host_level_lut_u16, host_stats_from_dn_hist,
                  host_stripe_plan, host_stripe_resized_rows, host_synrgb_luts, host_u8_rescale_lut, host_window,
                  host_f32_valid_threshold, host_f32_bin4096_thresholds, host_f32_level_thresholds,
                  host_f32_clahe_bin_thresholds, host_stats_from_bins4096, resize_output_dims,
                  batch_dualpol_synrgb_resized, batch_dualpol_synrgb_resized_f32, TiffReader, TiffPair, TiffWriter, host_update_geotransform)
from ._lib import Stats, F32Partial  # noqa: F401
