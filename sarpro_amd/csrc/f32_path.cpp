// f32_path.cpp -- f32-input flavour (arbitrary float samples: pol-op results, resampled reads).
// PENDING: entry points report an error until the threshold-table path lands.
#include "internal.h"

using namespace sarpro;

static int pending(sarpro_hip_ctx *ctx) {
    if (!ctx) return SARPRO_HIP_ERR_INVALID_ARG;
    ctx->err = "f32-input flavour not implemented yet";
    return SARPRO_HIP_ERR_INVALID_ARG;
}

extern "C" int sarpro_hip_autoscale_band_f32(sarpro_hip_ctx *ctx, const float *, size_t, size_t, int, int, uint8_t *, uint16_t *, sarpro_hip_stats *) { return pending(ctx); }
extern "C" int sarpro_hip_autoscale_band_f32_dev(sarpro_hip_ctx *ctx, const float *, size_t, size_t, size_t, int, int, void *, size_t, sarpro_hip_stats *) { return pending(ctx); }
extern "C" int sarpro_hip_db_mask_f32(sarpro_hip_ctx *ctx, const float *, size_t, size_t, double *, uint8_t *) { return pending(ctx); }
extern "C" int sarpro_hip_tamed_synrgb_u8_f32(sarpro_hip_ctx *ctx, const float *, size_t, size_t, int, uint8_t *) { return pending(ctx); }
extern "C" int sarpro_hip_dualpol_synrgb_f32(sarpro_hip_ctx *ctx, const float *, const float *, size_t, size_t, int, int, uint8_t *, uint8_t *, uint8_t *, sarpro_hip_stats *) { return pending(ctx); }
