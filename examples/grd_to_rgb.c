/* grd_to_rgb.c -- the C ABI end to end, no Python, no GDAL: two single-band u16 strip TIFFs (VV, VH as in a
 * Sentinel-1 GRD product's measurement/ folder) -> calibrate + autoscale + synthetic RGB on the GPU -> RGB TIFF.
 * The files are streamed: the TIFF reader fills pinned chunks while the previous chunk crosses PCIe and the first
 * device pass runs; the RGB streams back into the TIFF sink.
 *
 *   gcc -std=c99 -O2 -Iinclude examples/grd_to_rgb.c -Lsarpro_amd -lsarpro_hip -Wl,-rpath,$PWD/sarpro_amd -o grd_to_rgb
 *   ./grd_to_rgb vv.tiff vh.tiff out.tiff [strategy 0..6, default 4 = CLAHE]
 */
#include <stdio.h>
#include <stdlib.h>

#include "sarpro_hip.h"

int main(int argc, char **argv) {
    if (argc < 4) {
        fprintf(stderr, "usage: %s vv.tiff vh.tiff out.tiff [strategy]\n", argv[0]);
        return 2;
    }
    const int strategy = argc > 4 ? atoi(argv[4]) : SARPRO_STRATEGY_CLAHE;
    sarpro_hip_tiff *bands[2] = {NULL, NULL};
    sarpro_hip_tiff_info info[2];
    for (int b = 0; b < 2; ++b)
        if (sarpro_hip_tiff_open(argv[1 + b], &bands[b], &info[b]) != SARPRO_HIP_OK) {
            fprintf(stderr, "%s\n", sarpro_hip_tiff_last_error());
            return 1;
        }
    if (info[0].width != info[1].width || info[0].height != info[1].height) {
        fprintf(stderr, "the two bands differ in size\n");
        return 1;
    }
    sarpro_hip_tiff_writer *out = NULL;
    if (sarpro_hip_tiff_create(argv[3], info[0].width, info[0].height, 3, 8, NULL, bands[0], &out) != SARPRO_HIP_OK) {
        fprintf(stderr, "%s\n", sarpro_hip_tiff_last_error());
        return 1;
    }
    sarpro_hip_ctx *ctx = NULL;
    int rc = sarpro_hip_ctx_create(0, 0, &ctx);
    if (rc != SARPRO_HIP_OK) {
        fprintf(stderr, "no GPU context: %s\n", sarpro_hip_last_error(NULL));
        return 1;
    }
    sarpro_hip_stats stats[2];
    rc = sarpro_hip_dualpol_synrgb_stream_u16(ctx, sarpro_hip_tiff_pair_reader, bands, (size_t)info[0].height, (size_t)info[0].width,
                                              strategy, SARPRO_SYNRGB_DEFAULT, 0, sarpro_hip_tiff_row_sink, out, stats);
    if (rc != SARPRO_HIP_OK) fprintf(stderr, "processing failed (%d): %s %s\n", rc, sarpro_hip_last_error(ctx), sarpro_hip_tiff_last_error());
    const int rc2 = sarpro_hip_tiff_finish(out);
    if (rc == SARPRO_HIP_OK && rc2 == SARPRO_HIP_OK)
        printf("%llu x %llu: band 1 %.2f..%.2f dB (median %.2f), band 2 %.2f..%.2f dB (median %.2f)\n",
               (unsigned long long)info[0].width, (unsigned long long)info[0].height, stats[0].low_clip, stats[0].high_clip,
               stats[0].median_db, stats[1].low_clip, stats[1].high_clip, stats[1].median_db);
    sarpro_hip_ctx_destroy(ctx);
    sarpro_hip_tiff_close(bands[0]);
    sarpro_hip_tiff_close(bands[1]);
    return rc == SARPRO_HIP_OK && rc2 == SARPRO_HIP_OK ? 0 : 1;
}
