// internal.h -- helpers shared by the translation units of libsarpro_hip.so.
#pragma once
#include "context.h"

namespace sarpro {

// Records a start/stop event pair around a launch when the context was created with flag 1.
struct KernelTimer {
    sarpro_hip_ctx *ctx;
    bool active = false;
    hipStream_t stream; // the stream the kernel is launched on (default: the context's)
    KernelTimer(sarpro_hip_ctx *c, const char *name, hipStream_t on = nullptr);
    ~KernelTimer();
};
// Wall-clock time of a host segment (reported with the kernel times, names start with "host:").
struct HostTimer {
    sarpro_hip_ctx *ctx;
    const char *name;
    long long t0 = 0;
    HostTimer(sarpro_hip_ctx *c, const char *n);
    ~HostTimer();
};
void timing_reset(sarpro_hip_ctx *ctx);
// A composite entry point (resize flows: several public calls in a row) resets the timers once and holds them, so that
// last_kernel_times reports every kernel of the composite call and not only those of its last step.
struct TimingHold {
    sarpro_hip_ctx *ctx;
    explicit TimingHold(sarpro_hip_ctx *c);
    ~TimingHold();
};
size_t round_up(size_t x, size_t m);
int get_plan(sarpro_hip_ctx *ctx, size_t rows_total, size_t cols, size_t row0, size_t rows_local, int vecw,
             StripePlan **out);
int stage_in_2d(sarpro_hip_ctx *ctx, DevBuf &buf, const void *host, size_t rows, size_t cols, size_t esz,
                size_t *pitch_elems);
int fetch_out_2d(sarpro_hip_ctx *ctx, void *host, const void *dev, size_t pitch_bytes, size_t row_bytes, size_t rows);

// one band -> final u8 raster on the device (pipeline.rs:42 at U8; tamed: 1 copol / 2 crosspol -> autoscale.rs:710)
// one band through the row reader into a pitched device raster: pinned ring, hipMemcpyAsync on the side stream,
// the reader fills chunk k + 1 while chunk k crosses PCIe (api.cpp, streaming ingest)
int stream_upload_band(sarpro_hip_ctx *ctx, sarpro_hip_row_reader reader, void *user, int band, size_t rows, size_t cols,
                       uint16_t *d_dst, size_t pitch, size_t chunk_rows);
int band_u8_dev(sarpro_hip_ctx *ctx, const uint16_t *d_in, size_t rows, size_t cols, size_t in_pitch, int strategy, int tamed,
                uint8_t *d_out, size_t out_pitch);
// the same for a ROW STRIPE of the band (rows [row0, row0 + rows_local) of rows_total; the context holds a communicator): every global
// quantity is all-reduced, the stripe's levels are those of the one-piece raster.  A rank with an empty stripe joins the reductions.
int stripe_run_f32_tamed(sarpro_hip_ctx *ctx, const float *d_in, size_t rows_total, size_t cols, size_t row0, size_t rows_local, size_t in_pitch, int tamed,
                         uint8_t *d_out, size_t out_pitch); // f32_path.cpp: a9 over row stripes
int band_u8_stripe_dev(sarpro_hip_ctx *ctx, const uint16_t *d_in, size_t rows_total, size_t cols, size_t row0, size_t rows_local, size_t in_pitch,
                       int strategy, int tamed, uint8_t *d_out, size_t out_pitch);
// composition of a part (n of n_total pixels; `reduce`: the parts are the ranks') of a flat dual-pol u8 product (api.cpp)
int synrgb_flat_dev(sarpro_hip_ctx *ctx, int mode, int strategy, const uint8_t *d_b1, const uint8_t *d_b2, size_t n, size_t n_total, bool reduce,
                    uint8_t *d_rgb);
// the same band up to its DN -> final u8 TABLE (percentile strategies on the device chain: the table is the whole autoscale), for
// a consumer that applies it itself (the horizontal resize pass).  out->lut == nullptr: this band / strategy has no such table
// (CLAHE, host route): take band_u8_dev.  The table and the state it points to live in the context until its next chain.
struct ResizeLutSrc;
int band_u8_table_dev(sarpro_hip_ctx *ctx, const uint16_t *d_in, size_t rows, size_t cols, size_t in_pitch, int strategy, int tamed, ResizeLutSrc *out);
// both bands of a dual-pol product in one chain (nb = 2; Tamed: band 0 copol, band 1 crosspol) or one band (nb = 1, as above); out[nb]
int bands_u8_table_dev(sarpro_hip_ctx *ctx, const uint16_t *const d_in[], int nb, size_t rows, size_t cols, size_t in_pitch, int strategy, int tamed,
                       ResizeLutSrc *out);
int comm_allreduce_sum_u64_async(sarpro_hip_ctx *ctx, uint64_t *d_buf, size_t count);
void comm_replay_rewind(sarpro_hip_ctx *ctx);   // start of a stripe call: COMM_REPLAY answers from the first recorded buffer again (COMM_RECORD: forget the old ones)
void comm_saved_release(sarpro_hip_ctx *ctx);
void comm_abort_local_group(sarpro_hip_ctx *ctx); // a rank of an in-process group failed outside a collective: release its peers (comm.cpp)
// lut_src != nullptr: d_in is the u16 DN raster (in_pitch in u16 elements, elem_size 1 = the output's) and the horizontal pass reads
// it through the table; returns kResizeLutUnsupported (nothing enqueued) when that form does not apply to this shape
constexpr int kResizeLutUnsupported = 0x5251;
int resize_pad_dev(sarpro_hip_ctx *ctx, const void *d_in, size_t cols, size_t rows, size_t in_pitch, size_t target_size,
                   int elem_size, int pad, void *d_out, size_t out_pitch, sarpro_hip_resize_meta *meta, const ResizeLutSrc *lut_src = nullptr);

} // namespace sarpro
