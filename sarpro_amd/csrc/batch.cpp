// batch.cpp -- save.rs orchestration without the writers, and the batch driver.
//   * sarpro_hip_process_band_resized_*: save_processed_image (save.rs:23-170) up to the raster the
//     writer receives: pipeline -> resize -> pad, device-resident.
//   * sarpro_hip_batch_dualpol_synrgb_resized_u16: process_directory_to_path semantics
//     (api/mod.rs:474-536; cli/runner.rs:277-345) for in-memory scenes: scenes are dealt to workers_per_device worker
//     threads per GPU (no collective: scenes are independent), failures are counted and -- with
//     continue_on_error -- do not stop the batch (BatchReport, api/mod.rs:453-458).
#include <algorithm>
#include <atomic>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <pthread.h>
#include <sched.h>

#include <cctype>
#include <cstdio>

#include "internal.h"

using namespace sarpro;

// "0-31,64-95\n" (sysfs cpulist syntax) -> CPU numbers; returns how many were written (at most max), -1 on a syntax error
extern "C" int sarpro_hip_host_parse_cpulist(const char *s, int *cpus, int max) {
    if (!s || (!cpus && max > 0)) return -1;
    int n = 0;
    const char *p = s;
    while (*p) {
        while (*p == ',' || isspace((unsigned char)*p)) ++p;
        if (!*p) break;
        if (!isdigit((unsigned char)*p)) return -1;
        char *e = nullptr;
        long a = strtol(p, &e, 10), b = a;
        p = e;
        if (*p == '-') {
            ++p;
            if (!isdigit((unsigned char)*p)) return -1;
            b = strtol(p, &e, 10);
            p = e;
        }
        if (b < a) return -1;
        for (long c = a; c <= b; ++c) { if (n < max) cpus[n] = (int)c; ++n; if (n >= (1 << 20)) return -1; }
    }
    return n < max ? n : max;
}

// A batch worker pulls ~55 GB/s from host memory into its GPU (pinned ring, H2D at PCIe line rate); eight of them on the wrong
// socket share one inter-socket link.  Bind the calling thread to the CPUs of the NUMA node the GPU hangs off (sysfs:
// /sys/bus/pci/devices/<bdf>/numa_node, /sys/devices/system/node/node<N>/cpulist) BEFORE its context allocates the pinned
// ring, so that the ring is first-touched there.  Returns the node, or -1 when the platform says nothing (single node, no
// sysfs, container without the files): the worker then runs unbound, as before.  SARPRO_HIP_BATCH_NO_NUMA=1 switches it off.
static int bind_thread_to_device_numa(int dev) {
    if (getenv("SARPRO_HIP_BATCH_NO_NUMA")) return -1;
    char bdf[64] = {0};
    if (hipDeviceGetPCIBusId(bdf, (int)sizeof(bdf), dev) != hipSuccess) return -1;
    for (char *c = bdf; *c; ++c) *c = (char)tolower((unsigned char)*c);
    char path[256];
    snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bdf);
    int node = -1;
    if (FILE *f = fopen(path, "r")) { if (fscanf(f, "%d", &node) != 1) node = -1; fclose(f); }
    if (node < 0) return -1;
    snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
    char list[4096] = {0};
    if (FILE *f = fopen(path, "r")) { const size_t k = fread(list, 1, sizeof(list) - 1, f); list[k] = 0; fclose(f); } else return -1;
    std::vector<int> cpus(4096);
    const int n = sarpro_hip_host_parse_cpulist(list, cpus.data(), (int)cpus.size());
    if (n <= 0) return -1;
    // only CPUs the thread may already run on: a mask set by taskset / numactl / a job scheduler is narrowed, never widened
    cpu_set_t allowed, set;
    CPU_ZERO(&allowed);
    if (pthread_getaffinity_np(pthread_self(), sizeof(allowed), &allowed) != 0) return -1;
    CPU_ZERO(&set);
    int kept = 0;
    for (int i = 0; i < n; ++i)
        if (cpus[i] >= 0 && cpus[i] < CPU_SETSIZE && CPU_ISSET(cpus[i], &allowed)) { CPU_SET(cpus[i], &set); ++kept; }
    if (kept == 0) return -1; // the node's CPUs are all outside the inherited mask: stay where the caller put us
    if (pthread_setaffinity_np(pthread_self(), sizeof(set), &set) != 0) return -1;
    return node;
}

#define HIPCHK(ctx, expr)                                                                         \
    do {                                                                                          \
        hipError_t e__ = (expr);                                                                  \
        if (e__ != hipSuccess) {                                                                  \
            (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e__);                      \
            return e__ == hipErrorOutOfMemory ? SARPRO_HIP_ERR_OOM : SARPRO_HIP_ERR_HIP;          \
        }                                                                                         \
    } while (0)
#define RETCHK(expr)                                   \
    do {                                               \
        int rc__ = (expr);                             \
        if (rc__ != SARPRO_HIP_OK) return rc__;        \
    } while (0)

static int fail(sarpro_hip_ctx *ctx, int code, const char *msg) {
    if (ctx) ctx->err = msg;
    return code;
}

static int process_band_resized(sarpro_hip_ctx *ctx, const void *in, bool is_f32, size_t rows, size_t cols, int strategy,
                                int bit_depth, size_t target_size, int pad, void *out, sarpro_hip_resize_meta *meta) {
    if (!ctx) return SARPRO_HIP_ERR_INVALID_ARG;
    if (bit_depth != SARPRO_BITDEPTH_U8 && bit_depth != SARPRO_BITDEPTH_U16) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "bad bit depth");
    if (rows * cols && (!in || !out)) return fail(ctx, SARPRO_HIP_ERR_INVALID_ARG, "null raster");
    const size_t esz = bit_depth == SARPRO_BITDEPTH_U8 ? 1 : 2;
    size_t pitch = 0, fc = 0, fr = 0;
    TimingHold hold(ctx);
    RETCHK(stage_in_2d(ctx, ctx->stage_in[0], in, rows, cols, is_f32 ? 4 : 2, &pitch));
    HIPCHK(ctx, ctx->stage_out[1].reserve(std::max<size_t>(rows, 1) * pitch * esz));
    // process_scalar_data_pipeline at native resolution (save.rs:52)
    if (is_f32)
        RETCHK(sarpro_hip_autoscale_band_f32_dev(ctx, ctx->stage_in[0].as<float>(), rows, cols, pitch, strategy, bit_depth,
                                                 ctx->stage_out[1].p, pitch, nullptr));
    else
        RETCHK(sarpro_hip_autoscale_band_u16_dev(ctx, ctx->stage_in[0].as<uint16_t>(), rows, cols, pitch, strategy, bit_depth,
                                                 ctx->stage_out[1].p, pitch, nullptr));
    // resize_image_data_with_meta (save.rs:54-63)
    RETCHK(sarpro_hip_resize_output_dims(cols, rows, target_size, pad, &fc, &fr));
    const size_t opitch = round_up(std::max<size_t>(fc, 1), 64);
    HIPCHK(ctx, ctx->stage_out[0].reserve(std::max<size_t>(fr, 1) * opitch * esz));
    sarpro_hip_resize_meta m{};
    RETCHK(resize_pad_dev(ctx, ctx->stage_out[1].p, cols, rows, pitch, target_size, (int)esz, pad, ctx->stage_out[0].p, opitch, &m));
    if (meta) *meta = m;
    return fetch_out_2d(ctx, out, ctx->stage_out[0].p, opitch * esz, fc * esz, fr);
}

extern "C" int sarpro_hip_process_band_resized_u16(sarpro_hip_ctx *ctx, const uint16_t *in, size_t rows, size_t cols, int strategy,
                                                   int bit_depth, size_t target_size, int pad, void *out,
                                                   sarpro_hip_resize_meta *meta) {
    return process_band_resized(ctx, in, false, rows, cols, strategy, bit_depth, target_size, pad, out, meta);
}

extern "C" int sarpro_hip_process_band_resized_f32(sarpro_hip_ctx *ctx, const float *in, size_t rows, size_t cols, int strategy,
                                                   int bit_depth, size_t target_size, int pad, void *out,
                                                   sarpro_hip_resize_meta *meta) {
    return process_band_resized(ctx, in, true, rows, cols, strategy, bit_depth, target_size, pad, out, meta);
}

// The batch driver proper: one worker thread + context per listed device, scenes dealt dynamically, BatchReport semantics of
// api/mod.rs:453-458, 518-526.  `run_scene(ctx, i)` processes scene i on the worker's context.
template <typename RunScene, typename StatusOut>
static int run_batch(const int *devices, int ndevices, int workers_per_device, size_t nscenes, int continue_on_error, sarpro_hip_batch_report *report,
                     RunScene run_scene, StatusOut status_out) {
    if (workers_per_device < 0 || workers_per_device > 8) return SARPRO_HIP_ERR_INVALID_ARG;
    if (workers_per_device == 0) workers_per_device = 2; // scene i's download beside scene i + 1's upload (PCIe is full duplex)
    std::memset(report, 0, sizeof(*report));
    std::atomic<size_t> next{0}, processed{0}, errors{0};
    std::atomic<bool> stop{false};
    std::atomic<int> first_error{SARPRO_HIP_OK};
    std::vector<int> status(nscenes, SARPRO_HIP_OK);
    auto worker = [&](int dev) {
        sarpro_hip_ctx *ctx = nullptr;
        bind_thread_to_device_numa(dev); // before the context (and its pinned ring) exists
        int rc = sarpro_hip_ctx_create(dev, 0, &ctx);
        if (rc != SARPRO_HIP_OK) { // this worker cannot run; the others take its share
            int expected = SARPRO_HIP_OK;
            first_error.compare_exchange_strong(expected, rc);
            return;
        }
        for (;;) {
            if (stop.load()) break;
            const size_t i = next.fetch_add(1);
            if (i >= nscenes) break;
            rc = run_scene(ctx, i);
            status[i] = rc;
            if (rc == SARPRO_HIP_OK) {
                processed.fetch_add(1);
            } else {
                errors.fetch_add(1);
                int expected = SARPRO_HIP_OK;
                first_error.compare_exchange_strong(expected, rc);
                if (!continue_on_error) stop.store(true); // api/mod.rs:518-526
            }
        }
        sarpro_hip_ctx_destroy(ctx);
    };
    std::vector<std::thread> pool;
    for (int w = 0; w < workers_per_device; ++w) // (worker-major: with fewer scenes than workers every device still gets one first)
        for (int d = 0; d < ndevices && pool.size() < std::max<size_t>(nscenes, 1); ++d) pool.emplace_back(worker, devices[d]);
    for (auto &t : pool) t.join();
    report->processed = processed.load();
    report->errors = errors.load();
    report->skipped = nscenes - report->processed - report->errors; // not attempted after a fatal error
    for (size_t i = 0; i < nscenes; ++i) status_out(i, status[i]);
    if (report->errors && !continue_on_error) return first_error.load();
    if (report->processed + report->errors == 0 && nscenes) return first_error.load() ? first_error.load() : SARPRO_HIP_ERR_NO_DEVICE;
    return SARPRO_HIP_OK;
}

extern "C" int sarpro_hip_batch_dualpol_synrgb_resized_u16(const int *devices, int ndevices, int workers_per_device, const sarpro_hip_batch_scene *scenes,
                                                           size_t nscenes, int strategy, int mode, size_t target_size, int pad,
                                                           int continue_on_error, sarpro_hip_batch_report *report) {
    if (!devices || ndevices <= 0 || (!scenes && nscenes) || !report) return SARPRO_HIP_ERR_INVALID_ARG;
    return run_batch(devices, ndevices, workers_per_device, nscenes, continue_on_error, report,
                     [&](sarpro_hip_ctx *ctx, size_t i) {
                         const sarpro_hip_batch_scene &sc = scenes[i];
                         if (sc.reader) return sarpro_hip_dualpol_synrgb_resized_stream_u16(ctx, sc.reader, sc.reader_user, sc.rows, sc.cols, strategy, mode, target_size, pad, sc.rgb_out, nullptr);
                         return sarpro_hip_dualpol_synrgb_resized_u16(ctx, sc.band1, sc.band2, sc.rows, sc.cols, strategy, mode, target_size, pad, sc.rgb_out, nullptr);
                     },
                     [&](size_t i, int st) { if (scenes[i].status_out) *scenes[i].status_out = st; });
}

// The same batch for scenes whose bands are f32 (the reference's default flow: resampled on read, sentinel1.rs:1074-1108); flags:
// SARPRO_HIP_DUALPOL_* (api/mod.rs:404-437 is SARPRO_HIP_DUALPOL_PLAIN_PIPELINE).
extern "C" int sarpro_hip_batch_dualpol_synrgb_resized_f32(const int *devices, int ndevices, int workers_per_device, const sarpro_hip_batch_scene_f32 *scenes,
                                                           size_t nscenes, int strategy, int mode, unsigned flags, size_t target_size,
                                                           int pad, int continue_on_error, sarpro_hip_batch_report *report) {
    if (!devices || ndevices <= 0 || (!scenes && nscenes) || !report) return SARPRO_HIP_ERR_INVALID_ARG;
    return run_batch(devices, ndevices, workers_per_device, nscenes, continue_on_error, report,
                     [&](sarpro_hip_ctx *ctx, size_t i) {
                         const sarpro_hip_batch_scene_f32 &sc = scenes[i];
                         return sarpro_hip_dualpol_synrgb_resized_f32(ctx, sc.band1, sc.band2, sc.rows, sc.cols, strategy, mode, flags, target_size, pad, sc.rgb_out, nullptr);
                     },
                     [&](size_t i, int st) { if (scenes[i].status_out) *scenes[i].status_out = st; });
}
