// Built with -fsanitize=address,undefined by tests/test_native_sanitizers.py (CPU only): the strip-TIFF reader of
// csrc/tiff_io.cpp parses files it did not write, so it is run here over a valid file, every truncation of its
// directory region and a few thousand random corruptions.  Any out-of-bounds access, overflow or leak aborts the run;
// a clean exit (status 0) is the test.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "sarpro_hip.h"

static std::vector<uint8_t> slurp(const char *p) {
    std::vector<uint8_t> v;
    FILE *f = fopen(p, "rb");
    if (!f) return v;
    fseek(f, 0, SEEK_END);
    v.resize((size_t)ftell(f));
    fseek(f, 0, SEEK_SET);
    if (fread(v.data(), 1, v.size(), f) != v.size()) v.clear();
    fclose(f);
    return v;
}
static void spit(const char *p, const std::vector<uint8_t> &v) {
    FILE *f = fopen(p, "wb");
    if (!v.empty()) fwrite(v.data(), 1, v.size(), f);
    fclose(f);
}

static int try_read(const char *path) {
    sarpro_hip_tiff *t = nullptr;
    sarpro_hip_tiff_info info;
    int rc = sarpro_hip_tiff_open(path, &t, &info);
    if (rc != SARPRO_HIP_OK) return rc;
    if (info.width <= 4096 && info.height <= 4096) {
        std::vector<uint16_t> buf((size_t)info.width * 8);
        for (uint64_t r = 0; r < info.height; r += 8) {
            const size_t n = (size_t)(info.height - r < 8 ? info.height - r : 8);
            for (uint32_t s = 0; s < info.samples_per_pixel && s < 4; ++s)
                if (sarpro_hip_tiff_read_rows_u16(t, (int)s, (size_t)r, n, buf.data(), (size_t)info.width) != SARPRO_HIP_OK) break;
        }
    }
    sarpro_hip_tiff_close(t);
    return SARPRO_HIP_OK;
}

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    const std::string dir = argv[1], good = dir + "/good.tif", bad = dir + "/bad.tif";
    // a valid 3-sample u16 file with geo tags, written by the library's own sink
    const uint64_t W = 37, H = 29;
    std::vector<uint16_t> img(W * H * 3);
    for (size_t i = 0; i < img.size(); ++i) img[i] = (uint16_t)(i * 2654435761u >> 16);
    const double gt[6] = {500000.0, 10.0, 0.0, 4.2e6, 0.0, -10.0};
    sarpro_hip_tiff_writer *w = nullptr;
    if (sarpro_hip_tiff_create(good.c_str(), W, H, 3, 16, gt, nullptr, &w) != SARPRO_HIP_OK) return 3;
    if (sarpro_hip_tiff_write_rows(w, 0, H, img.data(), W * 3 * 2) != SARPRO_HIP_OK) return 4;
    if (sarpro_hip_tiff_finish(w) != SARPRO_HIP_OK) return 5;
    if (try_read(good.c_str()) != SARPRO_HIP_OK) return 6;
    const std::vector<uint8_t> ref = slurp(good.c_str());
    if (ref.empty()) return 7;
    // every truncation of the tail (directory + tag payloads) and of the header
    const size_t data_end = 16 + W * H * 3 * 2;
    for (size_t n = data_end > 64 ? data_end - 64 : 0; n < ref.size(); ++n) { spit(bad.c_str(), std::vector<uint8_t>(ref.begin(), ref.begin() + n)); (void)try_read(bad.c_str()); }
    for (size_t n = 0; n < 32; ++n) { spit(bad.c_str(), std::vector<uint8_t>(ref.begin(), ref.begin() + n)); (void)try_read(bad.c_str()); }
    // random corruptions of the header and the directory region (1-4 bytes each), deterministic
    uint64_t s = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    int opened = 0;
    for (int it = 0; it < 4000; ++it) {
        std::vector<uint8_t> v = ref;
        const int nb = 1 + (int)(rnd() % 4);
        for (int k = 0; k < nb; ++k) {
            const size_t span = 16 + (v.size() - data_end);
            size_t pos = (size_t)(rnd() % span);
            pos = pos < 16 ? pos : data_end + (pos - 16);
            v[pos] = (uint8_t)rnd();
        }
        spit(bad.c_str(), v);
        if (try_read(bad.c_str()) == SARPRO_HIP_OK) ++opened;
    }
    printf("fuzz done: %d of 4000 corrupted files still opened\n", opened);
    remove(bad.c_str());
    remove(good.c_str());
    return 0;
}
