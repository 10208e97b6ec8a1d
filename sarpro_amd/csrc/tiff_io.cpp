// tiff_io.cpp -- host ingest / egress shims: a reader for the uncompressed strip TIFFs
// that Sentinel-1 GRD measurement rasters are (io/gdal.rs:107-141 reads them through GDAL RasterIO), and
// an uncompressed strip TIFF / BigTIFF sink (io/writers/tiff.rs:6-78 writes through GDAL's GTiff driver).
// They let an end-to-end run start from files and end in files on a box without GDAL; the decoders stay
// host plug-ins behind the row reader / row sink callbacks of sarpro_hip_dualpol_synrgb_stream_u16.
//
// Scope, on purpose: baseline TIFF 6.0 + BigTIFF, II or MM byte order, Compression = 1, strips (any
// RowsPerStrip), 8- or 16-bit unsigned samples, chunky or planar.  Tiles, compression, palettes and sub-IFDs are
// reported as SARPRO_HIP_ERR_IO with a message, not guessed at.  GeoTIFF: ModelPixelScale / ModelTiepoint are
// parsed and written, the GeoKey directory and its double / ASCII parameter tags are carried over verbatim.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "sarpro_hip.h"

namespace {

enum : uint16_t {
    kImageWidth = 256, kImageLength = 257, kBitsPerSample = 258, kCompression = 259, kPhotometric = 262,
    kStripOffsets = 273, kSamplesPerPixel = 277, kRowsPerStrip = 278, kStripByteCounts = 279, kPlanarConfig = 284,
    kExtraSamples = 338, kSampleFormat = 339, kTileWidth = 322,
    kModelPixelScale = 33550, kModelTiepoint = 33922, kGeoKeyDirectory = 34735, kGeoDoubleParams = 34736,
    kGeoAsciiParams = 34737,
};
enum : uint16_t { tBYTE = 1, tASCII = 2, tSHORT = 3, tLONG = 4, tDOUBLE = 12, tLONG8 = 16 };

thread_local std::string g_err;
int io_fail(const std::string &m) { g_err = m; return SARPRO_HIP_ERR_IO; }

size_t type_size(uint16_t t) {
    switch (t) {
    case 1: case 2: case 6: case 7: return 1;
    case 3: case 8: return 2;
    case 4: case 9: case 11: case 13: return 4;
    case 5: case 10: case 12: case 16: case 17: case 18: return 8;
    default: return 0;
    }
}

bool host_is_big_endian() { const uint16_t x = 1; return *reinterpret_cast<const uint8_t *>(&x) == 0; }

} // namespace

struct sarpro_hip_tiff {
    FILE *f = nullptr;
    bool swap = false, big = false;
    sarpro_hip_tiff_info info{};
    std::vector<uint64_t> strip_off, strip_len;
    std::vector<uint16_t> geo_keys;
    std::vector<double> geo_doubles;
    std::string geo_ascii;
    std::vector<uint8_t> scratch;

    template <typename T> T rd(const uint8_t *p) const {
        T v;
        std::memcpy(&v, p, sizeof(T));
        if (swap) { uint8_t *b = reinterpret_cast<uint8_t *>(&v); std::reverse(b, b + sizeof(T)); }
        return v;
    }
    bool read_at(uint64_t off, void *dst, size_t n) {
        if (fseeko(f, (off_t)off, SEEK_SET) != 0) return false;
        return fread(dst, 1, n, f) == n;
    }
    // the values of one IFD entry as u64 / double
    bool values(uint16_t type, uint64_t count, const uint8_t *inline_or_off, size_t inline_cap, std::vector<uint8_t> &raw) {
        const size_t ts = type_size(type);
        if (!ts) return false;
        const uint64_t bytes = ts * count;
        raw.resize((size_t)bytes);
        if (bytes <= inline_cap) { std::memcpy(raw.data(), inline_or_off, (size_t)bytes); return true; }
        const uint64_t off = big ? rd<uint64_t>(inline_or_off) : rd<uint32_t>(inline_or_off);
        return read_at(off, raw.data(), (size_t)bytes);
    }
    uint64_t as_u64(uint16_t type, const uint8_t *p) const {
        switch (type) {
        case tBYTE: return *p;
        case tSHORT: return rd<uint16_t>(p);
        case tLONG: return rd<uint32_t>(p);
        case tLONG8: return rd<uint64_t>(p);
        default: return 0;
        }
    }
};

extern "C" const char *sarpro_hip_tiff_last_error(void) { return g_err.c_str(); }

static int tiff_open_impl(const char *path, sarpro_hip_tiff **out, sarpro_hip_tiff_info *info_out);
extern "C" int sarpro_hip_tiff_open(const char *path, sarpro_hip_tiff **out, sarpro_hip_tiff_info *info_out) {
    try { return tiff_open_impl(path, out, info_out); }
    catch (...) { if (out) *out = nullptr; return io_fail("out of memory or internal error while parsing the TIFF directory"); } // never unwind across the C ABI (a failed parse may leak its handle)
}
static int tiff_open_impl(const char *path, sarpro_hip_tiff **out, sarpro_hip_tiff_info *info_out) {
    if (!path || !out) return SARPRO_HIP_ERR_INVALID_ARG;
    *out = nullptr;
    FILE *f = fopen(path, "rb");
    if (!f) return io_fail(std::string("cannot open ") + path);
    sarpro_hip_tiff *t = new sarpro_hip_tiff();
    t->f = f;
    auto bail = [&](const std::string &m) { fclose(f); delete t; return io_fail(m + " (" + path + ")"); };
    uint8_t hdr[16];
    if (fread(hdr, 1, 8, f) != 8) return bail("short header");
    const bool file_be = hdr[0] == 'M' && hdr[1] == 'M';
    if (!file_be && !(hdr[0] == 'I' && hdr[1] == 'I')) return bail("not a TIFF");
    t->swap = file_be != host_is_big_endian();
    const uint16_t magic = t->rd<uint16_t>(hdr + 2);
    uint64_t ifd = 0;
    if (magic == 42) { ifd = t->rd<uint32_t>(hdr + 4); }
    else if (magic == 43) {
        t->big = true;
        if (fread(hdr + 8, 1, 8, f) != 8 || t->rd<uint16_t>(hdr + 4) != 8) return bail("bad BigTIFF header");
        ifd = t->rd<uint64_t>(hdr + 8);
    } else return bail("not a TIFF");
    const size_t esz = t->big ? 20 : 12, cap = t->big ? 8 : 4;
    uint8_t cnt_raw[8];
    if (!t->read_at(ifd, cnt_raw, t->big ? 8 : 2)) return bail("truncated IFD");
    const uint64_t nent = t->big ? t->rd<uint64_t>(cnt_raw) : t->rd<uint16_t>(cnt_raw);
    if (nent > 4096) return bail("implausible IFD");
    std::vector<uint8_t> ents((size_t)nent * esz);
    if (!t->read_at(ifd + (t->big ? 8 : 2), ents.data(), ents.size())) return bail("truncated IFD");
    sarpro_hip_tiff_info &I = t->info;
    I.samples_per_pixel = 1; I.bits_per_sample = 1; I.compression = 1; I.planar = 1; I.sample_format = 1;
    I.rows_per_strip = ~0ull; I.big_endian = file_be; I.bigtiff = t->big;
    std::vector<uint8_t> raw;
    for (uint64_t e = 0; e < nent; ++e) {
        const uint8_t *p = ents.data() + e * esz;
        const uint16_t tag = t->rd<uint16_t>(p), type = t->rd<uint16_t>(p + 2);
        const uint64_t count = t->big ? t->rd<uint64_t>(p + 4) : t->rd<uint32_t>(p + 4);
        const uint8_t *val = p + (t->big ? 12 : 8);
        const bool wanted = tag == kImageWidth || tag == kImageLength || tag == kBitsPerSample || tag == kCompression ||
                            tag == kStripOffsets || tag == kSamplesPerPixel || tag == kRowsPerStrip || tag == kStripByteCounts ||
                            tag == kPlanarConfig || tag == kSampleFormat || tag == kTileWidth || tag == kModelPixelScale ||
                            tag == kModelTiepoint || tag == kGeoKeyDirectory || tag == kGeoDoubleParams || tag == kGeoAsciiParams;
        if (!wanted || count == 0) continue;
        if (count > (1ull << 28) || !t->values(type, count, val, cap, raw)) return bail("unreadable tag " + std::to_string(tag));
        const size_t ts = type_size(type);
        auto u = [&](uint64_t i) { return t->as_u64(type, raw.data() + i * ts); };
        switch (tag) {
        case kImageWidth: I.width = u(0); break;
        case kImageLength: I.height = u(0); break;
        case kBitsPerSample:
            I.bits_per_sample = (uint32_t)u(0);
            for (uint64_t i = 1; i < count; ++i) if (u(i) != u(0)) return bail("samples of different widths");
            break;
        case kCompression: I.compression = (uint32_t)u(0); break;
        case kSamplesPerPixel: I.samples_per_pixel = (uint32_t)u(0); break;
        case kRowsPerStrip: I.rows_per_strip = u(0); break;
        case kPlanarConfig: I.planar = (uint32_t)u(0); break;
        case kSampleFormat: I.sample_format = (uint32_t)u(0); break;
        case kTileWidth: I.tiled = 1; break;
        case kStripOffsets: t->strip_off.resize((size_t)count); for (uint64_t i = 0; i < count; ++i) t->strip_off[i] = u(i); break;
        case kStripByteCounts: t->strip_len.resize((size_t)count); for (uint64_t i = 0; i < count; ++i) t->strip_len[i] = u(i); break;
        case kModelPixelScale:
            if (type == tDOUBLE && count >= 3) { for (int i = 0; i < 3; ++i) I.pixel_scale[i] = t->rd<double>(raw.data() + 8 * i); I.has_geo |= 1; }
            break;
        case kModelTiepoint:
            if (type == tDOUBLE && count >= 6) { for (int i = 0; i < 6; ++i) I.tiepoint[i] = t->rd<double>(raw.data() + 8 * i); I.has_geo |= 2; I.tiepoint_count = (uint32_t)(count / 6); }
            break;
        case kGeoKeyDirectory:
            if (type == tSHORT) { t->geo_keys.resize((size_t)count); for (uint64_t i = 0; i < count; ++i) t->geo_keys[i] = (uint16_t)u(i); I.has_geo |= 4; }
            break;
        case kGeoDoubleParams:
            if (type == tDOUBLE) { t->geo_doubles.resize((size_t)count); for (uint64_t i = 0; i < count; ++i) t->geo_doubles[i] = t->rd<double>(raw.data() + 8 * i); }
            break;
        case kGeoAsciiParams:
            if (type == tASCII) t->geo_ascii.assign(reinterpret_cast<const char *>(raw.data()), (size_t)count);
            break;
        }
    }
    if (!I.width || !I.height) return bail("no image dimensions");
    if (I.width > 0x7FFFFFFFull || I.height > 0x7FFFFFFFull) return bail("implausible image dimensions");
    if (I.samples_per_pixel < 1 || I.samples_per_pixel > 64) return bail("implausible SamplesPerPixel");
    if (I.planar != 1 && I.planar != 2) return bail("bad PlanarConfiguration");
    if (I.rows_per_strip == 0) return bail("RowsPerStrip = 0");
    if (I.tiled) return bail("tiled TIFFs are not supported by this shim");
    if (I.compression != 1) return bail("compressed TIFFs are not supported by this shim (Compression = " + std::to_string(I.compression) + ")");
    if (I.bits_per_sample != 8 && I.bits_per_sample != 16) return bail("only 8- and 16-bit samples are supported");
    if (I.sample_format != 1) return bail("only unsigned integer samples are supported");
    if (I.rows_per_strip == ~0ull || I.rows_per_strip > I.height) I.rows_per_strip = I.height;
    const uint64_t strips_per_plane = (I.height + I.rows_per_strip - 1) / I.rows_per_strip;
    const uint64_t planes = I.planar == 2 ? I.samples_per_pixel : 1;
    if (t->strip_off.size() != strips_per_plane * planes) return bail("strip table does not match the image");
    if (!t->strip_len.empty() && t->strip_len.size() != t->strip_off.size()) return bail("strip tables of different lengths");
    { // every strip must hold its rows (StripByteCounts, when present, is checked rather than trusted)
        const uint64_t bps = I.bits_per_sample / 8, px = I.planar == 2 ? bps : bps * I.samples_per_pixel;
        for (size_t i = 0; i < t->strip_len.size(); ++i) {
            const uint64_t s_in_plane = i % strips_per_plane;
            const uint64_t rows_here = std::min<uint64_t>(I.rows_per_strip, I.height - s_in_plane * I.rows_per_strip);
            if (t->strip_len[i] < rows_here * I.width * px) return bail("strip " + std::to_string(i) + " is shorter than its rows");
        }
    }
    *out = t;
    if (info_out) *info_out = I;
    return SARPRO_HIP_OK;
}

extern "C" void sarpro_hip_tiff_close(sarpro_hip_tiff *t) {
    if (!t) return;
    if (t->f) fclose(t->f);
    delete t;
}

// rows [row0, row0 + nrows) of one sample as u16 (8-bit samples are widened)
static int tiff_read_impl(sarpro_hip_tiff *t, int sample, size_t row0, size_t nrows, uint16_t *dst, size_t dst_pitch);
extern "C" int sarpro_hip_tiff_read_rows_u16(sarpro_hip_tiff *t, int sample, size_t row0, size_t nrows, uint16_t *dst, size_t dst_pitch) {
    try { return tiff_read_impl(t, sample, row0, nrows, dst, dst_pitch); }
    catch (...) { return io_fail("out of memory while reading rows"); }
}
static int tiff_read_impl(sarpro_hip_tiff *t, int sample, size_t row0, size_t nrows, uint16_t *dst, size_t dst_pitch) {
    if (!t || !dst) return SARPRO_HIP_ERR_INVALID_ARG;
    const sarpro_hip_tiff_info &I = t->info;
    if (sample < 0 || (uint32_t)sample >= I.samples_per_pixel || row0 > I.height || nrows > I.height - row0 || dst_pitch < I.width) return io_fail("read outside the image");
    const size_t bps = I.bits_per_sample / 8;
    const bool planar = I.planar == 2;
    const size_t px_stride = planar ? bps : bps * I.samples_per_pixel, row_bytes = (size_t)I.width * px_stride;
    const uint64_t strips_per_plane = (I.height + I.rows_per_strip - 1) / I.rows_per_strip;
    for (size_t r = row0; r < row0 + nrows;) {
        const uint64_t s = r / I.rows_per_strip, in_strip = r - s * I.rows_per_strip;
        const size_t take = (size_t)std::min<uint64_t>(I.rows_per_strip - in_strip, row0 + nrows - r);
        const uint64_t off = t->strip_off[(size_t)(s + (planar ? (uint64_t)sample * strips_per_plane : 0))] + in_strip * row_bytes;
        t->scratch.resize(take * row_bytes);
        if (!t->read_at(off, t->scratch.data(), t->scratch.size())) return io_fail("short read in strip " + std::to_string(s));
        for (size_t k = 0; k < take; ++k) {
            const uint8_t *src = t->scratch.data() + k * row_bytes + (planar ? 0 : (size_t)sample * bps);
            uint16_t *d = dst + (r - row0 + k) * dst_pitch;
            if (bps == 2 && px_stride == 2 && !t->swap) std::memcpy(d, src, (size_t)I.width * 2);
            else
                for (size_t c = 0; c < I.width; ++c) {
                    const uint8_t *p = src + c * px_stride;
                    d[c] = bps == 1 ? (uint16_t)p[0] : t->rd<uint16_t>(p);
                }
        }
        r += take;
    }
    return SARPRO_HIP_OK;
}

// sarpro_hip_row_reader over two single-band files: user = sarpro_hip_tiff *[2]
extern "C" int sarpro_hip_tiff_pair_reader(void *user, int band, size_t row0, size_t nrows, uint16_t *dst, size_t dst_pitch) {
    sarpro_hip_tiff **pair = reinterpret_cast<sarpro_hip_tiff **>(user);
    if (!pair || band < 0 || band > 1) return SARPRO_HIP_ERR_INVALID_ARG;
    return sarpro_hip_tiff_read_rows_u16(pair[band], 0, row0, nrows, dst, dst_pitch);
}

// ------------------------------------------------------------------------------------------------
// writer: pixel data first (one strip per row chunk as it arrives would need the table up front, so the
// layout is fixed: RowsPerStrip rows per strip, strips contiguous from byte 16), IFD at the end.
// ------------------------------------------------------------------------------------------------
struct sarpro_hip_tiff_writer {
    FILE *f = nullptr;
    uint64_t width = 0, height = 0;
    uint32_t samples = 1, bits = 8;
    bool big = false;
    uint64_t rows_per_strip = 1, data_off = 16, row_bytes = 0;
    bool has_gt = false;
    double gt[6] = {0, 1, 0, 0, 0, -1};
    std::vector<uint16_t> geo_keys;
    std::vector<double> geo_doubles;
    std::string geo_ascii;
};

extern "C" int sarpro_hip_tiff_create(const char *path, uint64_t width, uint64_t height, uint32_t samples, uint32_t bits,
                                      const double *geotransform6, const sarpro_hip_tiff *geo_keys_from, sarpro_hip_tiff_writer **out) {
    if (!path || !out || !width || !height || !samples || (bits != 8 && bits != 16)) return SARPRO_HIP_ERR_INVALID_ARG;
    *out = nullptr;
    FILE *f = fopen(path, "wb");
    if (!f) return io_fail(std::string("cannot create ") + path);
    sarpro_hip_tiff_writer *w = new sarpro_hip_tiff_writer();
    w->f = f; w->width = width; w->height = height; w->samples = samples; w->bits = bits;
    w->row_bytes = width * samples * (bits / 8);
    w->rows_per_strip = std::max<uint64_t>(1, (8u << 20) / w->row_bytes); // ~8 MiB strips
    w->rows_per_strip = std::min(w->rows_per_strip, height);
    w->big = w->row_bytes * height + (1u << 20) >= 0xFFFFFFFFull;
    if (geotransform6) { w->has_gt = true; std::memcpy(w->gt, geotransform6, sizeof(w->gt)); }
    if (geo_keys_from) { w->geo_keys = geo_keys_from->geo_keys; w->geo_doubles = geo_keys_from->geo_doubles; w->geo_ascii = geo_keys_from->geo_ascii; }
    uint8_t hdr[16] = {0};
    const bool be = host_is_big_endian();
    hdr[0] = hdr[1] = be ? 'M' : 'I';
    const uint16_t magic = w->big ? 43 : 42;
    std::memcpy(hdr + 2, &magic, 2);
    if (w->big) { const uint16_t eight = 8; std::memcpy(hdr + 4, &eight, 2); } // offset size 8, pad 0; IFD offset patched in finish
    if (fwrite(hdr, 1, 16, f) != 16) { fclose(f); delete w; return io_fail("write failed"); }
    *out = w;
    return SARPRO_HIP_OK;
}

extern "C" int sarpro_hip_tiff_write_rows(sarpro_hip_tiff_writer *w, size_t row0, size_t nrows, const void *src, size_t src_pitch_bytes) {
    if (!w || !src) return SARPRO_HIP_ERR_INVALID_ARG;
    if (row0 + nrows > w->height || src_pitch_bytes < w->row_bytes) return io_fail("write outside the image");
    if (fseeko(w->f, (off_t)(w->data_off + row0 * w->row_bytes), SEEK_SET) != 0) return io_fail("seek failed");
    const uint8_t *p = reinterpret_cast<const uint8_t *>(src);
    if (src_pitch_bytes == w->row_bytes) {
        if (fwrite(p, 1, nrows * w->row_bytes, w->f) != nrows * w->row_bytes) return io_fail("write failed");
    } else {
        for (size_t r = 0; r < nrows; ++r)
            if (fwrite(p + r * src_pitch_bytes, 1, w->row_bytes, w->f) != w->row_bytes) return io_fail("write failed");
    }
    return SARPRO_HIP_OK;
}

// sarpro_hip_row_sink writing interleaved u8 rows: user = sarpro_hip_tiff_writer *
extern "C" int sarpro_hip_tiff_row_sink(void *user, size_t row0, size_t nrows, const uint8_t *src, size_t src_pitch_bytes) {
    return sarpro_hip_tiff_write_rows(reinterpret_cast<sarpro_hip_tiff_writer *>(user), row0, nrows, src, src_pitch_bytes);
}

extern "C" int sarpro_hip_tiff_finish(sarpro_hip_tiff_writer *w) {
    if (!w) return SARPRO_HIP_ERR_INVALID_ARG;
    struct Entry { uint16_t tag, type; uint64_t count; std::vector<uint8_t> data; };
    std::vector<Entry> ents;
    auto add = [&](uint16_t tag, uint16_t type, uint64_t count, const void *data) {
        Entry e{tag, type, count, {}};
        e.data.assign(reinterpret_cast<const uint8_t *>(data), reinterpret_cast<const uint8_t *>(data) + type_size(type) * count);
        ents.push_back(std::move(e));
    };
    auto add_u = [&](uint16_t tag, uint64_t v) {
        if (v <= 0xFFFF) { const uint16_t x = (uint16_t)v; add(tag, tSHORT, 1, &x); }
        else if (v <= 0xFFFFFFFFull) { const uint32_t x = (uint32_t)v; add(tag, tLONG, 1, &x); }
        else add(tag, tLONG8, 1, &v);
    };
    const uint64_t nstrips = (w->height + w->rows_per_strip - 1) / w->rows_per_strip;
    add_u(kImageWidth, w->width);
    add_u(kImageLength, w->height);
    { std::vector<uint16_t> b(w->samples, (uint16_t)w->bits); add(kBitsPerSample, tSHORT, w->samples, b.data()); }
    add_u(kCompression, 1);
    add_u(kPhotometric, w->samples >= 3 ? 2 : 1); // RGB | BlackIsZero
    {
        std::vector<uint64_t> off(nstrips), len(nstrips);
        for (uint64_t s = 0; s < nstrips; ++s) {
            off[s] = w->data_off + s * w->rows_per_strip * w->row_bytes;
            len[s] = std::min(w->rows_per_strip, w->height - s * w->rows_per_strip) * w->row_bytes;
        }
        if (w->big) { add(kStripOffsets, tLONG8, nstrips, off.data()); }
        else { std::vector<uint32_t> o32(off.begin(), off.end()); add(kStripOffsets, tLONG, nstrips, o32.data()); }
        add_u(kSamplesPerPixel, w->samples);
        add_u(kRowsPerStrip, w->rows_per_strip);
        if (w->big) { add(kStripByteCounts, tLONG8, nstrips, len.data()); }
        else { std::vector<uint32_t> l32(len.begin(), len.end()); add(kStripByteCounts, tLONG, nstrips, l32.data()); }
    }
    add_u(kPlanarConfig, 1);
    if (w->samples == 2 || w->samples > 3) { std::vector<uint16_t> x(w->samples - (w->samples > 3 ? 3 : 1), 0); add(kExtraSamples, tSHORT, x.size(), x.data()); }
    { std::vector<uint16_t> sf(w->samples, 1); add(kSampleFormat, tSHORT, w->samples, sf.data()); }
    if (w->has_gt) { // north-up geotransform -> ModelPixelScale + one tiepoint at pixel (0, 0)
        const double scale[3] = {w->gt[1], -w->gt[5], 0.0}, tie[6] = {0, 0, 0, w->gt[0], w->gt[3], 0};
        add(kModelPixelScale, tDOUBLE, 3, scale);
        add(kModelTiepoint, tDOUBLE, 6, tie);
    }
    if (!w->geo_keys.empty()) add(kGeoKeyDirectory, tSHORT, w->geo_keys.size(), w->geo_keys.data());
    if (!w->geo_doubles.empty()) add(kGeoDoubleParams, tDOUBLE, w->geo_doubles.size(), w->geo_doubles.data());
    if (!w->geo_ascii.empty()) add(kGeoAsciiParams, tASCII, w->geo_ascii.size(), w->geo_ascii.data());
    std::sort(ents.begin(), ents.end(), [](const Entry &a, const Entry &b) { return a.tag < b.tag; });

    uint64_t pos = w->data_off + w->height * w->row_bytes;
    pos = (pos + 7) & ~7ull;
    const size_t esz = w->big ? 20 : 12, cap = w->big ? 8 : 4;
    const uint64_t ifd_off = pos, ifd_bytes = (w->big ? 8 : 2) + ents.size() * esz + (w->big ? 8 : 4);
    uint64_t extra = ifd_off + ifd_bytes;
    std::vector<uint8_t> ifd((size_t)ifd_bytes, 0), tail;
    size_t q = 0;
    if (w->big) { const uint64_t n = ents.size(); std::memcpy(ifd.data(), &n, 8); q = 8; }
    else { const uint16_t n = (uint16_t)ents.size(); std::memcpy(ifd.data(), &n, 2); q = 2; }
    for (const Entry &e : ents) {
        std::memcpy(ifd.data() + q, &e.tag, 2);
        std::memcpy(ifd.data() + q + 2, &e.type, 2);
        if (w->big) std::memcpy(ifd.data() + q + 4, &e.count, 8);
        else { const uint32_t c = (uint32_t)e.count; std::memcpy(ifd.data() + q + 4, &c, 4); }
        uint8_t *val = ifd.data() + q + (w->big ? 12 : 8);
        if (e.data.size() <= cap) std::memcpy(val, e.data.data(), e.data.size());
        else {
            const uint64_t at = extra + tail.size();
            if (w->big) std::memcpy(val, &at, 8);
            else { const uint32_t a32 = (uint32_t)at; std::memcpy(val, &a32, 4); }
            tail.insert(tail.end(), e.data.begin(), e.data.end());
            while (tail.size() & 7) tail.push_back(0);
        }
        q += esz;
    }
    int rc = SARPRO_HIP_OK;
    if (!w->big && extra + tail.size() > 0xFFFFFFFFull) rc = io_fail("classic TIFF overflow");
    if (rc == SARPRO_HIP_OK && (fseeko(w->f, (off_t)ifd_off, SEEK_SET) != 0 || fwrite(ifd.data(), 1, ifd.size(), w->f) != ifd.size() ||
                                (!tail.empty() && fwrite(tail.data(), 1, tail.size(), w->f) != tail.size())))
        rc = io_fail("write failed");
    if (rc == SARPRO_HIP_OK) { // patch the IFD offset into the header
        if (w->big) { if (fseeko(w->f, 8, SEEK_SET) != 0 || fwrite(&ifd_off, 1, 8, w->f) != 8) rc = io_fail("write failed"); }
        else { const uint32_t o32 = (uint32_t)ifd_off; if (fseeko(w->f, 4, SEEK_SET) != 0 || fwrite(&o32, 1, 4, w->f) != 4) rc = io_fail("write failed"); }
    }
    if (fclose(w->f) != 0 && rc == SARPRO_HIP_OK) rc = io_fail("close failed");
    delete w;
    return rc;
}

// save.rs:71-81: the geotransform of a resized / padded product
extern "C" void sarpro_hip_host_update_geotransform(double gt[6], size_t cols, size_t rows, const sarpro_hip_resize_meta *m) {
    if (!gt || !m) return;
    if (m->scale_x > 0.0) gt[1] = gt[1] * ((double)cols / (double)m->final_cols);
    if (m->scale_y > 0.0) gt[5] = gt[5] * ((double)rows / (double)m->final_rows);
    gt[0] = gt[0] - (double)m->pad_left * gt[1];
    gt[3] = gt[3] - (double)m->pad_top * gt[5];
}
