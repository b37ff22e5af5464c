// libgoblin_host.so -- image output: Film::writeImage's post-processing and file formats.
//
//   bloom            GoblinImageIO.cpp:169-218
//   toneMapping      GoblinImageIO.cpp:220-236 (Reinhard '02 global operator as written there)
//   writeImage       GoblinImageIO.cpp:146-167 (dispatch on the file extension)
//   .ppm             GoblinImageIO.cpp:101-127 (ASCII P3, gamma 2.2)
//   .exr             GoblinImageIO.cpp:35-98: three HALF channels B, G, R through tinyexr.  The reference links the
//                    tinyexr copy that sits next to its sources; this writer emits the same OpenEXR 2.0 single-part
//                    scanline layout with the same float->half rule (tinyexr.h:7164-7199: round half up on the 13
//                    dropped mantissa bits) but NO_COMPRESSION instead of tinyexr's default ZIP, so pixel values are
//                    identical and any EXR reader opens the file; the bytes differ.
//
// rgb buffers are W*H*3 floats, row-major, top row first.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../../include/goblin_hip.h"
#include "../abi_guard.h"

namespace gbl_host_detail {
gbl_status fail(gbl_status st, const std::string& msg);   // scene_loader.cpp: sets gbl_host_last_error
}
using gbl_host_detail::fail;

namespace {

inline float luminance(const float* c) { return 0.212671f * c[0] + 0.715160f * c[1] + 0.072169f * c[2]; }   // GoblinColor.h

// tinyexr.h:7164-7199 float_to_half_full
uint16_t float_to_half(float value) {
    uint32_t u;
    memcpy(&u, &value, 4);
    const uint32_t sign = u >> 31, exponent = (u >> 23) & 0xffu, mantissa = u & 0x7fffffu;
    uint16_t o = 0;
    if (exponent == 0) {
        o = 0;
    } else if (exponent == 255) {
        o = static_cast<uint16_t>((31u << 10) | (mantissa ? 0x200u : 0u));
    } else {
        int newexp = static_cast<int>(exponent) - 127 + 15;
        if (newexp >= 31) {
            o = static_cast<uint16_t>(31u << 10);
        } else if (newexp <= 0) {
            if ((14 - newexp) <= 24) {
                uint32_t mant = mantissa | 0x800000u;
                o = static_cast<uint16_t>((mant >> (14 - newexp)) & 0x3ffu);
                if ((mant >> (13 - newexp)) & 1u) o = static_cast<uint16_t>(o + 1);
            }
        } else {
            o = static_cast<uint16_t>((static_cast<uint32_t>(newexp) << 10) | (mantissa >> 13));
            if (mantissa & 0x1000u) o = static_cast<uint16_t>(o + 1);
        }
    }
    return static_cast<uint16_t>(o | (sign << 15));
}

void put(std::vector<uint8_t>& b, const void* p, size_t n) {
    const uint8_t* c = static_cast<const uint8_t*>(p);
    b.insert(b.end(), c, c + n);
}
void put_str(std::vector<uint8_t>& b, const char* s) { put(b, s, strlen(s) + 1); }
void put_i32(std::vector<uint8_t>& b, int32_t v) { put(b, &v, 4); }   // little-endian host (x86-64 / the GPU box)
void put_attr(std::vector<uint8_t>& b, const char* name, const char* type, const std::vector<uint8_t>& value) {
    put_str(b, name);
    put_str(b, type);
    put_i32(b, static_cast<int32_t>(value.size()));
    put(b, value.data(), value.size());
}

}  // namespace

extern "C" {

static void gbl_host_bloom_impl(float* rgb, int32_t width, int32_t height, float bloom_radius, float bloom_weight) {
    if (bloom_radius <= 0.0f || bloom_weight <= 0.0f) return;
    int fw = static_cast<int>(std::ceil(bloom_radius * std::max(width, height))) / 2;
    if (fw <= 0) return;   // the reference would divide 0 by 0 for every pixel; a zero-tap filter is a no-op here
    std::vector<float> filter(static_cast<size_t>(fw) * fw);
    for (int y = 0; y < fw; ++y)
        for (int x = 0; x < fw; ++x) {
            float d = sqrtf(static_cast<float>(x * x + y * y)) / static_cast<float>(fw);
            filter[y * fw + x] = powf(std::max(0.0f, 1.0f - d), 4.0f);
        }
    std::vector<float> res(static_cast<size_t>(width) * height * 3, 0.0f);
    for (int y = 0; y < height; ++y)
        for (int x = 0; x < width; ++x) {
            int x0 = std::max(0, x - fw + 1), x1 = std::min(x + fw - 1, width - 1);
            int y0 = std::max(0, y - fw + 1), y1 = std::min(y + fw - 1, height - 1);
            float* out = &res[3 * (static_cast<size_t>(y) * width + x)];
            float wsum = 0.0f;
            for (int py = y0; py <= y1; ++py)
                for (int px = x0; px <= x1; ++px) {
                    int fx = std::abs(px - x), fy = std::abs(py - y);
                    if (fx == 0 && fy == 0) continue;
                    float w = filter[fy * fw + fx];
                    const float* in = rgb + 3 * (static_cast<size_t>(py) * width + px);
                    out[0] += w * in[0];
                    out[1] += w * in[1];
                    out[2] += w * in[2];
                    wsum += w;
                }
            float inv = 1.0f / wsum;   // Color::operator/= multiplies by the reciprocal
            out[0] *= inv;
            out[1] *= inv;
            out[2] *= inv;
        }
    for (size_t i = 0; i < res.size(); ++i) rgb[i] = (1.0f - bloom_weight) * rgb[i] + bloom_weight * res[i];
}
void gbl_host_bloom(float* rgb, int32_t width, int32_t height, float bloom_radius, float bloom_weight) {   // out of memory leaves the image as it was
    (void)gbl_guard([&] { gbl_host_bloom_impl(rgb, width, height, bloom_radius, bloom_weight); return GBL_OK; }, [](const std::string&) {});
}

static void gbl_host_tone_map_impl(float* rgb, int32_t width, int32_t height) {
    const size_t n = static_cast<size_t>(width) * height;
    float ywa = 0.0f;
    for (size_t i = 0; i < n; ++i) ywa += logf(1e4f + luminance(rgb + 3 * i));
    ywa = expf(ywa / (width * height));
    float invy2 = 1.0f / (ywa * ywa);
    for (size_t i = 0; i < n; ++i) {
        float y = luminance(rgb + 3 * i);
        float s = (1.0f + y * invy2) / (1.0f + y);
        rgb[3 * i] *= s;
        rgb[3 * i + 1] *= s;
        rgb[3 * i + 2] *= s;
    }
}
void gbl_host_tone_map(float* rgb, int32_t width, int32_t height) {   // out of memory leaves the image as it was
    (void)gbl_guard([&] { gbl_host_tone_map_impl(rgb, width, height); return GBL_OK; }, [](const std::string&) {});
}

static gbl_status gbl_host_write_ppm_impl(const char* path, const float* rgb, int32_t width, int32_t height) {
    FILE* fp = fopen(path, "w");
    if (!fp) return fail(GBL_ERR_IO, std::string("can not open file ") + path);
    fprintf(fp, "P3\n%d %d\n%d\n", width, height, 255);
    const float inv_gamma = 1.0f / 2.2f;
    for (size_t i = 0; i < static_cast<size_t>(width) * height; ++i) {
        int v[3];
        for (int c = 0; c < 3; ++c) {
            float g = powf(rgb[3 * i + c], inv_gamma);
            g = g < 0.0f ? 0.0f : (g > 1.0f ? 1.0f : g);   // NaN (pow of a negative) falls through clamp as in the reference
            v[c] = static_cast<int>(g * 255.0f);
        }
        fprintf(fp, "%d %d %d ", v[0], v[1], v[2]);
    }
    fclose(fp);
    return GBL_OK;
}
gbl_status gbl_host_write_ppm(const char* path, const float* rgb, int32_t width, int32_t height) {
    return gbl_guard([&] { return gbl_host_write_ppm_impl(path, rgb, width, height); }, [](const std::string&) {});
}

static gbl_status gbl_host_write_exr_impl(const char* path, const float* rgb, int32_t width, int32_t height) {
    if (width <= 0 || height <= 0) return fail(GBL_ERR_INVALID, "image size must be positive");
    std::vector<uint8_t> b;
    const uint32_t magic = 20000630u, version = 2u;   // single-part scanline, short names
    put(b, &magic, 4);
    put(b, &version, 4);
    {   // channels, alphabetical as OpenEXR requires (and as the reference orders them): B, G, R, all HALF
        std::vector<uint8_t> ch;
        for (const char* name : {"B", "G", "R"}) {
            put_str(ch, name);
            put_i32(ch, 1);          // pixel type HALF
            uint8_t plinear[4] = {0, 0, 0, 0};
            put(ch, plinear, 4);     // pLinear + 3 reserved
            put_i32(ch, 1);          // xSampling
            put_i32(ch, 1);          // ySampling
        }
        ch.push_back(0);
        put_attr(b, "channels", "chlist", ch);
    }
    {
        std::vector<uint8_t> v(1, 0);   // NO_COMPRESSION
        put_attr(b, "compression", "compression", v);
    }
    for (const char* name : {"dataWindow", "displayWindow"}) {
        std::vector<uint8_t> v;
        put_i32(v, 0); put_i32(v, 0); put_i32(v, width - 1); put_i32(v, height - 1);
        put_attr(b, name, "box2i", v);
    }
    {
        std::vector<uint8_t> v(1, 0);   // INCREASING_Y
        put_attr(b, "lineOrder", "lineOrder", v);
    }
    {
        std::vector<uint8_t> v;
        float one = 1.0f;
        put(v, &one, 4);
        put_attr(b, "pixelAspectRatio", "float", v);
    }
    {
        std::vector<uint8_t> v;
        float zero = 0.0f;
        put(v, &zero, 4); put(v, &zero, 4);
        put_attr(b, "screenWindowCenter", "v2f", v);
    }
    {
        std::vector<uint8_t> v;
        float one = 1.0f;
        put(v, &one, 4);
        put_attr(b, "screenWindowWidth", "float", v);
    }
    b.push_back(0);   // end of header
    // offset table: one entry per scanline block (1 line per block without compression)
    const size_t line_bytes = static_cast<size_t>(width) * 3 * 2;
    const uint64_t table_start = b.size();
    uint64_t offset = table_start + 8ull * height;
    for (int y = 0; y < height; ++y) {
        put(b, &offset, 8);
        offset += 8 + line_bytes;
    }
    std::vector<uint16_t> line(static_cast<size_t>(width) * 3);
    for (int y = 0; y < height; ++y) {
        put_i32(b, y);
        put_i32(b, static_cast<int32_t>(line_bytes));
        const float* row = rgb + 3 * static_cast<size_t>(y) * width;
        for (int c = 0; c < 3; ++c) {   // channel-planar within the line: B, G, R
            const int src = 2 - c;
            for (int x = 0; x < width; ++x) line[static_cast<size_t>(c) * width + x] = float_to_half(row[3 * x + src]);
        }
        put(b, line.data(), line_bytes);
    }
    FILE* fp = fopen(path, "wb");
    if (!fp) return fail(GBL_ERR_IO, std::string("can't open ") + path);
    size_t wrote = fwrite(b.data(), 1, b.size(), fp);
    fclose(fp);
    if (wrote != b.size()) return fail(GBL_ERR_IO, std::string("short write to ") + path);
    return GBL_OK;
}
gbl_status gbl_host_write_exr(const char* path, const float* rgb, int32_t width, int32_t height) {
    return gbl_guard([&] { return gbl_host_write_exr_impl(path, rgb, width, height); }, [](const std::string&) {});
}

static gbl_status gbl_host_write_image_impl(const char* path, float* rgb, int32_t width, int32_t height, int32_t tone_mapping) {
    if (!path || !rgb) return fail(GBL_ERR_INVALID, "null argument");
    std::string filename(path);
    size_t dot = filename.rfind(".");
    if (dot == std::string::npos) return gbl_host_write_ppm((filename + ".ppm").c_str(), rgb, width, height);
    std::string ext = filename.substr(dot);
    if (ext == ".ppm" || ext == ".PPM") {
        if (tone_mapping) gbl_host_tone_map(rgb, width, height);
        return gbl_host_write_ppm(path, rgb, width, height);
    }
    if (ext == ".exr" || ext == ".EXR") return gbl_host_write_exr(path, rgb, width, height);
    if (ext == ".pfm" || ext == ".PFM") return gbl_host_write_pfm(path, rgb, width, height);   // build-side extra
    return gbl_host_write_ppm((filename + ".ppm").c_str(), rgb, width, height);   // "format is not supported yet"
}
gbl_status gbl_host_write_image(const char* path, float* rgb, int32_t width, int32_t height, int32_t tone_mapping) {
    return gbl_guard([&] { return gbl_host_write_image_impl(path, rgb, width, height, tone_mapping); }, [](const std::string&) {});
}

}  // extern "C"
