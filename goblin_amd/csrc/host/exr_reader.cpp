// libgoblin_host.so -- image input: loadImage (GoblinImageIO.cpp:14-34, 128-144).
//
// The reference reads textures and environment maps through tinyexr's LoadEXR (tinyexr.h:11215-11420): a single-part
// scanline or tiled OpenEXR file, HALF or FLOAT channels, returned as W*H float4 -- one channel is replicated into all
// four, otherwise R, G, B are required and A defaults to 1.  This reader is written from the OpenEXR file layout
// (magic 20000630, version, attribute list, offset table, scanline blocks) and covers what a Goblin scene can hold:
// single-part SCANLINE files, compression NONE / RLE / ZIPS / ZIP (the writers' defaults; tinyexr itself writes ZIP),
// HALF / FLOAT channels, any data window.  Tiled, multi-part, deep, PIZ / PXR24 / B44 files are GBL_ERR_UNSUPPORTED.
// The ZIP blocks need an inflate: RFC 1950 / 1951 decoder below (stored, fixed and dynamic Huffman blocks).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../../include/goblin_hip.h"
#include "../abi_guard.h"

namespace gbl_host_detail {
gbl_status fail(gbl_status st, const std::string& msg);   // scene_loader.cpp: sets gbl_host_last_error
}
using gbl_host_detail::fail;

namespace {

// ---- inflate (RFC 1951) over a zlib stream (RFC 1950) -------------------------------------------------------------
struct BitReader {
    const uint8_t* p;
    size_t n, pos = 0;
    uint32_t bits = 0;
    int count = 0;
    bool overrun = false;
    BitReader(const uint8_t* data, size_t size) : p(data), n(size) {}
    uint32_t get(int k) {   // k <= 16 bits, LSB first
        while (count < k) {
            if (pos >= n) {
                overrun = true;
                return 0;
            }
            bits |= static_cast<uint32_t>(p[pos++]) << count;
            count += 8;
        }
        const uint32_t v = bits & ((1u << k) - 1u);
        bits >>= k;
        count -= k;
        return v;
    }
    void align() {
        bits = 0;
        count = 0;
    }
};

struct Huffman {
    // canonical code: count[len], symbols sorted by (len, value)
    uint16_t count[16];
    uint16_t symbol[288];
    bool build(const uint8_t* lengths, int n) {
        memset(count, 0, sizeof(count));
        for (int i = 0; i < n; ++i) count[lengths[i]]++;
        count[0] = 0;
        int left = 1;
        for (int len = 1; len < 16; ++len) {
            left <<= 1;
            left -= count[len];
            if (left < 0) return false;   // over-subscribed
        }
        uint16_t offs[16];
        offs[1] = 0;
        for (int len = 1; len < 15; ++len) offs[len + 1] = static_cast<uint16_t>(offs[len] + count[len]);
        for (int i = 0; i < n; ++i)
            if (lengths[i]) symbol[offs[lengths[i]]++] = static_cast<uint16_t>(i);
        return true;
    }
    int decode(BitReader& br) const {
        int code = 0, first = 0, index = 0;
        for (int len = 1; len < 16; ++len) {
            code |= static_cast<int>(br.get(1));
            if (br.overrun) return -1;
            const int c = count[len];
            if (code - c < first) return symbol[index + (code - first)];
            index += c;
            first += c;
            first <<= 1;
            code <<= 1;
        }
        return -1;
    }
};

bool inflate_zlib(const uint8_t* src, size_t src_len, std::vector<uint8_t>& out, size_t expected) {
    static const uint16_t len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint16_t len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t dist_base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint16_t dist_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    if (src_len < 6) return false;
    if ((src[0] & 0x0f) != 8 || ((src[0] << 8) | src[1]) % 31 != 0 || (src[1] & 0x20)) return false;   // deflate, no preset dictionary
    BitReader br(src + 2, src_len - 2);
    out.clear();
    out.reserve(expected);
    for (;;) {
        const uint32_t last = br.get(1), type = br.get(2);
        if (br.overrun) return false;
        if (type == 0) {
            br.align();
            if (br.pos + 4 > br.n) return false;
            const uint32_t len = br.p[br.pos] | (br.p[br.pos + 1] << 8), nlen = br.p[br.pos + 2] | (br.p[br.pos + 3] << 8);
            br.pos += 4;
            if ((len ^ 0xffffu) != nlen || br.pos + len > br.n) return false;
            out.insert(out.end(), br.p + br.pos, br.p + br.pos + len);
            br.pos += len;
        } else if (type == 1 || type == 2) {
            Huffman lit, dist;
            uint8_t lengths[320];
            if (type == 1) {
                for (int i = 0; i < 144; ++i) lengths[i] = 8;
                for (int i = 144; i < 256; ++i) lengths[i] = 9;
                for (int i = 256; i < 280; ++i) lengths[i] = 7;
                for (int i = 280; i < 288; ++i) lengths[i] = 8;
                lit.build(lengths, 288);
                for (int i = 0; i < 30; ++i) lengths[i] = 5;
                dist.build(lengths, 30);
            } else {
                static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                const int nlen = static_cast<int>(br.get(5)) + 257, ndist = static_cast<int>(br.get(5)) + 1, ncode = static_cast<int>(br.get(4)) + 4;
                if (br.overrun || nlen > 286 || ndist > 30) return false;
                uint8_t cl[19] = {0};
                for (int i = 0; i < ncode; ++i) cl[order[i]] = static_cast<uint8_t>(br.get(3));
                Huffman clh;
                if (!clh.build(cl, 19)) return false;
                int i = 0;
                while (i < nlen + ndist) {
                    const int sym = clh.decode(br);
                    if (sym < 0) return false;
                    if (sym < 16) {
                        lengths[i++] = static_cast<uint8_t>(sym);
                    } else {
                        int rep, val = 0;
                        if (sym == 16) {
                            if (i == 0) return false;
                            val = lengths[i - 1];
                            rep = 3 + static_cast<int>(br.get(2));
                        } else if (sym == 17) {
                            rep = 3 + static_cast<int>(br.get(3));
                        } else {
                            rep = 11 + static_cast<int>(br.get(7));
                        }
                        if (i + rep > nlen + ndist) return false;
                        while (rep--) lengths[i++] = static_cast<uint8_t>(val);
                    }
                }
                if (br.overrun || lengths[256] == 0) return false;
                if (!lit.build(lengths, nlen) || !dist.build(lengths + nlen, ndist)) return false;
            }
            for (;;) {
                const int sym = lit.decode(br);
                if (sym < 0) return false;
                if (sym < 256) {
                    out.push_back(static_cast<uint8_t>(sym));
                } else if (sym == 256) {
                    break;
                } else {
                    const int li = sym - 257;
                    if (li >= 29) return false;
                    const size_t len = len_base[li] + br.get(len_extra[li]);
                    const int ds = dist.decode(br);
                    if (ds < 0 || ds >= 30) return false;
                    const size_t d = dist_base[ds] + br.get(dist_extra[ds]);
                    if (br.overrun || d > out.size()) return false;
                    const size_t from = out.size() - d;
                    for (size_t k = 0; k < len; ++k) out.push_back(out[from + k]);
                }
                if (out.size() > expected + 65536) return false;   // a corrupt stream may not grow without bound
            }
        } else {
            return false;
        }
        if (last) break;
    }
    return true;
}

// OpenEXR's byte shuffling behind ZIP / RLE: delta decode, then de-interleave the two halves
void unpredict(std::vector<uint8_t>& buf, std::vector<uint8_t>& out) {
    for (size_t i = 1; i < buf.size(); ++i) buf[i] = static_cast<uint8_t>(buf[i - 1] + buf[i] - 128);
    out.resize(buf.size());
    const size_t half = (buf.size() + 1) / 2;
    size_t a = 0, b = half;
    for (size_t i = 0; i < buf.size();) {
        out[i++] = buf[a++];
        if (i < buf.size()) out[i++] = buf[b++];
    }
}

bool rle_decode(const uint8_t* src, size_t n, std::vector<uint8_t>& out, size_t expected) {
    out.clear();
    size_t i = 0;
    while (i < n) {
        const int8_t c = static_cast<int8_t>(src[i++]);
        if (c < 0) {
            const size_t count = static_cast<size_t>(-static_cast<int>(c));
            if (i + count > n) return false;
            out.insert(out.end(), src + i, src + i + count);
            i += count;
        } else {
            if (i >= n) return false;
            out.insert(out.end(), static_cast<size_t>(c) + 1, src[i++]);
        }
        if (out.size() > expected) return false;
    }
    return true;
}

float half_to_float(uint16_t h) {
    const uint32_t sign = (h >> 15) & 1u, exp = (h >> 10) & 0x1fu, man = h & 0x3ffu;
    uint32_t u;
    if (exp == 0) {
        if (man == 0) {
            u = sign << 31;
        } else {   // subnormal half: normalise
            int e = -1;
            uint32_t m = man;
            do {
                ++e;
                m <<= 1;
            } while ((m & 0x400u) == 0);
            u = (sign << 31) | static_cast<uint32_t>(127 - 15 - e) << 23 | ((m & 0x3ffu) << 13);
        }
    } else if (exp == 31) {
        u = (sign << 31) | 0x7f800000u | (man << 13);
    } else {
        u = (sign << 31) | ((exp + 127 - 15) << 23) | (man << 13);
    }
    float f;
    memcpy(&f, &u, 4);
    return f;
}

struct Cursor {
    const std::vector<uint8_t>& b;
    size_t pos;
    bool ok = true;
    bool read(void* dst, size_t n) {
        if (pos + n > b.size()) return ok = false;
        memcpy(dst, b.data() + pos, n);
        pos += n;
        return true;
    }
    bool str(std::string* out, size_t max_len = 255) {
        out->clear();
        while (pos < b.size() && b[pos] != 0) {
            out->push_back(static_cast<char>(b[pos++]));
            if (out->size() > max_len) return ok = false;
        }
        if (pos >= b.size()) return ok = false;
        ++pos;
        return true;
    }
};

struct Channel {
    std::string name;
    int32_t type;   // 0 UINT, 1 HALF, 2 FLOAT
};

gbl_status read_exr(const char* path, float** rgba_out, int32_t* width_out, int32_t* height_out) {
    if (!path || !rgba_out || !width_out || !height_out) return fail(GBL_ERR_INVALID, "null argument");
    *rgba_out = nullptr;
    FILE* fp = fopen(path, "rb");
    if (!fp) return fail(GBL_ERR_IO, std::string("unable to read image ") + path);
    std::vector<uint8_t> file;
    {
        uint8_t chunk[65536];
        size_t got;
        while ((got = fread(chunk, 1, sizeof(chunk), fp)) > 0) file.insert(file.end(), chunk, chunk + got);
        fclose(fp);
    }
    const std::string where = std::string("unable to read image ") + path + ": ";
    Cursor c{file, 0};
    uint32_t magic = 0, version = 0;
    if (!c.read(&magic, 4) || !c.read(&version, 4) || magic != 20000630u) return fail(GBL_ERR_IO, where + "not an OpenEXR file");
    if ((version & 0xffu) != 2u) return fail(GBL_ERR_UNSUPPORTED, where + "OpenEXR file version other than 2");
    if (version & 0x200u) return fail(GBL_ERR_UNSUPPORTED, where + "tiled OpenEXR files are outside this reader (scanline only)");
    if (version & (0x800u | 0x1000u)) return fail(GBL_ERR_UNSUPPORTED, where + "deep / multi-part OpenEXR files are outside this reader");
    std::vector<Channel> channels;
    int32_t compression = -1, line_order = 0;
    int32_t dw[4] = {0, 0, -1, -1};
    bool have_dw = false;
    for (;;) {
        std::string name, type;
        if (!c.str(&name)) return fail(GBL_ERR_IO, where + "truncated header");
        if (name.empty()) break;
        int32_t size = 0;
        if (!c.str(&type) || !c.read(&size, 4) || size < 0 || c.pos + static_cast<size_t>(size) > file.size())
            return fail(GBL_ERR_IO, where + "truncated header");
        Cursor a{file, c.pos};
        c.pos += static_cast<size_t>(size);
        if (name == "channels") {
            for (;;) {
                Channel ch;
                if (!a.str(&ch.name)) return fail(GBL_ERR_IO, where + "bad channel list");
                if (ch.name.empty()) break;
                int32_t xs = 0, ys = 0;
                uint8_t reserved[4];
                if (!a.read(&ch.type, 4) || !a.read(reserved, 4) || !a.read(&xs, 4) || !a.read(&ys, 4)) return fail(GBL_ERR_IO, where + "bad channel list");
                if (xs != 1 || ys != 1) return fail(GBL_ERR_UNSUPPORTED, where + "sub-sampled channels are outside this reader");
                channels.push_back(ch);
            }
        } else if (name == "compression") {
            uint8_t v = 0;
            a.read(&v, 1);
            compression = v;
        } else if (name == "dataWindow") {
            have_dw = a.read(dw, 16);
        } else if (name == "lineOrder") {
            uint8_t v = 0;
            a.read(&v, 1);
            line_order = v;
        }
    }
    (void)line_order;   // every block carries its own y: placed by coordinate, whatever the file order
    if (channels.empty() || !have_dw || compression < 0) return fail(GBL_ERR_IO, where + "header lacks channels / dataWindow / compression");
    if (compression > 3) return fail(GBL_ERR_UNSUPPORTED, where + "compression other than NONE / RLE / ZIPS / ZIP is outside this reader");
    const int64_t width = static_cast<int64_t>(dw[2]) - dw[0] + 1, height = static_cast<int64_t>(dw[3]) - dw[1] + 1;
    if (width <= 0 || height <= 0 || width > 65536 || height > 65536) return fail(GBL_ERR_IO, where + "bad data window");
    size_t pixel_bytes = 0;
    for (const Channel& ch : channels) {
        if (ch.type == 1) pixel_bytes += 2;
        else if (ch.type == 2) pixel_bytes += 4;
        else return fail(GBL_ERR_UNSUPPORTED, where + "UINT channels are outside this reader");
    }
    const int lines_per_block = compression == 3 ? 16 : 1;
    const size_t blocks = static_cast<size_t>((height + lines_per_block - 1) / lines_per_block);
    std::vector<uint64_t> offsets(blocks);
    if (!c.read(offsets.data(), 8 * blocks)) return fail(GBL_ERR_IO, where + "truncated offset table");
    // channel planes as float
    std::vector<std::vector<float>> planes(channels.size(), std::vector<float>(static_cast<size_t>(width * height), 0.0f));
    std::vector<uint8_t> raw, tmp;
    for (size_t bi = 0; bi < blocks; ++bi) {
        Cursor d{file, static_cast<size_t>(offsets[bi])};
        int32_t y = 0, data_size = 0;
        if (offsets[bi] >= file.size() || !d.read(&y, 4) || !d.read(&data_size, 4) || data_size < 0 || d.pos + static_cast<size_t>(data_size) > file.size())
            return fail(GBL_ERR_IO, where + "bad scanline block");
        const int64_t row0 = static_cast<int64_t>(y) - dw[1];
        if (row0 < 0 || row0 >= height) return fail(GBL_ERR_IO, where + "scanline block outside the data window");
        const int64_t rows = std::min<int64_t>(lines_per_block, height - row0);
        const size_t expected = static_cast<size_t>(rows) * static_cast<size_t>(width) * pixel_bytes;
        const uint8_t* src = file.data() + d.pos;
        const uint8_t* pixels = nullptr;
        if (compression == 0 || static_cast<size_t>(data_size) == expected) {   // stored as is (also what a writer does when packing did not help)
            if (static_cast<size_t>(data_size) != expected) return fail(GBL_ERR_IO, where + "scanline block of the wrong size");
            pixels = src;
        } else {
            const bool ok = compression == 1 ? rle_decode(src, static_cast<size_t>(data_size), tmp, expected)
                                             : inflate_zlib(src, static_cast<size_t>(data_size), tmp, expected);
            if (!ok || tmp.size() != expected) return fail(GBL_ERR_IO, where + "corrupt compressed scanline block");
            unpredict(tmp, raw);
            pixels = raw.data();
        }
        const uint8_t* p = pixels;
        for (int64_t r = 0; r < rows; ++r) {
            for (size_t ci = 0; ci < channels.size(); ++ci) {
                float* dst = planes[ci].data() + static_cast<size_t>((row0 + r) * width);
                if (channels[ci].type == 1) {
                    for (int64_t x = 0; x < width; ++x) {
                        uint16_t h;
                        memcpy(&h, p + 2 * x, 2);
                        dst[x] = half_to_float(h);
                    }
                    p += 2 * width;
                } else {
                    memcpy(dst, p, static_cast<size_t>(4 * width));
                    p += 4 * width;
                }
            }
        }
    }
    // LoadEXR's RGBA assembly (tinyexr.h:11266-11420)
    int idx[4] = {-1, -1, -1, -1};
    for (size_t ci = 0; ci < channels.size(); ++ci) {
        if (channels[ci].name == "R") idx[0] = static_cast<int>(ci);
        else if (channels[ci].name == "G") idx[1] = static_cast<int>(ci);
        else if (channels[ci].name == "B") idx[2] = static_cast<int>(ci);
        else if (channels[ci].name == "A") idx[3] = static_cast<int>(ci);
    }
    const size_t n = static_cast<size_t>(width * height);
    float* out = static_cast<float*>(malloc(n * 4 * sizeof(float)));
    if (!out) return fail(GBL_ERR_OOM, where + "out of memory");
    if (channels.size() == 1) {
        for (size_t i = 0; i < n; ++i) out[4 * i] = out[4 * i + 1] = out[4 * i + 2] = out[4 * i + 3] = planes[0][i];
    } else {
        static const char* names[3] = {"R", "G", "B"};
        for (int k = 0; k < 3; ++k)
            if (idx[k] < 0) {
                free(out);
                return fail(GBL_ERR_IO, where + names[k] + " channel not found");
            }
        for (size_t i = 0; i < n; ++i) {
            out[4 * i] = planes[static_cast<size_t>(idx[0])][i];
            out[4 * i + 1] = planes[static_cast<size_t>(idx[1])][i];
            out[4 * i + 2] = planes[static_cast<size_t>(idx[2])][i];
            out[4 * i + 3] = idx[3] >= 0 ? planes[static_cast<size_t>(idx[3])][i] : 1.0f;
        }
    }
    *rgba_out = out;
    *width_out = static_cast<int32_t>(width);
    *height_out = static_cast<int32_t>(height);
    return GBL_OK;
}

}  // namespace

namespace gbl_host_detail {
// for the scene loader (image textures, image based lights): W*H float4, malloc'ed
gbl_status load_image(const std::string& path, float** rgba, int32_t* w, int32_t* h) {
    const size_t dot = path.rfind('.');
    if (dot == std::string::npos) return fail(GBL_ERR_IO, "error loading image " + path + " :unrecognized file format");
    const std::string ext = path.substr(dot);
    if (ext != ".exr" && ext != ".EXR") return fail(GBL_ERR_IO, "error loading image " + path + " :unsupported format " + ext);
    return read_exr(path.c_str(), rgba, w, h);
}
}  // namespace gbl_host_detail

extern "C" {

gbl_status gbl_host_read_image(const char* path, float** rgba_out, int32_t* width_out, int32_t* height_out) {
    return gbl_guard([&] {
        if (!path) return fail(GBL_ERR_INVALID, "null argument");
        return gbl_host_detail::load_image(path, rgba_out, width_out, height_out);
    }, [](const std::string& what) { (void)fail(GBL_ERR_INTERNAL, what); });
}

void gbl_host_free_image(float* rgba) { free(rgba); }

}  // extern "C"
