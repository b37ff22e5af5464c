// libgoblin_host.so -- MIP pyramids of image textures and image based lights.
//
// MIPMap<T>::MIPMap (GoblinTexture.cpp:40-68) restated on plain float arrays (T = float: 1 channel; T = Color: 4, of
// which resizeImage only ever touches r, g, b -- Color::operator+= leaves alpha alone, GoblinColor.h:37-42): an image
// whose sides are not powers of two is first resized up to the next ones, then every level is resizeImage of the one
// above it (:531-597: a separable gaussian of width floor(max(2, ratio)) over 2 * width taps, indices clamped at the
// border), down to 1 x 1.  Same float operations in the same order, so the texels equal the reference's.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

#include "../../../include/goblin_hip.h"

namespace gbl_host_detail {

namespace {

inline int floor_int(float f) { return static_cast<int>(floorf(f)); }
inline bool is_pow2(uint32_t n) { return (n & (n - 1)) == 0; }
inline uint32_t round_up_pow2(uint32_t n) {
    n--;
    n |= n >> 1;
    n |= n >> 2;
    n |= n >> 4;
    n |= n >> 8;
    n |= n >> 16;
    return n + 1;
}
// Goblin::log2 (GoblinUtils.h:84-87), not libm's: logf times a float 1 / ln 2 -- so the level count of a power-of-two side is
// whatever that product floors to, as in the reference
inline float goblin_log2(float n) {
    static const float inv_log2 = 1.0f / logf(2.0f);
    return logf(n) * inv_log2;
}
inline float gaussian(float x, float w, float falloff = 2.0f) { return std::max(0.0f, expf(-falloff * x * x) - expf(-falloff * w * w)); }

// resizeImage<T>, GoblinTexture.cpp:531-597
std::vector<float> resize_image(const std::vector<float>& src, int sw, int sh, int dw, int dh, int channels) {
    const float filter_width = floorf(std::max(2.0f, std::max(static_cast<float>(sw) / static_cast<float>(dw), static_cast<float>(sh) / static_cast<float>(dh))));
    const int n = floor_int(filter_width) * 2;
    std::vector<float> s_weight(static_cast<size_t>(dw) * n), t_weight(static_cast<size_t>(dh) * n);
    std::vector<int> s_index(dw), t_index(dh);
    auto taps = [&](int count, int src_size, std::vector<float>& weight, std::vector<int>& index) {
        for (int s = 0; s < count; ++s) {
            const float center = (static_cast<float>(s) + 0.5f) / count * src_size;
            index[s] = floor_int(center - filter_width + 0.5f);
            float sum = 0.0f;
            const int off = s * n;
            for (int i = 0; i < n; ++i) {
                const float p = index[s] + 0.5f + i;
                weight[off + i] = gaussian(p - center, filter_width);
                sum += weight[off + i];
            }
            const float inv = 1.0f / sum;
            for (int i = 0; i < n; ++i) weight[off + i] *= inv;
        }
    };
    taps(dw, sw, s_weight, s_index);
    taps(dh, sh, t_weight, t_index);
    const int live = channels == 4 ? 3 : channels;   // Color: rgb accumulate, alpha stays Color(0.0f).a = 1
    std::vector<float> dst(static_cast<size_t>(dw) * dh * channels, 0.0f);
    for (int t = 0; t < dh; ++t) {
        for (int s = 0; s < dw; ++s) {
            float* d = dst.data() + (static_cast<size_t>(t) * dw + s) * channels;
            if (channels == 4) d[3] = 1.0f;
            for (int i = 0; i < n; ++i) {
                const int src_t = std::min(std::max(t_index[t] + i, 0), sh - 1);
                for (int j = 0; j < n; ++j) {
                    const int src_s = std::min(std::max(s_index[s] + j, 0), sw - 1);
                    const float w = t_weight[static_cast<size_t>(t) * n + i] * s_weight[static_cast<size_t>(s) * n + j];
                    const float* p = src.data() + (static_cast<size_t>(src_t) * sw + src_s) * channels;
                    for (int c = 0; c < live; ++c) d[c] += w * p[c];
                }
            }
        }
    }
    return dst;
}

}  // namespace

// Appends the pyramid of `level0` (w x h texels of `channels` floats) to `pool` and describes it.
gbl_image build_mipmap(std::vector<float>& pool, std::vector<float> level0, int w, int h, int channels) {
    if (!is_pow2(static_cast<uint32_t>(w)) || !is_pow2(static_cast<uint32_t>(h))) {
        const int wp = static_cast<int>(round_up_pow2(static_cast<uint32_t>(w))), hp = static_cast<int>(round_up_pow2(static_cast<uint32_t>(h)));
        level0 = resize_image(level0, w, h, wp, hp, channels);
        w = wp;
        h = hp;
    }
    gbl_image img;
    img.width = static_cast<uint32_t>(w);
    img.height = static_cast<uint32_t>(h);
    img.channels = static_cast<uint32_t>(channels);
    img.levels = static_cast<uint32_t>(floor_int(std::max(goblin_log2(static_cast<float>(w)), goblin_log2(static_cast<float>(h)))) + 1);
    img.texel_offset = pool.size();
    std::vector<float> cur = std::move(level0);
    int cw = w, ch = h;
    pool.insert(pool.end(), cur.begin(), cur.end());
    for (uint32_t l = 1; l < img.levels; ++l) {
        const int nw = std::max(1, cw >> 1), nh = std::max(1, ch >> 1);
        cur = resize_image(cur, cw, ch, nw, nh, channels);
        cw = nw;
        ch = nh;
        pool.insert(pool.end(), cur.begin(), cur.end());
    }
    return img;
}

}  // namespace gbl_host_detail
