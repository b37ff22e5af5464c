// The reference's OWN sample stream, generated on the device (sample_mode = GBL_SAMPLES_STREAM).
//
// RenderTask::run (GoblinRenderer.cpp:29-52) gives every 8x8 sample tile one mt19937 (RNGImp, GoblinUtils.cpp:13-56)
// seeded with the next value of the never-seeded libc rand(), and walks the tile pixel by pixel:
// Sampler::requestSamples (GoblinSampler.cpp:108-197) draws the pixel's S = roundToSquare(spp) Sample records from it
// -- jittered strata, then the lens / per-column / in-pattern shuffles -- and every Li evaluation takes three more
// floats per evalAttenuation call (BSDFSample(rng), GoblinPathtracer.cpp:103,150,159) whose VALUES are never used
// but which move the stream.  The position of pixel p+1 in the stream therefore depends on how the paths of pixel p
// went, so a tile is inherently sequential: one workgroup owns a tile, and per pixel
//   1. emits the pixel's raw 32-bit draws (the twist runs over the 624-word state in LDS, four barrier phases),
//   2. applies the reference's shuffles as position permutations (one lane per column, columns in LDS),
//   3. assembles the S records in the reference's float layout,
//   4. traces the S paths with the replay kernel's own code, counting the discarded draws,
//   5. skips that many outputs.
// Film accumulators then equal the reference's up to float summation order, at any size, with nothing uploaded.
// Slower than the counter-based native law (the throughput mode): this is the bit-faithful one.
//
// Draw order per pixel (restated from requestSamples; F1 / F2 = total slots of the 1D / 2D patterns):
//   floats  image 2S | lens 2S | 1D columns F1 x S | 2D columns F2 x 2S            = S (4 + F1 + 2 F2)
//   uints   lens shuffle S | column shuffles (F1 + F2) x S | per sample (F1 + F2)   = S (1 + 2 F1 + 2 F2)
// The path tracer's quota (PathTracer::querySampleQuota, GoblinPathtracer.cpp:181-208) is D x {light 1D, bsdf 1D,
// pick 1D; light 2D, bsdf 2D} one-slot patterns followed by the BSSRDF block's 4 1D and 2 2D n-slot patterns.
#pragma once
#include <stdint.h>

#include "../device_scene.h"

#define GBL_MT_N 624
#define GBL_MT_M 397

#define GBL_STREAM_MAX_RUNS 16
struct StreamLayout {
    uint32_t S, root;               // samples per pixel and its root
    // the quota as runs of equally sized patterns, in request order: count r?c[i] patterns of r?n[i] slots each
    uint32_t nr1, nr2;
    uint32_t r1c[GBL_STREAM_MAX_RUNS], r1n[GBL_STREAM_MAX_RUNS], r2c[GBL_STREAM_MAX_RUNS], r2n[GBL_STREAM_MAX_RUNS];
    uint32_t F1, F2;                // total 1D / 2D slots
    uint32_t NF, NU;                // float / uint draws per pixel
    uint32_t ncols;                 // shuffled columns: lens, then every 1D slot, then every 2D slot
    uint32_t dims;
    uint32_t patterns;              // 1D + 2D patterns requested (of any slot count)
    // Which columns anybody will READ (the generator still makes every draw).  path_quota: the layout is the path tracer's --
    // per-bounce sets first, then the BSSRDF block -- and of its D sets the integrator reads used_bounces = D - 1 (the loop of
    // PathTracer::Li runs D - 1 times, GoblinPathtracer.cpp:76: the last set only moves the stream), the lens sample only under
    // a thin lens, the BSSRDF block only with subsurface materials in the scene.  0: every column (AO, Whitted).
    uint32_t path_quota, used_bounces, use_lens, use_bssrdf;
};
__host__ __device__ inline void stream_layout_finish(StreamLayout& L) {
    L.F1 = L.F2 = L.patterns = 0u;
    L.path_quota = L.used_bounces = L.use_lens = L.use_bssrdf = 0u;
    for (uint32_t i = 0; i < L.nr1; ++i) L.F1 += L.r1c[i] * L.r1n[i], L.patterns += L.r1n[i] != 0u ? L.r1c[i] : 0u;
    for (uint32_t i = 0; i < L.nr2; ++i) L.F2 += L.r2c[i] * L.r2n[i], L.patterns += L.r2n[i] != 0u ? L.r2c[i] : 0u;
    L.NF = L.S * (4u + L.F1 + 2u * L.F2);
    L.NU = L.S * (1u + 2u * L.F1 + 2u * L.F2);
    L.ncols = 1u + L.F1 + L.F2;
    L.dims = 4u + L.F1 + 2u * L.F2;
}
// path tracer: D x {light 1D, bsdf 1D, pick 1D; light 2D, bsdf 2D} + the BSSRDF block (4 1D and 2 2D patterns of nb / nb2
// slots); AO (ao != 0): one 2D pattern of `ao` directions (AORenderer::querySampleQuota, GoblinAO.cpp:39-42)
__host__ __device__ inline StreamLayout stream_layout(int spp, int root, int max_depth, int nb, int nb2, int ao = 0) {
    StreamLayout L;
    L.S = static_cast<uint32_t>(spp);
    L.root = static_cast<uint32_t>(root);
    if (ao != 0) {
        L.nr1 = 0u;
        L.nr2 = 1u;
        L.r2c[0] = 1u;
        L.r2n[0] = static_cast<uint32_t>(ao);
    } else {
        L.nr1 = 2u;
        L.r1c[0] = 3u * static_cast<uint32_t>(max_depth); L.r1n[0] = 1u;
        L.r1c[1] = 4u; L.r1n[1] = static_cast<uint32_t>(nb);
        L.nr2 = 2u;
        L.r2c[0] = 2u * static_cast<uint32_t>(max_depth); L.r2n[0] = 1u;
        L.r2c[1] = 2u; L.r2n[1] = static_cast<uint32_t>(nb2);
    }
    stream_layout_finish(L);
    return L;
}
// Whitted renderer (WhittedRenderer::querySampleQuota, GoblinWhitted.cpp:46-70): per light {light 1D, bsdf 1D} and
// {light 2D, bsdf 2D} of that light's n slots, the pick 1D, the BSSRDF block.  light_n[i] = DevLight::wh_n.
// Returns false when the lights do not fit GBL_STREAM_MAX_RUNS.
template <class GetN>
__host__ __device__ inline bool stream_layout_whitted(StreamLayout& L, int spp, int root, int nb, int nb2, int num_lights, GetN light_n) {
    L.S = static_cast<uint32_t>(spp);
    L.root = static_cast<uint32_t>(root);
    if (num_lights + 2 > GBL_STREAM_MAX_RUNS) return false;
    L.nr1 = L.nr2 = 0u;
    for (int i = 0; i < num_lights; ++i) {
        L.r1c[L.nr1] = 2u; L.r1n[L.nr1++] = light_n(i);
        L.r2c[L.nr2] = 2u; L.r2n[L.nr2++] = light_n(i);
    }
    L.r1c[L.nr1] = 1u; L.r1n[L.nr1++] = 1u;
    L.r1c[L.nr1] = 4u; L.r1n[L.nr1++] = static_cast<uint32_t>(nb);
    L.r2c[L.nr2] = 2u; L.r2n[L.nr2++] = static_cast<uint32_t>(nb2);
    stream_layout_finish(L);
    return true;
}
// What a shuffled column needs when its elements are placed into the records, made once per workgroup (stream_columns): looking
// the column up in the layout's runs per pixel walked StreamLayout's arrays in scratch memory, 40 % of the assembly phase.
struct DevStreamCol {
    uint32_t two_d;   // 0: 1D slot, 1: 2D slot (column 0, the lens sample, is a one-slot 2D pattern)
    uint32_t raw;     // where the column's floats start among the pixel's raw draws
    uint32_t rec;     // the slot's float offset in a record
    float a, b;       // the stratum's corner: j * strata | ux * strata, uy * strata (stratifiedUniform1D / 2D, GoblinSampler.cpp:276-307)
    float sub;        // the sub-cell's width
    uint32_t col;     // the column's number (the table only lists the columns that are read)
    uint32_t pad;
};
// words of global scratch one workgroup needs: raw draws, the column table, records
// tail_per_sample: most outputs one sample can take after its record (the integrator's discarded draws + the medium's)
__host__ __device__ inline uint64_t stream_scratch_words(const StreamLayout& L, uint32_t tail_per_sample = 0) {
    return static_cast<uint64_t>(L.NF) + L.NU + static_cast<uint64_t>(L.ncols) * (sizeof(DevStreamCol) / 4) + static_cast<uint64_t>(L.S) * L.dims +
           static_cast<uint64_t>(L.S) * (3u + tail_per_sample);
}

#ifdef __HIPCC__
struct StreamCtx {
    uint32_t* mt;       // LDS: two blocks of GBL_MT_N state words
    uint32_t pos;       // next unread word of the current block (GBL_MT_N = exhausted) ...
    uint32_t which;     // ... and which block that is: the same in every thread of the workgroup, so the cursor needs no LDS word
                        // and advancing it no barrier
    uint32_t* lperm;    // LDS scratch for the shuffles (the traversal stacks' region, idle while samples are generated)
    uint32_t lperm_words;
    uint32_t* raw;      // global, this workgroup's: the pixel's NF + NU raw draws
    DevStreamCol* cols; // global: descriptors of the ncols_used columns somebody reads, in column order (stream_columns)
    uint32_t ncols_used;
    float* recs;        // global: S x dims floats
};

// std::mt19937(seed): the Knuth initialiser, serial over the state.  The state lives twice in LDS (c.mt = two blocks of
// GBL_MT_N words): a refresh writes the other block, so no word is overwritten while it is still being read -- a phase needs
// one barrier, not two, and reading outputs from the current block needs none.
__device__ __forceinline__ void mt_seed(StreamCtx& c, uint32_t seed) {
    __syncthreads();   // (nobody still reads the blocks)
    if (threadIdx.x == 0) {
        uint32_t x = seed;
        c.mt[0] = x;
        for (uint32_t i = 1; i < GBL_MT_N; ++i) {
            x = 1812433253u * (x ^ (x >> 30)) + i;
            c.mt[i] = x;
        }
    }
    c.pos = GBL_MT_N;
    c.which = 0u;
    __syncthreads();
}
__device__ __forceinline__ uint32_t mt_mix(uint32_t a, uint32_t b, uint32_t far) {
    uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
    return far ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}
__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}
// A workgroup barrier for phases that only exchange LDS words: __syncthreads() also waits for every global store in flight.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
// the generator's state to / from global memory (GBL_MT_N + 1 words; the medium phase rewinds the stream).  Whole workgroup.
__device__ __forceinline__ void mt_save(const StreamCtx& c, uint32_t* dst) {
    const uint32_t* cur = c.mt + c.which * GBL_MT_N;
    for (uint32_t t = threadIdx.x; t < GBL_MT_N; t += blockDim.x) dst[t] = cur[t];
    if (threadIdx.x == 0) dst[GBL_MT_N] = c.pos;
    __syncthreads();
}
__device__ __forceinline__ void mt_restore(StreamCtx& c, const uint32_t* src) {
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < GBL_MT_N; t += blockDim.x) c.mt[t] = src[t];
    c.pos = src[GBL_MT_N];
    c.which = 0u;
    __syncthreads();
}
// One state refresh into the other block.  Word i of the new block needs old words i, i + 1 and word i + M of whichever block
// is current there: [0, N-M) reads old words only, [N-M, 2(N-M)) the first range's new words, the rest the second's -- three
// phases of <= 227 independent words, the wrap-around word in the third.  (Tempering and storing the outputs inside these phases
// was measured: the refresh's share of an instrumented frame 45 -> 79 ms.)
__device__ __forceinline__ void mt_twist(StreamCtx& c) {
    constexpr uint32_t R = GBL_MT_N - GBL_MT_M;   // 227
    const uint32_t t = threadIdx.x;
    const uint32_t* old = c.mt + c.which * GBL_MT_N;
    uint32_t* nw = c.mt + (c.which ^ 1u) * GBL_MT_N;
    if (t < R) nw[t] = mt_mix(old[t], old[t + 1], old[t + GBL_MT_M]);
    lds_barrier();
    if (t < R) nw[R + t] = mt_mix(old[R + t], old[R + t + 1], nw[t]);
    lds_barrier();
    if (t < GBL_MT_N - 1 - 2 * R) nw[2 * R + t] = mt_mix(old[2 * R + t], old[2 * R + t + 1], nw[R + t]);
    else if (t == GBL_BLOCK - 1) nw[GBL_MT_N - 1] = mt_mix(old[GBL_MT_N - 1], nw[0], nw[GBL_MT_M - 1]);
    lds_barrier();
    c.pos = 0u;
    c.which ^= 1u;
}
// the next `count` outputs of the tile's generator, written to dst (null: skipped).  Called by the whole workgroup.
// (The block a refresh writes was last read three barriers earlier -- by the refresh before it -- so the outputs are read
//  without one.)
__device__ __forceinline__ void stream_emit(StreamCtx& c, uint32_t* dst, uint32_t count) {
    uint32_t done = 0;
    while (done < count) {
        if (c.pos == GBL_MT_N) mt_twist(c);
        const uint32_t* cur = c.mt + c.which * GBL_MT_N;
        const uint32_t n = min(GBL_MT_N - c.pos, count - done);
        if (dst)
            for (uint32_t t = threadIdx.x; t < n; t += blockDim.x) dst[done + t] = mt_temper(cur[c.pos + t]);
        c.pos += n;
        done += n;
    }
    if (dst) __syncthreads();   // what was stored is visible to the workgroup from here on
}

// the next `count` outputs reduced modulo S (<= 256) and left in LDS as bytes, output i at ld[(i % S) * B + i / S]: the column
// shuffles' draws, which only ever matter modulo S, where the lanes that apply them find them (stream_generate_pixel).
// Called by the whole workgroup.
__device__ __forceinline__ void stream_emit_mod8(StreamCtx& c, unsigned char* ld, uint32_t count, uint32_t S, uint32_t B) {
    const uint32_t pow2 = (S & (S - 1u)) == 0u ? S - 1u : 0u, shift = 31u - static_cast<uint32_t>(__clz(static_cast<int>(S)));
    uint32_t done = 0;
    while (done < count) {
        if (c.pos == GBL_MT_N) mt_twist(c);
        const uint32_t* cur = c.mt + c.which * GBL_MT_N;
        const uint32_t n = min(GBL_MT_N - c.pos, count - done);
        for (uint32_t t = threadIdx.x; t < n; t += blockDim.x) {
            const uint32_t i = done + t, x = mt_temper(cur[c.pos + t]);
            const uint32_t col = pow2 ? i >> shift : i / S, k = pow2 ? (i & pow2) : i - col * S;
            ld[k * B + col] = static_cast<unsigned char>(pow2 ? (x & pow2) : x % S);
        }
        c.pos += n;
        done += n;
    }
    __syncthreads();
}

// RNGImp::randomFloat: uniform_real_distribution<float>(0, 1) over one 32-bit draw -- generate_canonical<float, 24>
// divides the draw (rounded to float) by 2^32 and steps a result of 1.0 down to the float below it
__device__ __forceinline__ float stream_u01(uint32_t x) {
    float r = static_cast<float>(x) / 4294967296.0f;
    return r >= 1.0f ? 0.99999994f : r;
}

// stratifiedUniform2D element: stratum `slot` of an n-point pattern, sub-cell p of the root x root grid (GoblinSampler.cpp:288-307)
__device__ __forceinline__ void stream_strat2(const StreamLayout& L, uint32_t n, uint32_t slot, uint32_t p, float f0, float f1, float* x, float* y) {
    int r = static_cast<int>(sqrtf(static_cast<float>(n)));
    float strata = 1.0f / r;
    float sub = strata / static_cast<int>(L.root);
    int ux = static_cast<int>(slot) % r, uy = static_cast<int>(slot) / r;
    int px = static_cast<int>(p % L.root), py = static_cast<int>(p / L.root);
    float xo = px + f0, yo = py + f1;
    *x = ux * strata + xo * sub;
    *y = uy * strata + yo * sub;
}

// Which pattern a shuffled column belongs to: column 0 is the lens sample, columns 1 .. F1 the 1D slots, the rest the 2D slots.
struct StreamCol {
    uint32_t two_d;   // 0: 1D slot, 1: 2D slot
    uint32_t slot;    // running 1D / 2D slot number (the record's float offset follows from it)
    uint32_t n, j;    // the pattern's slot count and this slot's number inside it
};
__device__ __forceinline__ StreamCol stream_column(const StreamLayout& L, uint32_t col) {   // col >= 1
    StreamCol c;
    uint32_t s = col - 1u;
    c.two_d = s >= L.F1 ? 1u : 0u;
    if (c.two_d) s -= L.F1;
    c.slot = s;
    c.n = 1u;
    c.j = 0u;
    const uint32_t nr = c.two_d ? L.nr2 : L.nr1;
    for (uint32_t r = 0; r < nr; ++r) {
        const uint32_t cnt = c.two_d ? L.r2c[r] : L.r1c[r], n = c.two_d ? L.r2n[r] : L.r1n[r];
        if (s < cnt * n) {
            c.n = n;
            c.j = s % n;
            break;
        }
        s -= cnt * n;
    }
    return c;
}

// Does anybody read column `col` of a pixel's records?  (StreamLayout::path_quota)
__device__ __forceinline__ bool stream_col_needed(const StreamLayout& L, uint32_t col) {
    if (L.path_quota == 0u) return true;
    if (col == 0u) return L.use_lens != 0u;
    uint32_t s = col - 1u;
    if (s < L.F1) return s < L.r1c[0] ? (s / 3u) < L.used_bounces : L.use_bssrdf != 0u;   // {light, bsdf, pick} per bounce, then the block
    s -= L.F1;
    return s < L.r2c[0] ? (s / 2u) < L.used_bounces : L.use_bssrdf != 0u;                  // {light, bsdf} per bounce, then the block
}

// The workgroup's column table, once per launch: the columns that are read, in column order.  Called by the whole workgroup.
__device__ __forceinline__ void stream_columns(StreamCtx& c, const StreamLayout& L) {
    const uint32_t S = L.S;
    uint32_t used = 0;
    for (uint32_t col = 0; col < L.ncols; ++col) used += stream_col_needed(L, col) ? 1u : 0u;
    c.ncols_used = used;
    for (uint32_t col = threadIdx.x; col < L.ncols; col += blockDim.x) {
        if (!stream_col_needed(L, col)) continue;
        uint32_t at = 0;
        for (uint32_t q = 0; q < col; ++q) at += stream_col_needed(L, q) ? 1u : 0u;
        DevStreamCol d;
        d.col = col;
        d.pad = 0u;
        if (col == 0u) {   // the lens sample: stream_strat2(L, 1, 0, ...)
            d.two_d = 1u;
            d.raw = 2 * S;
            d.rec = 2u;
            d.a = d.b = 0.0f;   // r = 1, strata = 1: ux * strata = uy * strata = 0
            d.sub = 1.0f / static_cast<int>(L.root);
        } else {
            const StreamCol sc = stream_column(L, col);
            d.two_d = sc.two_d;
            if (sc.two_d == 0u) {
                const float strata = 1.0f / static_cast<float>(sc.n);
                d.raw = 4 * S + sc.slot * S;
                d.rec = 4u + sc.slot;
                d.a = sc.j * strata;
                d.b = 0.0f;
                d.sub = strata / static_cast<int>(S);
            } else {
                const int r = static_cast<int>(sqrtf(static_cast<float>(sc.n)));
                const float strata = 1.0f / r;
                const int ux = static_cast<int>(sc.j) % r, uy = static_cast<int>(sc.j) / r;
                d.raw = 4 * S + L.F1 * S + 2 * sc.slot * S;
                d.rec = 4u + L.F1 + 2 * sc.slot;
                d.a = ux * strata;
                d.b = uy * strata;
                d.sub = strata / static_cast<int>(L.root);
            }
        }
        c.cols[at] = d;
    }
    __syncthreads();
}

// Sampler::requestSamples for the pixel (cx, cy): fills c.recs.  Called by the whole workgroup.
// The shuffles run one lane per column on 16-bit positions in LDS (the idle traversal stacks' region), a round of columns at a
// time, and every round's columns go straight from there into the records: the permutations never travel through global
// memory (66 KB written and read back per pixel at configs[1] until round 3) and configs[1]'s 65 columns are one round.
// tm (instrumented builds): wall_clock64 ticks spent emitting / permuting / assembling, accumulated by thread 0
__device__ __forceinline__ void stream_generate_pixel(StreamCtx& c, const StreamLayout& L, int cx, int cy,
                                                      unsigned long long* tm = nullptr) {
    const uint32_t S = L.S;
    unsigned long long t0 = tm ? wall_clock64() : 0ull;
    // (raising the wave priority for this phase, the part only this workgroup can do, over the other workgroups' traversal loops:
    //  no change, 179.8 against 179.0 ms)
    // the in-pattern shuffles' draws (the last S (F1 + F2) uints) only matter to patterns of more than one slot: a one-slot
    // pattern swaps its slot with itself.  Without such patterns (the path tracer's quota without a BSSRDF block) they are passed over.
    const uint32_t n_shuffled = L.NF + L.ncols * S;
    const bool in_pattern = L.F1 + L.F2 != L.patterns;
    unsigned short* lp = reinterpret_cast<unsigned short*>(c.lperm);
    // When every column's positions (16 bits) AND its shuffle draws modulo S (a byte: S <= 256) fit the LDS region together, the
    // draws never go through global memory: they are left where the shuffling lanes read them (configs[1]: 31.5 KB of the 40).
    const bool ld_draws = S <= 256u && L.ncols <= static_cast<uint32_t>(GBL_BLOCK) && 3u * S * L.ncols <= 4u * c.lperm_words;
    const uint32_t B = ld_draws ? L.ncols : min(static_cast<uint32_t>(GBL_BLOCK), (2u * c.lperm_words) / S);   // columns per round
    unsigned char* ld = reinterpret_cast<unsigned char*>(lp + S * B);
    if (ld_draws) {
        stream_emit(c, c.raw, L.NF);
        stream_emit_mod8(c, ld, L.ncols * S, S, B);
        if (in_pattern) stream_emit(c, c.raw + n_shuffled, L.NF + L.NU - n_shuffled);
    } else {
        stream_emit(c, c.raw, in_pattern ? L.NF + L.NU : n_shuffled);
    }
    if (!in_pattern) stream_emit(c, nullptr, L.NF + L.NU - n_shuffled);
    if (tm) {
        const unsigned long long t1 = wall_clock64();
        tm[0] += t1 - t0;
        t0 = t1;
    }
    // image samples: not shuffled (sample k sits in sub-cell k, GoblinSampler.cpp:130-131)
    for (uint32_t k = threadIdx.x; k < S; k += blockDim.x) {
        float* rec = c.recs + static_cast<size_t>(k) * L.dims;
        float x, y;
        stream_strat2(L, 1u, 0u, k, stream_u01(c.raw[2 * k]), stream_u01(c.raw[2 * k + 1]), &x, &y);
        rec[0] = cx + x;
        rec[1] = cy + y;
    }
    for (uint32_t c0 = 0; c0 < L.ncols; c0 += B) {
        // ---- shuffle<T>(buffer, S, dim, rng): for n in [0, S): swap(element n, element rng.randomUInt() % S)
        // (GoblinSampler.h:149-157), tracked as the position permutation of each column
        const uint32_t b = threadIdx.x, col = c0 + b;
        const uint32_t nb = min(B, L.ncols - c0);
        const bool col_read = b < nb && stream_col_needed(L, col);   // (a column nobody reads needs no permutation)
        if (col_read && ld_draws) {
            for (uint32_t k = 0; k < S; ++k) lp[k * B + b] = static_cast<unsigned short>(k);
            for (uint32_t n = 0; n < S; ++n) {
                const uint32_t ia = n * B + b, ib = static_cast<uint32_t>(ld[n * B + b]) * B + b;
                const unsigned short va = lp[ia], vb = lp[ib];
                lp[ia] = vb;
                lp[ib] = va;
            }
        } else if (col_read) {
            for (uint32_t k = 0; k < S; ++k) lp[k * B + b] = static_cast<unsigned short>(k);
            const uint32_t* u = c.raw + L.NF + col * S;
            const uint32_t pow2 = (S & (S - 1u)) == 0u ? S - 1u : 0u;   // x % S without the division where S is a power of two
            uint32_t o[16], nx[16];
#pragma unroll
            for (uint32_t i = 0; i < 16; ++i) nx[i] = i < S ? u[i] : 0u;
            for (uint32_t n0 = 0; n0 < S; n0 += 16) {
#pragma unroll
                for (uint32_t i = 0; i < 16; ++i) o[i] = nx[i];
#pragma unroll
                for (uint32_t i = 0; i < 16; ++i) nx[i] = n0 + 16 + i < S ? u[n0 + 16 + i] : 0u;   // (the next sixteen draws travel while these are applied)
#pragma unroll
                for (uint32_t i = 0; i < 16; ++i) {
                    if (n0 + i >= S) break;
                    const uint32_t other = pow2 ? (o[i] & pow2) : o[i] % S;
                    const uint32_t ia = (n0 + i) * B + b, ib = other * B + b;
                    const unsigned short va = lp[ia], vb = lp[ib];
                    lp[ia] = vb;
                    lp[ib] = va;
                }
            }
        }
        __syncthreads();
        if (tm) {
            const unsigned long long t1 = wall_clock64();
            tm[1] += t1 - t0;
            t0 = t1;
        }
        // ---- records: sample k takes, in every column of the round, the element its position's permutation points at
        // (eight columns at a time: their draws are fetched together, then placed)
        const uint32_t root_pow2 = (L.root & (L.root - 1u)) == 0u ? L.root - 1u : 0u, root_shift = 31u - static_cast<uint32_t>(__clz(static_cast<int>(L.root)));
        // the table's entries whose columns belong to this round: [u0, u1)
        uint32_t u0 = 0, u1 = c.ncols_used;
        if (nb != L.ncols) {
            u0 = u1 = 0;
            for (uint32_t q = 0; q < c.ncols_used; ++q) {
                const uint32_t qc = c.cols[q].col;
                u0 += qc < c0 ? 1u : 0u;
                u1 += qc < c0 + nb ? 1u : 0u;
            }
        }
        for (uint32_t k = threadIdx.x; k < S; k += blockDim.x) {
            float* rec = c.recs + static_cast<size_t>(k) * L.dims;
            for (uint32_t b0 = u0; b0 < u1; b0 += 8) {
                uint32_t pp[8];
                DevStreamCol dc[8];
                float f0[8], f1[8];
#pragma unroll
                for (uint32_t i = 0; i < 8; ++i) {
                    const uint32_t bb = min(b0 + i, u1 - 1u);
                    dc[i] = c.cols[bb];
                    pp[i] = lp[k * B + (dc[i].col - c0)];
                }
#pragma unroll
                for (uint32_t i = 0; i < 8; ++i) {
                    const uint32_t e = dc[i].raw + (dc[i].two_d ? 2 * pp[i] : pp[i]);
                    f0[i] = stream_u01(c.raw[e]);
                    f1[i] = dc[i].two_d ? stream_u01(c.raw[e + 1]) : 0.0f;
                }
#pragma unroll
                for (uint32_t i = 0; i < 8; ++i) {
                    const uint32_t p = pp[i];
                    if (b0 + i >= u1) {
                    } else if (dc[i].two_d == 0u) {
                        const float off = static_cast<int>(p) + f0[i];
                        rec[dc[i].rec] = dc[i].a + off * dc[i].sub;   // stratifiedUniform1D, GoblinSampler.cpp:276-286
                    } else {   // stratifiedUniform2D, :288-307: sub-cell p of the root x root grid
                        const int px = static_cast<int>(root_pow2 ? (p & root_pow2) : p % L.root), py = static_cast<int>(root_pow2 ? (p >> root_shift) : p / L.root);
                        const float xo = px + f0[i], yo = py + f1[i];
                        rec[dc[i].rec] = dc[i].a + xo * dc[i].sub;
                        rec[dc[i].rec + 1] = dc[i].b + yo * dc[i].sub;
                    }
                }
            }
        }
        __syncthreads();   // the round's positions have been read before the next round overwrites them
        if (tm) {
            const unsigned long long t1 = wall_clock64();
            tm[2] += t1 - t0;
            t0 = t1;
        }
    }
    // ---- per-sample shuffles inside each pattern (:185-196); a one-slot pattern swaps its slot with itself
    const uint32_t* uper = c.raw + L.NF + L.ncols * S;   // in-pattern shuffle draws, F1 + F2 per sample
    for (uint32_t k = threadIdx.x; in_pattern && k < S; k += blockDim.x) {
        float* rec = c.recs + static_cast<size_t>(k) * L.dims;
        const uint32_t* us = uper + static_cast<size_t>(k) * (L.F1 + L.F2);
        uint32_t off1 = 0;
        for (uint32_t r = 0; r < L.nr1; ++r) {
            const uint32_t n = L.r1n[r];
            for (uint32_t i = 0; i < L.r1c[r]; ++i, off1 += n) {
                if (n <= 1u) continue;
                float* pat = rec + 4 + off1;
                const uint32_t* up = us + off1;
                for (uint32_t m = 0; m < n; ++m) {
                    const uint32_t other = up[m] % n;
                    const float tmp = pat[m];
                    pat[m] = pat[other];
                    pat[other] = tmp;
                }
            }
        }
        uint32_t off2 = 0;
        for (uint32_t r = 0; r < L.nr2; ++r) {
            const uint32_t n = L.r2n[r];
            for (uint32_t i = 0; i < L.r2c[r]; ++i, off2 += n) {
                if (n <= 1u) continue;
                float* pat = rec + 4 + L.F1 + 2 * off2;
                const uint32_t* up = us + L.F1 + off2;
                for (uint32_t m = 0; m < n; ++m) {
                    const uint32_t other = up[m] % n;
                    const float t0_ = pat[2 * m], t1_ = pat[2 * m + 1];
                    pat[2 * m] = pat[2 * other];
                    pat[2 * m + 1] = pat[2 * other + 1];
                    pat[2 * other] = t0_;
                    pat[2 * other + 1] = t1_;
                }
            }
        }
    }
    __syncthreads();
    if (tm) tm[2] += wall_clock64() - t0;
}
#endif
