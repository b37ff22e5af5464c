// libgoblin_hip.so, kernel unit: the first-hit passes (subsurface term, participating medium), the film resolve, the
// device BLAS build (kernels/lbvh.h) and the device self tests of the C ABI.
#include "abi_guard.h"
#include <chrono>
#include "gbl_internal.h"
#include "kernels/render_kernels.h"
#include "kernels/subsurface.h"
#include "kernels/volume.h"
#include "kernels/lbvh.h"
#include <hipcub/hipcub.hpp>

#include <cstring>

// (the first-hit passes are not instrumented: gbl_stats counts the integrator kernels' queries)
gbl_li_kernel gbl_kernel_sss(bool replay) { return replay ? sss_kernel<true, false> : sss_kernel<false, false>; }

gbl_render_kernel gbl_kernel_vol(bool replay) { return replay ? vol_kernel<true, false> : vol_kernel<false, false>; }

// li[i] = 1 * (tr[i] * li[i] + Lv[i]) over the call's camera samples: the caller's li_out, after the splat has read it
__global__ void vol_combine_kernel(float4* li, const float4* vol, uint64_t n) {
    const uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 L = li[i], tr = vol[2 * i], lv = vol[2 * i + 1];
    li[i] = make_float4(1.0f * (tr.x * L.x + lv.x), 1.0f * (tr.y * L.y + lv.y), 1.0f * (tr.z * L.z + lv.z), L.w);
}
// device-built trees: the triangle bounds (uploaded per original triangle) in the order the build left the triangles in
__global__ void tri_bounds_gather_kernel(const DevTri* tris, const DevTriBound* by_id, DevTriBound* out, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = by_id[tris[i].shade];
}
void gbl_launch_tri_bounds_gather(const DevTri* tris, const DevTriBound* by_id, DevTriBound* out, uint32_t n) {
    if (n) hipLaunchKernelGGL(tri_bounds_gather_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, tris, by_id, out, n);
}
void gbl_launch_vol_combine(float4* li, const float4* vol, uint64_t n, hipStream_t stream) {
    hipLaunchKernelGGL(vol_combine_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, stream, li, vol, n);
}

// Film::writeImage's normalise step on the device: rgb = color / weight
__global__ void film_resolve_kernel(const float* accum, float* rgb, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 a = reinterpret_cast<const float4*>(accum)[i];
    float inv = 1.0f / a.w;
    rgb[3 * i + 0] = a.x * inv;
    rgb[3 * i + 1] = a.y * inv;
    rgb[3 * i + 2] = a.z * inv;
}
void gbl_launch_film_resolve(const float* accum, float* rgb, int n, hipStream_t stream) {
    hipLaunchKernelGGL(film_resolve_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, accum, rgb, n);
}

// ---------------------------------------------------------------------------
// Device BLAS build (kernels/lbvh.h).  One mesh at a time: nodes are written at node_base (absolute child
// references), DevTri records at tri_base in Morton order.  Returns the mesh's root reference.
// ---------------------------------------------------------------------------
namespace {
struct LbvhScratch {
    std::vector<void*> ptrs;
    ~LbvhScratch() {
        for (void* p : ptrs) (void)hipFree(p);
    }
    template <class T>
    bool alloc(T** out, size_t n) {
        void* p = nullptr;
        if (hipMalloc(&p, std::max<size_t>(1, n) * sizeof(T)) != hipSuccess) return false;
        ptrs.push_back(p);
        *out = static_cast<T*>(p);
        return true;
    }
};
}  // namespace

gbl_status gbl_build_blas_device(gbl_ctx* ctx, const float* d_pos, const uint32_t* d_idx, uint32_t n, const float* lo, const float* hi,
                             DevNode* d_nodes, int32_t node_base, DevTri* d_tris, uint32_t tri_base, uint32_t shade_base, uint32_t tri_flags,
                             int32_t* root_out, uint32_t* nodes_out, int* depth_out) {
    LbvhScratch sc;
    unsigned long long *keys = nullptr, *keys_sorted = nullptr;
    LbvhBox* tri_box = nullptr;
    LbvhTree t;
    memset(&t, 0, sizeof(t));
    LbvhFrontier *fa = nullptr, *fb = nullptr;
    uint32_t* counters = nullptr;   // [0] next frontier size, [1] nodes emitted
    const size_t ni = n > 1 ? n - 1 : 1;
    if (!sc.alloc(&keys, n) || !sc.alloc(&keys_sorted, n) || !sc.alloc(&tri_box, n) || !sc.alloc(&t.left, ni) || !sc.alloc(&t.right, ni) ||
        !sc.alloc(&t.parent, 2 * static_cast<size_t>(n)) || !sc.alloc(&t.first, ni) || !sc.alloc(&t.last, ni) || !sc.alloc(&t.box, ni) ||
        !sc.alloc(&t.leaf_box, n) || !sc.alloc(&t.visits, ni) || !sc.alloc(&fa, ni) || !sc.alloc(&fb, ni) || !sc.alloc(&counters, 2)) {
        ctx->error = "hipMalloc(device BVH build scratch) failed";
        return GBL_ERR_OOM;
    }
    LbvhBox mesh;
    for (int a = 0; a < 3; ++a) {
        mesh.lo[a] = lo[a];
        mesh.hi[a] = hi[a];
    }
    const dim3 block(256), grid((n + 255) / 256);
    hipLaunchKernelGGL(lbvh_keys, grid, block, 0, 0, d_pos, d_idx, n, mesh, keys, tri_box);
    size_t temp_bytes = 0;
    HIP_TRY(ctx, hipcub::DeviceRadixSort::SortKeys(nullptr, temp_bytes, keys, keys_sorted, static_cast<int>(n), 0, 62));
    unsigned char* temp = nullptr;
    if (!sc.alloc(&temp, temp_bytes)) {
        ctx->error = "hipMalloc(radix sort scratch) failed";
        return GBL_ERR_OOM;
    }
    HIP_TRY(ctx, hipcub::DeviceRadixSort::SortKeys(temp, temp_bytes, keys, keys_sorted, static_cast<int>(n), 0, 62));
    hipLaunchKernelGGL(lbvh_tris, grid, block, 0, 0, d_pos, d_idx, keys_sorted, n, shade_base, tri_flags, d_tris + tri_base);
    *nodes_out = 0;
    *depth_out = 0;
    if (n <= GBL_MAX_LEAF_TRIS) {   // the whole mesh is one leaf
        *root_out = ~static_cast<int32_t>((tri_base << 2) | (n - 1u));
        HIP_TRY(ctx, hipDeviceSynchronize());
        return GBL_OK;
    }
    hipLaunchKernelGGL(lbvh_gather_boxes, grid, block, 0, 0, keys_sorted, tri_box, n, t.leaf_box);
    hipLaunchKernelGGL(lbvh_hierarchy, grid, block, 0, 0, keys_sorted, static_cast<int>(n), t);
    HIP_TRY(ctx, hipMemsetAsync(t.visits, 0, ni * sizeof(uint32_t), 0));
    hipLaunchKernelGGL(lbvh_fit, grid, block, 0, 0, static_cast<int>(n), t);
    // collapse, one 4-wide level per launch
    LbvhFrontier rootf = {0, 0};
    uint32_t init[2] = {0u, 1u};
    HIP_TRY(ctx, hipMemcpy(fa, &rootf, sizeof(rootf), hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(counters, init, sizeof(init), hipMemcpyHostToDevice));
    uint32_t n_in = 1;
    int depth = 0;
    while (n_in > 0) {
        ++depth;
        hipLaunchKernelGGL(lbvh_collapse, dim3((n_in + 255) / 256), block, 0, 0, t, fa, n_in, fb, counters, counters + 1, d_nodes + node_base,
                           node_base, tri_base);
        uint32_t h[2];
        HIP_TRY(ctx, hipMemcpy(h, counters, sizeof(h), hipMemcpyDeviceToHost));
        n_in = h[0];
        *nodes_out = h[1];
        HIP_TRY(ctx, hipMemsetAsync(counters, 0, sizeof(uint32_t), 0));
        std::swap(fa, fb);
        if (depth > 128) {
            ctx->error = "device BVH build did not terminate";
            return GBL_ERR_DEVICE;
        }
    }
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipDeviceSynchronize());
    *root_out = node_base;
    *depth_out = depth;
    return GBL_OK;
}

// ---------------------------------------------------------------------------
// VALU issue-rate microbenchmark (no memory traffic): every wave runs `iters` x 64 instructions of ONE kind over 16
// independent register chains, with w = blockDim.x / 256 waves resident on each SIMD of each CU (one workgroup per CU:
// the launch asks for more than half of a CU's LDS).  What the roofline of the traversal kernels is priced against
// (bench.py): the measured cost of a wave64 instruction in SIMD cycles, alone on its SIMD and beside 1-3 other waves.
// ---------------------------------------------------------------------------
typedef float gbl_f2 __attribute__((ext_vector_type(2)));
#define GBL_VI_16(STMT) STMT(0) STMT(1) STMT(2) STMT(3) STMT(4) STMT(5) STMT(6) STMT(7) STMT(8) STMT(9) STMT(10) STMT(11) STMT(12) STMT(13) STMT(14) STMT(15)
template <int OP>
__device__ __forceinline__ void valu_issue_loop(float (&a)[16], gbl_f2 (&p)[16], uint32_t (&u)[16], float b, float c, gbl_f2 pb, gbl_f2 pc, uint32_t ub, uint32_t iters);
__global__ __launch_bounds__(1024) void valu_issue_kernel(float* out, int op, uint32_t iters, unsigned long long* ticks) {
    extern __shared__ __align__(16) unsigned char vi_smem[];
    float a[16];
    gbl_f2 p[16];
    uint32_t u[16];
    const float b = 1.0f + 1e-7f * threadIdx.x, c = 1e-9f * blockIdx.x;
    const gbl_f2 pb = {b, b}, pc = {c, c};
    const uint32_t ub = 0x03020100u + threadIdx.x;
    for (int i = 0; i < 16; ++i) {
        a[i] = 0.001f * (i + 1);
        p[i] = gbl_f2{a[i], a[i]};
        u[i] = threadIdx.x * 16u + i;
    }
    asm volatile("s_mov_b32 vcc_lo, 0x55555555\n\ts_mov_b32 vcc_hi, 0x55555555\n\ts_mov_b32 s10, 0x33333333\n\ts_mov_b32 s11, 0x33333333" ::: "vcc", "s10", "s11");   // (v_cndmask_b32 reads it)
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    switch (op) {   // (wave-uniform: one scalar branch in front of the measured loop)
        case 0: valu_issue_loop<0>(a, p, u, b, c, pb, pc, ub, iters); break;
        case 1: valu_issue_loop<1>(a, p, u, b, c, pb, pc, ub, iters); break;
        case 2: valu_issue_loop<2>(a, p, u, b, c, pb, pc, ub, iters); break;
        case 3: valu_issue_loop<3>(a, p, u, b, c, pb, pc, ub, iters); break;
        case 4: valu_issue_loop<4>(a, p, u, b, c, pb, pc, ub, iters); break;
        case 5: valu_issue_loop<5>(a, p, u, b, c, pb, pc, ub, iters); break;
        case 6: valu_issue_loop<6>(a, p, u, b, c, pb, pc, ub, iters); break;
        case 7: valu_issue_loop<7>(a, p, u, b, c, pb, pc, ub, iters); break;
        case 8: valu_issue_loop<8>(a, p, u, b, c, pb, pc, ub, iters); break;
        case 9: valu_issue_loop<9>(a, p, u, b, c, pb, pc, ub, iters); break;
        case 10: valu_issue_loop<10>(a, p, u, b, c, pb, pc, ub, iters); break;
        case 11: valu_issue_loop<11>(a, p, u, b, c, pb, pc, ub, iters); break;
        case 12: valu_issue_loop<12>(a, p, u, b, c, pb, pc, ub, iters); break;
        case 13: valu_issue_loop<13>(a, p, u, b, c, pb, pc, ub, iters); break;
        case 14: valu_issue_loop<14>(a, p, u, b, c, pb, pc, ub, iters); break;
        case 15: valu_issue_loop<15>(a, p, u, b, c, pb, pc, ub, iters); break;
        case 16: valu_issue_loop<16>(a, p, u, b, c, pb, pc, ub, iters); break;
        case 17: valu_issue_loop<17>(a, p, u, b, c, pb, pc, ub, iters); break;
        case 18: valu_issue_loop<18>(a, p, u, b, c, pb, pc, ub, iters); break;
        case 19: valu_issue_loop<19>(a, p, u, b, c, pb, pc, ub, iters); break;
        case 20: valu_issue_loop<20>(a, p, u, b, c, pb, pc, ub, iters); break;
        case 21: valu_issue_loop<21>(a, p, u, b, c, pb, pc, ub, iters); break;
        case 22: valu_issue_loop<22>(a, p, u, b, c, pb, pc, ub, iters); break;
        case 23: valu_issue_loop<23>(a, p, u, b, c, pb, pc, ub, iters); break;
        case 24: valu_issue_loop<24>(a, p, u, b, c, pb, pc, ub, iters); break;
        default: valu_issue_loop<25>(a, p, u, b, c, pb, pc, ub, iters); break;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float sum = 0.0f;
    for (int i = 0; i < 16; ++i) sum += a[i] + p[i].x + p[i].y + __uint_as_float(u[i] & 0x3fffffffu);
    if (sum == 12345.678f) out[0] = sum + vi_smem[threadIdx.x];   // keeps the chains alive; never true
    if ((threadIdx.x & 63u) == 0u) {
        atomicAdd(ticks, t1 - t0);
        atomicAdd(ticks + 1, r1 - r0);   // the constant 100 MHz counter: shader clock = (t1 - t0) / (r1 - r0) x 100 MHz (MI355X_MICROARCH.md, DVFS item 6)
        atomicMax(ticks + 2, r1 - r0);   // longest / shortest span of a wave: the waves of a SIMD start together and do NOT finish together
        atomicMin(ticks + 3, r1 - r0);   // (the older wave is served first), so only the longest span is the SIMD's time
    }
}
template <int OP>
__device__ __forceinline__ void valu_issue_loop(float (&a)[16], gbl_f2 (&p)[16], uint32_t (&u)[16], float b, float c, gbl_f2 pb, gbl_f2 pc, uint32_t ub, uint32_t iters) {
#pragma unroll 1
    for (uint32_t it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
#define GBL_VI_FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define GBL_VI_PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pb), "v"(pc));
#define GBL_VI_ADD(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
#define GBL_VI_MAX3(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define GBL_VI_CVT(i) asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(u[i]));
#define GBL_VI_PERM(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(ub), "v"(ub));
#define GBL_VI_DPP(i) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(u[i]));
#define GBL_VI_CNDMASK(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(ub) : );
#define GBL_VI_AND(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[i]) : "v"(ub));
#define GBL_VI_RCP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
#define GBL_VI_MED3(i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define GBL_VI_CMP(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
            if constexpr (OP == 0) { GBL_VI_16(GBL_VI_FMA) }
            if constexpr (OP == 1) { GBL_VI_16(GBL_VI_PKFMA) }
            if constexpr (OP == 2) { GBL_VI_16(GBL_VI_ADD) }
            if constexpr (OP == 3) { GBL_VI_16(GBL_VI_MAX3) }
            if constexpr (OP == 4) { GBL_VI_16(GBL_VI_CVT) }
            if constexpr (OP == 5) { GBL_VI_16(GBL_VI_PERM) }
            if constexpr (OP == 6) { GBL_VI_16(GBL_VI_DPP) }
            if constexpr (OP == 7) { GBL_VI_16(GBL_VI_CNDMASK) }
            if constexpr (OP == 8) { GBL_VI_16(GBL_VI_AND) }
            if constexpr (OP == 9) { GBL_VI_16(GBL_VI_RCP) }
            if constexpr (OP == 10) { GBL_VI_16(GBL_VI_MED3) }
            if constexpr (OP == 11) { GBL_VI_16(GBL_VI_CMP) }
#define GBL_VI_MUL(i) asm volatile("v_mul_f32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define GBL_VI_FMAC(i) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(c));
#define GBL_VI_MAX(i) asm volatile("v_max_f32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define GBL_VI_MOV(i) asm volatile("v_mov_b32_e32 %0, %1" : "+v"(u[i]) : "v"(ub));
#define GBL_VI_ADDU(i) asm volatile("v_add_u32_e32 %0, %0, %1" : "+v"(u[i]) : "v"(ub));
#define GBL_VI_LSHL(i) asm volatile("v_lshlrev_b32_e32 %0, 1, %0" : "+v"(u[i]));
#define GBL_VI_ADD64(i) asm volatile("v_add_f32_e64 %0, %0, %1" : "+v"(a[i]) : "v"(c));
#define GBL_VI_FMAK(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
            if constexpr (OP == 12) { GBL_VI_16(GBL_VI_MUL) }
            if constexpr (OP == 13) { GBL_VI_16(GBL_VI_FMAC) }
            if constexpr (OP == 14) { GBL_VI_16(GBL_VI_MAX) }
            if constexpr (OP == 15) { GBL_VI_16(GBL_VI_MOV) }
            if constexpr (OP == 16) { GBL_VI_16(GBL_VI_ADDU) }
            if constexpr (OP == 17) { GBL_VI_16(GBL_VI_LSHL) }
            if constexpr (OP == 18) { GBL_VI_16(GBL_VI_ADD64) }   // v_add_f32 in its 64-bit (VOP3) encoding: is it the encoding or the operation?
            if constexpr (OP == 19) { GBL_VI_16(GBL_VI_FMAK) }    // v_fma_f32 reading two distinct registers instead of three
#define GBL_VI_MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(ub));
#define GBL_VI_MUL24(i) asm volatile("v_mul_u32_u24_e32 %0, %0, %1" : "+v"(u[i]) : "v"(ub));
#define GBL_VI_MULHI(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(u[i]) : "v"(ub));
#define GBL_VI_XOR(i) asm volatile("v_xor_b32_e32 %0, %0, %1" : "+v"(u[i]) : "v"(ub));
#define GBL_VI_LSHR(i) asm volatile("v_lshrrev_b32_e32 %0, 3, %0" : "+v"(u[i]));
#define GBL_VI_CNDS(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(u[i]) : "v"(ub));
            if constexpr (OP == 20) { GBL_VI_16(GBL_VI_MULLO) }   // the native sampler's hashes are made of these
            if constexpr (OP == 21) { GBL_VI_16(GBL_VI_MUL24) }
            if constexpr (OP == 22) { GBL_VI_16(GBL_VI_MULHI) }
            if constexpr (OP == 23) { GBL_VI_16(GBL_VI_XOR) }
            if constexpr (OP == 24) { GBL_VI_16(GBL_VI_LSHR) }
            if constexpr (OP == 25) { GBL_VI_16(GBL_VI_CNDS) }    // v_cndmask_b32 selecting by an SGPR pair other than vcc
        }
    }
}
extern "C" {

__global__ void selftest_sincos_kernel(const float* in, float* s, float* c, uint64_t n) {
    uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    s[i] = gbl_sinf(in[i]);
    c[i] = gbl_cosf(in[i]);
}

// rays: n x {kind (0 closest, 1 any), o(3), d(3), mint, maxt}; out: n x {t of the closest hit or -1 | 1 occluded or 0, instance, shading normal(3), tangent(3)}
__global__ void selftest_trace_kernel(DevScene sc, const float* rays, float* out, uint32_t n) {
    extern __shared__ __align__(16) unsigned char smem[];
    const LdsStack stk = {gbl_as_lds(reinterpret_cast<uint32_t*>(smem) + threadIdx.x)};
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* r = rays + 9 * i;
    LaneCounters cnt = {};
    Hit hit;
    hit.inst = -1;
    hit.t = -1.0f;
    const F3 o = f3(r[1], r[2], r[3]), d = f3(r[4], r[5], r[6]);
    float* q = out + 8 * i;
    for (int k = 0; k < 8; ++k) q[k] = 0.0f;
    if (r[0] != 0.0f) {
        q[0] = trace<true, false, true>(sc, o, d, r[7], r[8], stk, hit, cnt) ? 1.0f : 0.0f;
        q[1] = -1.0f;
    } else {
        const bool got = trace<false, false, true>(sc, o, d, r[7], r[8], stk, hit, cnt);
        q[0] = got ? hit.t : -1.0f;
        q[1] = got ? static_cast<float>(hit.inst) : -1.0f;
        if (got) {
            Frag fr;
            make_fragment<true>(sc, hit, o, d, fr);
            q[2] = fr.n.x; q[3] = fr.n.y; q[4] = fr.n.z;
            q[5] = fr.t.x; q[6] = fr.t.y; q[7] = fr.t.z;
        }
    }
}

static gbl_status gbl_selftest_trace_impl(gbl_ctx* ctx, const float* rays, float* out, uint32_t n) {
    if (!ctx || !rays || !out) return GBL_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (n == 0) return GBL_OK;
    const size_t lds = static_cast<size_t>(ctx->scene.stack_entries) * GBL_BLOCK * sizeof(uint32_t);
    if (lds > 64 * 1024)
        HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(selftest_trace_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         static_cast<int>(lds)));
    hipLaunchKernelGGL(selftest_trace_kernel, dim3((n + GBL_BLOCK - 1) / GBL_BLOCK), dim3(GBL_BLOCK), lds, nullptr, ctx->scene, rays, out, n);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipDeviceSynchronize());
    return GBL_OK;
}
gbl_status gbl_selftest_trace(gbl_ctx* ctx, const float* rays, float* out, uint32_t n) {
    return gbl_guard([&] { return gbl_selftest_trace_impl(ctx, rays, out, n); }, [&](const std::string& what) { if (ctx) ctx->error = what; });
}

// out[4 i ..] = {sqrtf(a), a / b, 1 / a, expected to be IEEE correctly rounded like the host's}
__global__ void selftest_arith_kernel(const float* a, const float* b, float* out, uint64_t n) {
    uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[4 * i] = sqrtf(a[i]);
    out[4 * i + 1] = a[i] / b[i];
    out[4 * i + 2] = 1.0f / a[i];
    const F3 v = normalize(f3(a[i], b[i], 0.5f));
    out[4 * i + 3] = v.x;
}
static gbl_status gbl_selftest_arith_impl(gbl_ctx* ctx, const float* a, const float* b, float* out, uint64_t n) {
    if (!ctx || !a || !b || !out) return GBL_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (n == 0) return GBL_OK;
    hipLaunchKernelGGL(selftest_arith_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, nullptr, a, b, out, n);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipDeviceSynchronize());
    return GBL_OK;
}
gbl_status gbl_selftest_arith(gbl_ctx* ctx, const float* a, const float* b, float* out, uint64_t n) {
    return gbl_guard([&] { return gbl_selftest_arith_impl(ctx, a, b, out, n); }, [&](const std::string& what) { if (ctx) ctx->error = what; });
}

// out[i] = fn(a[i] [, b[i]]) with fn of gbl_libm_fn, from kernels/refmath.h
__global__ void selftest_libm_kernel(int fn, const float* a, const float* b, float* out, uint64_t n) {
    uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = a[i], y = b ? b[i] : 0.0f;
    float r = 0.0f;
    switch (fn) {
        case GBL_LIBM_EXPF: r = gbl_expf(x); break;
        case GBL_LIBM_LOGF: r = gbl_logf(x); break;
        case GBL_LIBM_LOG2F: r = gbl_log2f(x); break;
        case GBL_LIBM_POWF: r = gbl_powf(x, y); break;
        case GBL_LIBM_ATANF: r = gbl_atanf(x); break;
        case GBL_LIBM_ATAN2F: r = gbl_atan2f(x, y); break;
        case GBL_LIBM_TANF: r = gbl_tanf(x); break;
        case GBL_LIBM_ACOSF: r = gbl_acosf(x); break;
        default: break;
    }
    out[i] = r;
}
static gbl_status gbl_selftest_libm_impl(gbl_ctx* ctx, int fn, const float* a, const float* b, float* out, uint64_t n) {
    if (!ctx || !a || !out || fn < 0 || fn > GBL_LIBM_ACOSF) return GBL_ERR_INVALID;
    if ((fn == GBL_LIBM_POWF || fn == GBL_LIBM_ATAN2F) && !b) return GBL_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (n == 0) return GBL_OK;
    hipLaunchKernelGGL(selftest_libm_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, nullptr, fn, a, b, out, n);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipDeviceSynchronize());
    return GBL_OK;
}
gbl_status gbl_selftest_libm(gbl_ctx* ctx, int fn, const float* a, const float* b, float* out, uint64_t n) {
    return gbl_guard([&] { return gbl_selftest_libm_impl(ctx, fn, a, b, out, n); }, [&](const std::string& what) { if (ctx) ctx->error = what; });
}

static gbl_status gbl_selftest_valu_issue_impl(gbl_ctx* ctx, int op, int waves_per_simd, uint32_t iters, double* out) {
    if (!ctx || !out || op < 0 || op >= GBL_VALU_OP_COUNT || waves_per_simd < 1 || waves_per_simd > 4 || iters == 0) return GBL_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    void (*const k)(float*, int, uint32_t, unsigned long long*) = valu_issue_kernel;
    const size_t lds = 96 * 1024;   // more than half of a CU's 160 KB: one workgroup per CU
    HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    float* d_out = nullptr;
    unsigned long long* d_ticks = nullptr;
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&d_out), 64));
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&d_ticks), 32));
    hipEvent_t e0, e1;
    HIP_TRY(ctx, hipEventCreate(&e0));
    HIP_TRY(ctx, hipEventCreate(&e1));
    const dim3 grid(ctx->num_cus), block(256 * waves_per_simd);
    float ms = 0.0f;
    unsigned long long ticks[4] = {0, 0, 0, 0};
    // The chip settles its clock under a load over seconds, not milliseconds (MI355X_MICROARCH.md, DVFS item 6): the same launch
    // back to back for warm_ms (2 s unless GBL_VALU_WARM_MS says otherwise) before the one that is read.
    double warm_ms = 2000.0;
    if (const char* e = getenv("GBL_VALU_WARM_MS")) warm_ms = atof(e);
    const auto w0 = std::chrono::steady_clock::now();
    do {
        for (int i = 0; i < 4; ++i) hipLaunchKernelGGL(k, grid, block, lds, nullptr, d_out, op, iters, d_ticks);
        HIP_TRY(ctx, hipDeviceSynchronize());
    } while (std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count() < warm_ms);
    {
        const unsigned long long init[4] = {0ull, 0ull, 0ull, ~0ull};
        HIP_TRY(ctx, hipMemcpy(d_ticks, init, 32, hipMemcpyHostToDevice));
    }
    HIP_TRY(ctx, hipEventRecord(e0, nullptr));
    hipLaunchKernelGGL(k, grid, block, lds, nullptr, d_out, op, iters, d_ticks);
    HIP_TRY(ctx, hipEventRecord(e1, nullptr));
    HIP_TRY(ctx, hipEventSynchronize(e1));
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipEventElapsedTime(&ms, e0, e1));
    HIP_TRY(ctx, hipMemcpy(ticks, d_ticks, 32, hipMemcpyDeviceToHost));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(d_out);
    (void)hipFree(d_ticks);
    const double waves = static_cast<double>(ctx->num_cus) * 4.0 * waves_per_simd;
    out[0] = ms;                                             // the launch, HIP events
    out[1] = waves * 64.0 * iters;                           // wave-instructions of the measured kind, whole launch
    out[2] = static_cast<double>(ticks[0]) / waves;          // s_memtime ticks per wave, first to last instruction
    out[3] = out[2] / (64.0 * iters);                        // ... per instruction of that wave
    out[4] = static_cast<double>(ticks[1]) / waves;          // s_memrealtime ticks (100 MHz) per wave over the same span
    out[5] = out[4] > 0.0 ? out[2] / out[4] * 0.1 : 0.0;      // shader clock in GHz while the loop ran
    out[6] = static_cast<double>(ticks[2]);                  // longest span of a wave, 100 MHz ticks
    out[7] = static_cast<double>(ticks[3]);                  // shortest
    return GBL_OK;
}
gbl_status gbl_selftest_valu_issue(gbl_ctx* ctx, int op, int waves_per_simd, uint32_t iters, double* out) {
    return gbl_guard([&] { return gbl_selftest_valu_issue_impl(ctx, op, waves_per_simd, iters, out); }, [&](const std::string& what) { if (ctx) ctx->error = what; });
}

static gbl_status gbl_selftest_sincos_impl(gbl_ctx* ctx, const float* in, float* sin_out, float* cos_out, uint64_t n) {
    if (!ctx || !in || !sin_out || !cos_out) return GBL_ERR_INVALID;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (n == 0) return GBL_OK;
    hipLaunchKernelGGL(selftest_sincos_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, nullptr, in, sin_out, cos_out, n);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipDeviceSynchronize());
    return GBL_OK;
}
gbl_status gbl_selftest_sincos(gbl_ctx* ctx, const float* in, float* sin_out, float* cos_out, uint64_t n) {
    return gbl_guard([&] { return gbl_selftest_sincos_impl(ctx, in, sin_out, cos_out, n); }, [&](const std::string& what) { if (ctx) ctx->error = what; });
}

}   // extern "C"
