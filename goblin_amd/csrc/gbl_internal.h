// Internal to libgoblin_hip.so: the context behind the C ABI's opaque handle and the table of device kernels.
//
// The kernels live in translation units of their own (kernels_*.hip), compiled side by side; the host side of the ABI
// (gbl_api.hip) picks an instantiation through the selectors below and launches it by pointer.  A selector returns
// nullptr for a combination that is not built.
#pragma once
#include <hip/hip_runtime.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/goblin_hip.h"
#include "device_scene.h"
#include "kernels/wf_args.h"

struct gbl_ctx {
    int device = 0;
    std::string error;
    std::vector<void*> allocations;
    DevScene scene;
    gbl_info info;
    uint32_t* work_counter = nullptr;
    void* prim_buf = nullptr;         // the primary pass's hits: entries x (float4 + int32)
    uint64_t prim_entries = 0;
    uint32_t* prim_items = nullptr;   // ... and its word per work item of the path kernel (RenderArgs::prim_items)
    uint64_t prim_items_cap = 0;
    unsigned long long* stats = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int num_cus = 256;
    void* rccl = nullptr;
    void* rccl_allreduce = nullptr;
    // wavefront pool (allocated on first use)
    uint32_t wf_pool = 0;
    uint32_t* wf_spill = nullptr;   // wf_ensure_spill()
    hipStream_t wf_aux = nullptr;   // shadow rays of iteration k trace here while the main stream traces extension rays k+1
    hipEvent_t wf_ev_shade = nullptr, wf_ev_shadow = nullptr;
    int wf_spill_levels = 0;
    WfArgs wf;
    float4* wf_li = nullptr;
    size_t wf_li_entries = 0;
    uint64_t li_budget = 0;   // li_budget_bytes()
    uint32_t* stream_seeds = nullptr;     // GBL_SAMPLES_STREAM: per-tile mt19937 seeds of the full sample window
    uint32_t* stream_scratch = nullptr;   // ... and the workgroups' sample-generation scratch
    uint64_t stream_scratch_bytes = 0;
    float* stream_xy = nullptr;           // ... and the image position of every camera sample of the call (for the splat)
    uint64_t stream_xy_bytes = 0;
    float* vol_buf = nullptr;    // per-sample {transmittance, Lv} of the render in flight (scenes with a participating medium)
    uint64_t vol_entries = 0;
    float4* sss_buf = nullptr;   // per-sample Lsubsurface of the render in flight (scenes with subsurface materials)
    uint64_t sss_entries = 0;
    std::map<int, float> auto_rays_per_path;   // GBL_SCHEDULE_AUTO's pilot: rays per camera path by 2 * max_ray_depth + russian_roulette (gbl_render)
    double build_ms = 0.0;    // pack_scene + BVH construction + node / triangle upload
    // what gbl_update_instances needs to rebuild the TLAS
    std::vector<gbl_instance> h_instances;
    std::vector<uint32_t> h_light_slots;   // DevLight::wh_n per light (the Whitted quota, host copy for the stream sampler's layout)
    std::vector<gbl_mesh> h_meshes;
    std::vector<gbl_material> h_materials;
    std::vector<float> mesh_lo, mesh_hi;
    std::vector<int32_t> mesh_root;
    int32_t tlas_base = 0;
    uint32_t tlas_capacity = 0;
    int blas_depth = 0;
    std::vector<int> mesh_stack_need;   // scene_prep.h PackedScene::mesh_stack_need
    bool has_directional = false;
    bool has_images = false;     // the scene holds MIP pyramids (image textures / image based lights)
    uint32_t* wf_host_flags = nullptr;   // pinned
    // ring of event triples for gbl_get_timings
    static const int kTimingRing = 64;
    hipEvent_t t_ev[64][3] = {};
    unsigned long long t_calls = 0;
};

#define HIP_TRY(ctx, expr)                                                                     \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            (ctx)->error = std::string(#expr) + ": " + hipGetErrorString(e_);                  \
            return GBL_ERR_DEVICE;                                                             \
        }                                                                                      \
    } while (0)

// ---- kernel table ----------------------------------------------------------------------------------------------
typedef void (*gbl_render_kernel)(DevScene, RenderArgs);
typedef void (*gbl_wf_kernel)(DevScene, RenderArgs, WfArgs);
typedef void (*gbl_li_kernel)(DevScene, RenderArgs, float4*);

// kernels_path.hip: the persistent megakernel and the AO kernel (kernels/render_kernels.h), native / replay samplers
gbl_render_kernel gbl_kernel_path(bool replay, bool stats, bool ext, bool exact_ties = false);
gbl_render_kernel gbl_kernel_ao(bool replay, bool stats, bool ext, bool exact_ties = false);
// kernels_quad.hip: the megakernel whose sparse interior steps put four lanes on each ray (kernels/quadtrace.h)
gbl_render_kernel gbl_kernel_path_quad(bool exact_ties);
gbl_render_kernel gbl_kernel_path_quad_primary(bool exact_ties);   // ... its paths starting at RenderArgs::prim_hit (primary_kernel's output)
gbl_render_kernel gbl_kernel_path_stream_quad(void);
gbl_render_kernel gbl_kernel_ao_quad(bool exact_ties);
void gbl_launch_primary(const DevScene& sc, const RenderArgs& ra, bool exact_ties, float4* prim_hit, int32_t* prim_inst, unsigned blocks, hipStream_t stream);   // kernels/packet.h
uint32_t gbl_quad_lds_words(void);          // LDS words of the quads' records, in the film tile's place
// kernels_stream.hip: the same two under GBL_SAMPLES_STREAM (kernels/stream.h)
gbl_render_kernel gbl_kernel_path_stream(bool stats, bool ext);
gbl_render_kernel gbl_kernel_ao_stream(bool ext);
// kernels_wavefront.hip (kernels/wavefront.h)
gbl_wf_kernel gbl_kernel_wf_trace(bool any, bool stats, bool ext, bool masks, bool ties);
gbl_wf_kernel gbl_kernel_wf_shade(bool replay, bool stats, bool ext);
gbl_wf_kernel gbl_kernel_wf_splat(bool replay, bool stats);
// kernels_whitted.hip (kernels/whitted.h)
gbl_li_kernel gbl_kernel_whitted(bool replay);
gbl_li_kernel gbl_kernel_whitted_stream(void);
// kernels_aux.hip: first-hit passes (kernels/subsurface.h, kernels/volume.h), film resolve, device BLAS build, self tests
gbl_li_kernel gbl_kernel_sss(bool replay);
gbl_render_kernel gbl_kernel_vol(bool replay);
void gbl_launch_vol_combine(float4* li, const float4* vol, uint64_t n, hipStream_t stream);
void gbl_launch_tri_bounds_gather(const DevTri* tris, const DevTriBound* by_id, DevTriBound* out, uint32_t n);
void gbl_launch_film_resolve(const float* accum, float* rgb, int n, hipStream_t stream);
gbl_status gbl_build_blas_device(gbl_ctx* ctx, const float* d_pos, const uint32_t* d_idx, uint32_t n, const float* lo, const float* hi,
                                 DevNode* d_nodes, int32_t node_base, DevTri* d_tris, uint32_t tri_base, uint32_t shade_base, uint32_t tri_flags,
                                 int32_t* root_out, uint32_t* nodes_out, int* depth_out);
