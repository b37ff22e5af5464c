// Argument block of the wavefront kernels (kernels/wavefront.h): the path pool and the per-wave queue regions.
// A header of its own so the host side can hold one without pulling the kernels in.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct WfArgs {
    // pool (one entry per slot)
    float4* ray_o;      // o.xyz, mint
    float4* ray_d;      // d.xyz, -
    float4* hit;        // t, b1, b2, as_float(tri)
    int32_t* hit_inst;  // -1: miss
    float4* s_thr;      // throughput.xyz, cosw
    float4* s_li;       // Li.xyz, fw
    float4* s_ld;       // this vertex's light-sampled term (f * L * |n.wi| * w / pdf).xyz, bsdf_pdf -- it counts iff s_vis says so
    float4* s_f;        // f.xyz, pick_pdf
    uint2* s_id;        // (light | (bounce + 4) << 16 | punch << 30 | first << 31, out_index): the sample number and the native
                        // sampler's pixel key follow from out_index
    uint8_t* s_vis;     // 1: the slot's shadow ray reached the light (written by wf_trace<true>, cleared by wf_shade with the ray)
    // per-wave queue regions: region w covers entries [64 w, 64 w + count[w])
    uint32_t* ext_q;      // slot ids
    uint32_t* ext_count;
    float4* sh_d;         // shadow ray: d.xyz, maxt -- its origin and mint are the slot's ray_o (the vertex it leaves from)
    uint32_t* sh_slot;    // ... and the slot it reports to
    float4* sh_c;         // mask scenes only: (lightPdf, isArea, -, -) -- the product is formed after the attenuation walk
    uint32_t* sh_count;
    uint32_t* wave_next;  // per shade-wave {ids handed out so far, the block taken from the shared reserve that is in use}: only the
                          // owning wave ever touches its two words
    uint32_t* steal_next; // the shared reserve's counter: blocks of 64 path ids beyond the statically dealt ones (wf_shade)
    uint32_t static_blocks;   // blocks dealt to every wave up front, round-robin (block j of wave w = j * waves + w)
    uint32_t total_blocks;    // total_paths / 64
    // mask scenes only (DevScene::has_masks): the isOpaque-filtered MIS hit + attenuation of an extension ray whose
    // closest hit is a mask (-2 in hit2_inst: same as the closest hit), and the shadow queue's un-multiplied terms
    float4* hit2;         // t, b1, b2, as_float(tri)
    int32_t* hit2_inst;
    float4* mis_tr;       // attenuation.xyz
    float4* sh_f;         // f.xyz, |n.wi|
    float4* sh_L;         // L.xyz, lWeight
    uint32_t* stack_spill; // global backing of the trace kernels' stacks beyond GBL_WF_STACK_LDS levels (SplitStack)
    uint32_t* live_flags; // [8]: set by wf_shade when any of its slots is still alive
    float4* li_buf;       // per-sample radiance of the pass, pixel-major: pixel * pass_spp + kk
    uint32_t pool_size;   // P, multiple of 256
    uint32_t total_paths; // path ids in this pass: local_tiles * 64 * pass_spp
    int32_t pass_k0, pass_spp;
    int32_t init;         // first wf_shade of a pass: every slot is empty
    int32_t flag_index;
};
