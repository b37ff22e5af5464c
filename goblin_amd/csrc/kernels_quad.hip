// libgoblin_hip.so, kernel unit: the persistent megakernel whose sparse interior steps put four lanes on each ray
// (kernels/quadtrace.h), under the native and replay samplers.
#include "gbl_internal.h"
#include "kernels/render_kernels.h"

// (the lean kernels of the native sampler only: the EXT builds are slower under the quad queries, the replay / instrumented builds
//  run one ray per lane -- gbl_api.hip.  exact_ties: the reference's tie rule and reachability test kept, trace.h TIES)
gbl_render_kernel gbl_kernel_path_quad(bool exact_ties) {
    return exact_ties ? path_trace_kernel<GBL_SRC_NATIVE, false, false, true, true> : path_trace_kernel<GBL_SRC_NATIVE, false, false, true>;
}
gbl_render_kernel gbl_kernel_ao_quad(bool exact_ties) {
    return exact_ties ? ao_kernel<GBL_SRC_NATIVE, false, false, true, true> : ao_kernel<GBL_SRC_NATIVE, false, false, true>;
}
uint32_t gbl_quad_lds_words(void) { return GBL_QUAD_LDS_WORDS; }
