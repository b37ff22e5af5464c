// libgoblin_hip.so, kernel unit: the persistent megakernel whose sparse interior steps put four lanes on each ray
// (kernels/quadtrace.h), under the native and replay samplers.
#include "gbl_internal.h"
#include "kernels/render_kernels.h"

gbl_render_kernel gbl_kernel_path_quad(bool replay, bool stats, bool ext) {
    if (stats) return replay ? path_trace_kernel<true, true, true, false, false, false, false, true> : path_trace_kernel<false, true, true, false, false, false, false, true>;   // instrumented builds are EXT
    if (replay) return ext ? path_trace_kernel<true, false, true, false, false, false, false, true> : path_trace_kernel<true, false, false, false, false, false, false, true>;
    return ext ? path_trace_kernel<false, false, true, false, false, false, false, true> : path_trace_kernel<false, false, false, false, false, false, false, true>;
}
// ... whose extension queries also park their last stragglers until the wave's next query (the lean kernels only)
gbl_render_kernel gbl_kernel_path_quad_park(bool replay) {
    return replay ? path_trace_kernel<true, false, false, false, false, false, true, true> : path_trace_kernel<false, false, false, false, false, false, true, true>;
}
uint32_t gbl_quad_park_words(void) { return GBL_QUAD_PARK_WORDS; }
gbl_render_kernel gbl_kernel_ao_quad(bool replay) {   // the lean AO kernel only
    return replay ? ao_kernel<true, false, false, false, true> : ao_kernel<false, false, false, false, true>;
}
uint32_t gbl_quad_lds_words(void) { return GBL_QUAD_LDS_WORDS; }
