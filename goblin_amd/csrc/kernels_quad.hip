// libgoblin_hip.so, kernel unit: the persistent megakernel whose sparse interior steps put four lanes on each ray
// (kernels/quadtrace.h), under the native and replay samplers.
#include "gbl_internal.h"
#include "kernels/render_kernels.h"

gbl_render_kernel gbl_kernel_path_quad(bool replay, bool stats, bool ext) {
    if (stats) return replay ? path_trace_kernel<GBL_SRC_REPLAY, true, true, true> : path_trace_kernel<GBL_SRC_NATIVE, true, true, true>;   // instrumented builds are EXT
    if (replay) return ext ? path_trace_kernel<GBL_SRC_REPLAY, false, true, true> : path_trace_kernel<GBL_SRC_REPLAY, false, false, true>;
    return ext ? path_trace_kernel<GBL_SRC_NATIVE, false, true, true> : path_trace_kernel<GBL_SRC_NATIVE, false, false, true>;
}
gbl_render_kernel gbl_kernel_ao_quad(bool replay) {   // the lean AO kernel only
    return replay ? ao_kernel<GBL_SRC_REPLAY, false, false, true> : ao_kernel<GBL_SRC_NATIVE, false, false, true>;
}
uint32_t gbl_quad_lds_words(void) { return GBL_QUAD_LDS_WORDS; }
