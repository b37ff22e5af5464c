// libgoblin_hip.so, kernel unit: the persistent megakernel with the workgroup's ray exchange (kernels/rayexchange.h) under the
// native and replay samplers.
#include "gbl_internal.h"
#include "kernels/render_kernels.h"

gbl_render_kernel gbl_kernel_path_exchange(bool replay, bool stats, bool ext) {
    if (stats) return replay ? path_trace_kernel<true, true, true, false, false, true> : path_trace_kernel<false, true, true, false, false, true>;   // instrumented builds are EXT
    if (replay) return ext ? path_trace_kernel<true, false, true, false, false, true> : path_trace_kernel<true, false, false, false, false, true>;
    return ext ? path_trace_kernel<false, false, true, false, false, true> : path_trace_kernel<false, false, false, false, false, true>;
}
uint32_t gbl_ray_exchange_lds_words(void) { return GBL_RX_LDS_WORDS; }
