// libgoblin_hip.so, kernel unit: the persistent megakernel whose extension queries leave their stragglers for the next
// iteration (kernels/suspend.h), under the native and replay samplers.
#include "gbl_internal.h"
#include "kernels/render_kernels.h"

gbl_render_kernel gbl_kernel_path_suspend(bool replay, bool stats, bool ext) {
    if (stats) return replay ? path_trace_kernel<true, true, true, false, false, false, true> : path_trace_kernel<false, true, true, false, false, false, true>;   // instrumented builds are EXT
    if (replay) return ext ? path_trace_kernel<true, false, true, false, false, false, true> : path_trace_kernel<true, false, false, false, false, false, true>;
    return ext ? path_trace_kernel<false, false, true, false, false, false, true> : path_trace_kernel<false, false, false, false, false, false, true>;
}
uint32_t gbl_suspend_park_words(void) { return GBL_SUSP_WORDS; }
