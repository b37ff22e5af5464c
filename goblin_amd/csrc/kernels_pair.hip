// libgoblin_hip.so, kernel unit: the megakernel with paired shadow + extension queries (kernels/pairkernel.h).
#include "gbl_internal.h"
#include "kernels/pairkernel.h"

gbl_render_kernel gbl_kernel_pair(bool replay, bool stats, bool ext) {
    if (stats) return replay ? pair_trace_kernel<true, true, true> : pair_trace_kernel<false, true, true>;   // instrumented builds are EXT
    if (replay) return ext ? pair_trace_kernel<true, false, true> : pair_trace_kernel<true, false, false>;
    return ext ? pair_trace_kernel<false, false, true> : pair_trace_kernel<false, false, false>;
}
