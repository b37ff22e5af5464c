// libgoblin_hip.so, kernel unit: the path and AO kernels under GBL_SAMPLES_STREAM -- the reference's own per-tile
// mt19937 sample stream generated on the device (kernels/stream.h, the STREAM instantiations of kernels/render_kernels.h).
#include "gbl_internal.h"
#include "kernels/render_kernels.h"

gbl_render_kernel gbl_kernel_path_stream(bool stats, bool ext) {
    if (stats) return path_trace_kernel<GBL_SRC_STREAM, true, true>;
    return ext ? path_trace_kernel<GBL_SRC_STREAM, false, true> : path_trace_kernel<GBL_SRC_STREAM, false, false>;
}

gbl_render_kernel gbl_kernel_ao_stream(bool ext) {   // (not instrumented)
    return ext ? ao_kernel<GBL_SRC_STREAM, false, true> : ao_kernel<GBL_SRC_STREAM, false, false>;
}
