// libgoblin_hip.so, kernel unit: the path and AO kernels under GBL_SAMPLES_STREAM -- the reference's own per-tile
// mt19937 sample stream generated on the device (kernels/stream.h, the STREAM instantiations of kernels/render_kernels.h).
#include "gbl_internal.h"
#include "kernels/render_kernels.h"

gbl_render_kernel gbl_kernel_path_stream(bool stats, bool ext) {
    if (stats) return path_trace_kernel<GBL_SRC_STREAM, true, true>;
    return ext ? path_trace_kernel<GBL_SRC_STREAM, false, true> : path_trace_kernel<GBL_SRC_STREAM, false, false>;
}

// the lean stream kernel with quad-per-ray queries (kernels/quadtrace.h): a pixel's 256 paths are not refilled as they end, so its
// waves spend most of their queries with a few live rays
gbl_render_kernel gbl_kernel_path_stream_quad(void) { return path_trace_kernel<GBL_SRC_STREAM, false, false, true>; }

gbl_render_kernel gbl_kernel_ao_stream(bool ext) {   // (not instrumented)
    return ext ? ao_kernel<GBL_SRC_STREAM, false, true> : ao_kernel<GBL_SRC_STREAM, false, false>;
}
