// libgoblin_hip.so, kernel unit: the persistent megakernel and the AO kernel, one ray per lane, under the native and
// replay samplers (kernels/render_kernels.h).  gbl_api.hip launches them through the selectors of gbl_internal.h.
#include "gbl_internal.h"
#include "kernels/render_kernels.h"

// exact_ties: the native sampler's lean kernel with the reference's tie rule and reachability test kept (trace.h TIES; every
// replay / instrumented / EXT-helper query follows them anyway)
gbl_render_kernel gbl_kernel_path(bool replay, bool stats, bool ext, bool exact_ties) {
    if (!replay && !stats && !ext && exact_ties) return path_trace_kernel<GBL_SRC_NATIVE, false, false, false, true>;
    if (stats) return replay ? path_trace_kernel<GBL_SRC_REPLAY, true, true> : path_trace_kernel<GBL_SRC_NATIVE, true, true>;   // instrumented builds are EXT
    if (replay) return ext ? path_trace_kernel<GBL_SRC_REPLAY, false, true> : path_trace_kernel<GBL_SRC_REPLAY, false, false>;
    return ext ? path_trace_kernel<GBL_SRC_NATIVE, false, true> : path_trace_kernel<GBL_SRC_NATIVE, false, false>;
}

gbl_render_kernel gbl_kernel_ao(bool replay, bool stats, bool ext, bool exact_ties) {
    if (!replay && !stats && !ext && exact_ties) return ao_kernel<GBL_SRC_NATIVE, false, false, false, true>;
    if (stats) return replay ? ao_kernel<GBL_SRC_REPLAY, true, true> : ao_kernel<GBL_SRC_NATIVE, true, true>;
    if (replay) return ext ? ao_kernel<GBL_SRC_REPLAY, false, true> : ao_kernel<GBL_SRC_REPLAY, false, false>;
    return ext ? ao_kernel<GBL_SRC_NATIVE, false, true> : ao_kernel<GBL_SRC_NATIVE, false, false>;
}
