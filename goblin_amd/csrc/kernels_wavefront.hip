// libgoblin_hip.so, kernel unit: the wavefront schedule (kernels/wavefront.h) and the shared splat kernel.
#include "gbl_internal.h"
#include "kernels/wavefront.h"

// MASKS builds are EXT; instrumented builds are EXT; the reference's tie rule and reachability test (trace.h TIES) are only ever
// left out of the native sampler's lean (non-EXT), un-instrumented kernels.
gbl_wf_kernel gbl_kernel_wf_trace(bool any, bool stats, bool ext, bool masks, bool ties) {
    if (masks) {
        if (any) return stats ? wf_trace<true, true, true, true> : wf_trace<true, false, true, true>;
        return stats ? wf_trace<false, true, true, true> : wf_trace<false, false, true, true>;
    }
    if (any && stats) return wf_trace<true, true, true>;
    if (any && (ties || ext)) return ext ? wf_trace<true, false, true> : wf_trace<true, false, false>;
    if (any) return wf_trace<true, false, false, false, false>;
    if (stats) return wf_trace<false, true, true>;
    if (ties || ext) return ext ? wf_trace<false, false, true> : wf_trace<false, false, false>;
    return wf_trace<false, false, false, false, false>;
}

gbl_wf_kernel gbl_kernel_wf_shade(bool replay, bool stats, bool ext) {
    if (stats) return replay ? wf_shade<true, true, true> : wf_shade<false, true, true>;
    if (replay) return ext ? wf_shade<true, false, true> : wf_shade<true, false, false>;
    return ext ? wf_shade<false, false, true> : wf_shade<false, false, false>;
}

gbl_wf_kernel gbl_kernel_wf_splat(bool replay, bool stats) {
    if (replay) return stats ? wf_splat<true, true> : wf_splat<true, false>;
    return stats ? wf_splat<false, true> : wf_splat<false, false>;
}
