// libgoblin_hip.so, kernel unit: the Whitted integrator (kernels/whitted.h).
#include "gbl_internal.h"
#include "kernels/whitted.h"

gbl_li_kernel gbl_kernel_whitted(bool replay, bool stats) {
    if (replay) return stats ? whitted_kernel<true, true> : whitted_kernel<true, false>;
    return stats ? whitted_kernel<false, true> : whitted_kernel<false, false>;
}

gbl_li_kernel gbl_kernel_whitted_stream(bool stats) { return stats ? whitted_stream_kernel<true> : whitted_stream_kernel<false>; }
