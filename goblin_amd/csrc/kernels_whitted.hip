// libgoblin_hip.so, kernel unit: the Whitted integrator (kernels/whitted.h).
#include "gbl_internal.h"
#include "kernels/whitted.h"

// (not instrumented: gbl_stats reports the camera samples of a Whitted render, no ray counters)
gbl_li_kernel gbl_kernel_whitted(bool replay) { return replay ? whitted_kernel<true, false> : whitted_kernel<false, false>; }

gbl_li_kernel gbl_kernel_whitted_stream(void) { return whitted_stream_kernel<false>; }
