// libgoblin_hip.so, kernel unit: the wave-pool schedule (kernels/wavepool.h).
#include "gbl_internal.h"
#include "kernels/wavepool.h"

gbl_render_kernel gbl_kernel_wavepool(bool replay, bool stats, bool ext) {
    if (stats) return replay ? wp_kernel<true, true, true> : wp_kernel<false, true, true>;   // instrumented builds are EXT
    if (replay) return ext ? wp_kernel<true, false, true> : wp_kernel<true, false, false>;
    return ext ? wp_kernel<false, false, true> : wp_kernel<false, false, false>;
}

uint32_t gbl_wavepool_slots(void) { return WP_SLOTS; }
uint64_t gbl_wavepool_bytes_per_wave(void) { return static_cast<uint64_t>(WP_FIELDS) * WP_SLOTS * sizeof(float4); }
