// placeholder, replaced below
#include "gbl_internal.h"
gbl_render_kernel gbl_kernel_wavepool(bool, bool, bool) { return nullptr; }
uint32_t gbl_wavepool_slots(void) { return 0; }
uint64_t gbl_wavepool_bytes_per_wave(void) { return 0; }
