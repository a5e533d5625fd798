// Version / error text of the C ABI.
#include "tg_host.h"

namespace tg {
char *last_error_buffer() {
    static thread_local char buf[512] = "";
    return buf;
}
static unsigned int *g_debug_bounds_flag = nullptr;
unsigned int *debug_bounds_flag() { return g_debug_bounds_flag; }
} // namespace tg

extern "C" int tg_debug_bounds_set_flag(uint32_t *device_word) {
#ifdef TG_DEBUG_BOUNDS
    tg::g_debug_bounds_flag = device_word;
    return TG_OK;
#else
    (void)device_word;
    return tg::fail(TG_ERR_UNSUPPORTED, "tg_debug_bounds_set_flag: this build has no bounds checks (make dbg)");
#endif
}

extern "C" const char *tg_version(void) { return "tchgeo-gfx950 0.1.0"; }
extern "C" const char *tg_last_error(void) { return tg::last_error_buffer(); }
