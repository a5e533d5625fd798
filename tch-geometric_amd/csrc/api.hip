// Version / error text of the C ABI.
#include "tg_host.h"

namespace tg {
char *last_error_buffer() {
    static thread_local char buf[512] = "";
    return buf;
}
} // namespace tg

extern "C" const char *tg_version(void) { return "tchgeo-gfx950 0.1.0"; }
extern "C" const char *tg_last_error(void) { return tg::last_error_buffer(); }
