// Version + error reporting of the C ABI.
#include <stdarg.h>

#include "ms_common.hpp"

namespace ms {
static thread_local char g_err[512] = "no error";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace ms

extern "C" int ms_version(void) { return MS_ABI_VERSION; }
extern "C" const char *ms_last_error_string(void) { return ms::g_err; }
