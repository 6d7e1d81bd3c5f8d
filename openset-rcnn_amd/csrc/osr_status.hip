// Error reporting and ABI version (include/osr.h).
#include "osr_common.h"

static thread_local char g_err[512] = "";

void osr_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* osr_last_error(void) { return g_err; }
extern "C" int32_t osr_abi_version(void) { return OSR_ABI_VERSION; }
