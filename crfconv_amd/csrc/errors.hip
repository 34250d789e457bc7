// Thread-local error message + ABI version.
#include "common.hpp"

namespace crf {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace crf

extern "C" const char* crfconv_last_error(void) { return crf::g_err; }
extern "C" int crfconv_abi_version(void) { return CRFCONV_ABI_VERSION; }
