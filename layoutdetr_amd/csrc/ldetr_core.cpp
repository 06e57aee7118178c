// Error reporting + ABI version for the layoutdetr_amd C ABI (include/ldetr_hip.h).
#include <stdarg.h>
#include <stdio.h>
#include "../../include/ldetr_hip.h"

namespace ldetr {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace ldetr

extern "C" const char* ldetr_last_error(void) { return ldetr::g_err; }
extern "C" int ldetr_abi_version(void) { return 22; }
