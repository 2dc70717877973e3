// capi.hip -- error reporting and device discovery of libsubgacc_hip.so.
#include "common.hpp"
#include <string.h>

namespace subgacc {

static thread_local char g_error[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}

}  // namespace subgacc

extern "C" int subgacc_abi_version(void) { return SUBGACC_ABI_VERSION; }

extern "C" const char *subgacc_last_error(void) { return subgacc::g_error; }

extern "C" int subgacc_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    int ok = 0;
    for (int d = 0; d < n; ++d) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, d) == hipSuccess && strncmp(p.gcnArchName, "gfx950", 6) == 0) ++ok;
    }
    return ok;
}
