// cm2_core.hip -- error reporting and device queries of the C ABI
#include "cm2_common.h"

#include <cstdarg>

namespace cm2 {
static thread_local char g_err[1024] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace cm2

extern "C" const char *cm2_last_error(void) { return cm2::g_err; }

extern "C" int cm2_abi_version(void) { return CM2_ABI_VERSION; }

extern "C" int cm2_device_info(int device, char *h_name, int *h_num_cu, double *h_hbm_gib)
{
    hipDeviceProp_t prop;
    CM2_HIP(hipGetDeviceProperties(&prop, device));
    if (h_name) {
        strncpy(h_name, prop.gcnArchName, 255);
        h_name[255] = 0;
    }
    if (h_num_cu) *h_num_cu = prop.multiProcessorCount;
    if (h_hbm_gib) *h_hbm_gib = (double)prop.totalGlobalMem / (1024.0 * 1024.0 * 1024.0);
    return 0;
}
