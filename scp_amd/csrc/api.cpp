// Library-level entry points of libscp_hip.so.
#include <string.h>
#include "scp_internal.h"

int g_scp_last_hip_error = 0;

extern "C" int scp_version(void) { return 100; }
extern "C" int scp_last_hip_error(void) { return g_scp_last_hip_error; }

extern "C" int scp_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int scp_device_name(char *buf, int cap) {
    if (!buf || cap <= 0) return SCP_EINVAL;
    hipDeviceProp_t p;
    HIP_TRY(hipGetDeviceProperties(&p, 0));
    strncpy(buf, p.name, cap - 1);
    buf[cap - 1] = 0;
    return SCP_OK;
}
