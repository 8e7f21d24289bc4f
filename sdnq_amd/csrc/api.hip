// Library-level entry points of the C ABI (include/sdnq_hip.h).
#include <hip/hip_runtime.h>
#include <string.h>

#include "../../include/sdnq_hip.h"

extern "C" int sdnq_hip_version(void) { return SDNQ_HIP_ABI_VERSION; }

extern "C" const char* sdnq_hip_strerror(int status) {
    switch (status) {
        case SDNQ_OK: return "ok";
        case SDNQ_ERR_NULL: return "required pointer is NULL";
        case SDNQ_ERR_DTYPE: return "unknown or unsupported dtype";
        case SDNQ_ERR_SHAPE: return "bad shape (M/N/K/group/rank)";
        case SDNQ_ERR_ALIGN: return "pointer or leading dimension is not 16-byte aligned";
        case SDNQ_ERR_UNSUPPORTED: return "valid in the reference but not implemented by this build";
        case SDNQ_ERR_ARCH: return "device is not gfx950 (MI355X)";
        case SDNQ_ERR_LAUNCH: return "kernel launch failed";
        case SDNQ_ERR_WORKSPACE: return "workspace too small";
        default: return "unknown status";
    }
}

// The reference parses gcnArchName[3:] as hex and compares (sdnext.py:101-105); this build targets
// exactly one ISA, so the gate is a prefix match on "gfx950".
extern "C" int sdnq_hip_device_supported(int ordinal) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, ordinal) != hipSuccess) return 0;
    return strncmp(prop.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
}
