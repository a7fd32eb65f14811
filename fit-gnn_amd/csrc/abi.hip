// abi.hip -- version and error-string entry points of libfitgnn_hip.so.
#include <hip/hip_runtime.h>

#include "fitgnn_hip.h"

extern "C" int fitgnn_abi_version(void) { return FITGNN_ABI_VERSION; }

extern "C" const char *fitgnn_error_string(int code) {
    switch (code) {
        case 0: return "success";
        case FITGNN_E_BADARG: return "fitgnn: bad argument (null pointer, negative size or unsupported shape)";
        case FITGNN_E_WORKSPACE: return "fitgnn: workspace too small (see *_workspace_bytes)";
        case FITGNN_E_ALIGN: return "fitgnn: pointer or leading dimension misaligned";
        default: break;
    }
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "fitgnn: unknown error code";
}
