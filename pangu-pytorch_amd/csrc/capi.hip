// ABI bookkeeping for libpangu_hip.so
#include "common.h"

extern "C" int pangu_abi_version(void) { return PANGU_ABI_VERSION; }

extern "C" const char* pangu_error_string(int code) {
  switch (code) {
    case PANGU_OK: return "ok";
    case PANGU_E_SHAPE: return "unsupported shape";
    case PANGU_E_NULL: return "null pointer";
    case PANGU_E_DTYPE: return "unsupported dtype";
    case PANGU_E_ARG: return "invalid argument";
    case PANGU_E_RANGE: return "matrix spans 4 GB or more (32-bit byte offsets): split the call by rows";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
  }
}
