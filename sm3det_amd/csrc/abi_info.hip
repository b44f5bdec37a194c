// abi_info.hip -- version / error-string entry points of libsm3det_hip.so
#include "common.h"

extern "C" {

const char* sm3_version(void) { return "sm3det_hip 0.1.0 gfx950"; }

const char* sm3_compiler_version(void) {
#if defined(__clang_version__)
  return "hipcc clang " __clang_version__;
#else
  return "hipcc";
#endif
}

const char* sm3_error_string(int code) {
  switch (code) {
    case SM3_OK: return "ok";
    case SM3_ERR_INVALID_ARG: return "invalid argument";
    case SM3_ERR_WORKSPACE: return "workspace too small";
    case SM3_ERR_LAUNCH: return "kernel launch failed";
    case SM3_ERR_UNSUPPORTED: return "unsupported configuration";
    default: return "unknown error";
  }
}

}  // extern "C"
