// capi.hip — library identification and error reporting of the C ABI.
#include "pn2_common.h"

thread_local int pn2_tls_hip_error = 0;

extern "C" int pn2_abi_version(void) { return 3; }   // bumped whenever entry points are added (round number)

extern "C" int pn2_last_hip_error(void) { return pn2_tls_hip_error; }

extern "C" const char *pn2_strerror(int code) {
  switch (code) {
    case PN2_OK: return "ok";
    case PN2_EINVAL: return "invalid size or argument combination";
    case PN2_ENULL: return "required pointer is NULL";
    case PN2_ELAUNCH: return "HIP kernel launch failed (see pn2_last_hip_error)";
    case PN2_ENOSPC: return "workspace too small";
    default: return "unknown pn2 error code";
  }
}
