// capi.hip — library identification and error reporting of the C ABI.
#include "pn2_common.h"

thread_local int pn2_tls_hip_error = 0;

extern "C" int pn2_abi_version(void) { return 11; }   // bumped whenever entry points are added

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

// ---- timing events (measurement only: bench.py's per-kernel table) ----------------------------------------------
// torch.cuda.Event records with a system-scope release fence: every record writes the L2 back and costs ~20 us of queue
// time — 500 records around the 250 launches of a step stretched a sampled step by ~11 ms.  These events are created with
// hipEventDisableSystemFence (timestamps only, no cache maintenance).
extern "C" void *pn2_event_create(void) {
  hipEvent_t ev = nullptr;
  if (hipEventCreateWithFlags(&ev, hipEventDisableSystemFence) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  return (void *)ev;
}
extern "C" int pn2_event_record(void *ev, void *stream) {
  if (!ev) return PN2_ENULL;
  return hipEventRecord((hipEvent_t)ev, (hipStream_t)stream) == hipSuccess ? PN2_OK : pn2_check_launch();
}
extern "C" int pn2_event_elapsed_ms(void *start, void *stop, float *ms) {
  if (!start || !stop || !ms) return PN2_ENULL;
  return hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop) == hipSuccess ? PN2_OK : pn2_check_launch();
}
extern "C" int pn2_event_destroy(void *ev) {
  if (!ev) return PN2_OK;
  return hipEventDestroy((hipEvent_t)ev) == hipSuccess ? PN2_OK : pn2_check_launch();
}
