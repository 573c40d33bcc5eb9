#include <hip/hip_runtime_api.h>

#include "common_host.h"

namespace vlsfr {
char* error_buffer() {
  static thread_local char buf[512] = {0};
  return buf;
}
}  // namespace vlsfr

extern "C" {
const char* vlsfr_last_error(void) { return vlsfr::error_buffer(); }
int vlsfr_version(void) { return 300; }
int vlsfr_bn_repl(void) { return VLSFR_BN_REPL; }

// ---- section 10 of include/vlsfr.h: HIP events for cross-stream ordering of the multi-GPU step
int vlsfr_event_create(void** ev) {
  if (!ev) return vlsfr::fail(VLSFR_EINVAL, "vlsfr_event_create: null argument");
  hipEvent_t e;
  hipError_t rc = hipEventCreateWithFlags(&e, hipEventDisableTiming);
  if (rc != hipSuccess) return vlsfr::fail(VLSFR_EHIP, "vlsfr_event_create: %s", hipGetErrorString(rc));
  *ev = (void*)e;
  return VLSFR_OK;
}
void vlsfr_event_destroy(void* ev) {
  if (ev) (void)hipEventDestroy((hipEvent_t)ev);
}
int vlsfr_event_record(void* ev, void* stream) {
  if (!ev) return vlsfr::fail(VLSFR_EINVAL, "vlsfr_event_record: null event");
  hipError_t rc = hipEventRecord((hipEvent_t)ev, (hipStream_t)stream);
  return rc == hipSuccess ? VLSFR_OK : vlsfr::fail(VLSFR_EHIP, "vlsfr_event_record: %s", hipGetErrorString(rc));
}
int vlsfr_stream_wait_event(void* stream, void* ev) {
  if (!ev) return vlsfr::fail(VLSFR_EINVAL, "vlsfr_stream_wait_event: null event");
  hipError_t rc = hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)ev, 0);
  return rc == hipSuccess ? VLSFR_OK : vlsfr::fail(VLSFR_EHIP, "vlsfr_stream_wait_event: %s", hipGetErrorString(rc));
}
}
