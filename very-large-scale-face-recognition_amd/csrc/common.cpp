#include <map>
#include <mutex>
#include <utility>

#include <hip/hip_runtime_api.h>

#include "common_host.h"

namespace vlsfr {
char* error_buffer() {
  static thread_local char buf[512] = {0};
  return buf;
}
// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device property of a kernel; launches come from several host threads
// (forward on the caller's thread, backward on autograd's): one mutex-guarded table keyed by (device, kernel) remembers the
// largest size set so far, so the attribute call happens once per device and kernel (or when a launch needs more).
int ensure_dynamic_lds(const void* kernel, int bytes, const char* who) {
  static std::mutex mu;
  static std::map<std::pair<int, const void*>, int> done;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return fail(VLSFR_EHIP, "%s: hipGetDevice: %s", who, hipGetErrorString(e));
  std::lock_guard<std::mutex> lk(mu);
  int& cur = done[std::make_pair(dev, kernel)];
  if (bytes > cur) {
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return fail(VLSFR_EHIP, "%s: hipFuncSetAttribute: %s", who, hipGetErrorString(e));
    cur = bytes;
  }
  return VLSFR_OK;
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
