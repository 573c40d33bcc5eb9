#include "common_host.h"

namespace vlsfr {
char* error_buffer() {
  static thread_local char buf[512] = {0};
  return buf;
}
}  // namespace vlsfr

extern "C" {
const char* vlsfr_last_error(void) { return vlsfr::error_buffer(); }
int vlsfr_version(void) { return 100; }
}
