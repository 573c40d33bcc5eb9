// Error plumbing shared by every translation unit of libvlsfr.so.
#pragma once
#include <cstdarg>
#include <cstdio>

#include "vlsfr.h"

namespace vlsfr {

char* error_buffer();  // thread-local, 512 bytes (common.cpp)

inline int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(error_buffer(), 512, fmt, ap);
  va_end(ap);
  return code;
}

// per-device, thread-safe hipFuncSetAttribute(MaxDynamicSharedMemorySize) cache (common.cpp)
int ensure_dynamic_lds(const void* kernel, int bytes, const char* who);

}  // namespace vlsfr
