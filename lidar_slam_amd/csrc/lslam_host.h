// lslam_host.h — host-side plumbing shared by the C-ABI translation units (internal).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/lslam.h"

namespace lslam {

// thread-local message behind lslam_last_error()
int fail(int code, const char* fmt, ...);
void clear_error();

#define LSLAM_HIP_TRY(expr)                                                                  \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess)                                                                    \
      return ::lslam::fail(LSLAM_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                           __FILE__, __LINE__);                                              \
  } while (0)

// Library-owned device scratch: one growing arena per (host thread, device).  carve() hands out
// 256-byte aligned slices; reserve() must be called first with the total.
class Arena {
 public:
  int reserve(int device, size_t bytes);      // LSLAM_OK or error
  void* carve(size_t bytes);
  void release();                             // frees the block (the caller has synchronised the device)
  static size_t align(size_t b) { return (b + 255) & ~(size_t)255; }

 private:
  void* base_ = nullptr;
  size_t cap_ = 0;
  size_t used_ = 0;
  int device_ = -1;
};
Arena& thread_arena();

// one stream per host thread for the host-pointer entry points
int thread_stream(hipStream_t* out);

}  // namespace lslam
