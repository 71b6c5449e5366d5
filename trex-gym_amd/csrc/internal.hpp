// Shared by the translation units of libtrex_hip.so (not installed, not part of the C-ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

// last-error plumbing of trex_last_error(): sets the thread-local message, returns `code`
int trex_fail(int code, const std::string &msg);

// caller allocations already validated as memory of a device (base address, bytes known to be good)
struct TrexSeen { const void *p; size_t bytes; };
int trex_check_device_buffer(int device, std::vector<TrexSeen> &seen, const void *p, size_t bytes, const char *what);

struct TrexDeviceGuard {
  int prev = -1;
  bool ok;
  explicit TrexDeviceGuard(int dev) {
    ok = hipGetDevice(&prev) == hipSuccess && (prev == dev || hipSetDevice(dev) == hipSuccess);
    if (prev == dev) prev = -1;
  }
  ~TrexDeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};
