// Shared host-side helpers for libmt3hip.so (error reporting; no device code here).
#ifndef MT3_COMMON_H_
#define MT3_COMMON_H_

#include <string>

#include "mt3_hip.h"

namespace mt3 {

// Records the message for mt3_last_error() (thread-local) and returns `code`.
int fail(int code, const std::string& msg);

}  // namespace mt3

#define MT3_HIP_CHECK(expr)                                                                        \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess)                                                                          \
      return ::mt3::fail(MT3_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));          \
  } while (0)

#endif  // MT3_COMMON_H_
