// Thread-local error string behind mt3_last_error().
#include <string>

#include "common.h"

namespace {
thread_local std::string g_last_error;
}

namespace mt3 {
int fail(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}
}  // namespace mt3

extern "C" {
const char* mt3_last_error(void) { return g_last_error.c_str(); }
int mt3_abi_version(void) { return 4; }   // 4: mt3_engine_transcribe (in-flight batching); 3: MT3_DECODE_ASYNC + mt3_engine_decode_wait, row retirement
}
