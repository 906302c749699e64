// Library-level entry points of libst_hip.so.
#include "common.h"

namespace st {
char* error_buffer() {
  static thread_local char buf[512] = {0};
  return buf;
}
}  // namespace st

extern "C" int st_version(void) { return 120; }  // 0.1.20: + G8 training chains, sub-pixel upsample forward

extern "C" const char* st_last_error(void) { return st::error_buffer(); }
