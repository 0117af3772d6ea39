// runtime.hip -- version + thread-local error text of the C ABI.
#include "common.h"

namespace isplib {
char *error_buffer() {
   static thread_local char buf[512] = {0};
   return buf;
}
}  // namespace isplib

extern "C" int isplib_hip_abi_version(void) { return ISPLIB_HIP_ABI_VERSION; }
extern "C" const char *isplib_hip_last_error(void) { return isplib::error_buffer(); }
