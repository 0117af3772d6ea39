// common.h -- error plumbing shared by the HIP translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#include "../../include/isplib_hip.h"

namespace isplib {

char *error_buffer();   // thread-local, 512 bytes (defined in runtime.hip)
int empty_row_init();   // 1: an empty row of max / min keeps the reference launcher's pre-fill (isplib_hip_set_empty_row; runtime.hip)

inline void clear_error() { error_buffer()[0] = '\0'; }

inline int fail(int code, const char *msg) {
   snprintf(error_buffer(), 512, "%s", msg);
   return code;
}

inline int hip_fail(hipError_t e, const char *what) {
   snprintf(error_buffer(), 512, "%s: %s", what, hipGetErrorString(e));
   return ISPLIB_HIP_ERROR;
}

inline int check_launch(const char *what) {
   const hipError_t e = hipGetLastError();
   return e == hipSuccess ? ISPLIB_SUCCESS : hip_fail(e, what);
}

#define ISPLIB_HIP_TRY(expr)                                        \
   do {                                                             \
      const hipError_t e_ = (expr);                                 \
      if (e_ != hipSuccess) return ::isplib::hip_fail(e_, #expr);   \
   } while (0)

}  // namespace isplib
