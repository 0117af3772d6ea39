// runtime.hip -- version + thread-local error text of the C ABI.
#include <stdlib.h>
#include <string.h>

#include "common.h"

namespace isplib {
char *error_buffer() {
   static thread_local char buf[512] = {0};
   return buf;
}
}  // namespace isplib

// The one convention of the path that nothing in the reference tree pins: what an EMPTY row of a max / min
// SpMM holds.  The reference's launcher pre-fills the output with lowest() / max() and the positions with nnz
// (csrc/fusedmm.cpp:147-150,171,177) and hands both to fusedMM_csr, whose body is absent: a body that leaves untouched what it
// has no entry for returns -FLT_MAX / +FLT_MAX ("init"); torch_sparse's CPU kernel, the one the iSpLib authors compared with
// (isplib/__init__.py:120-128), writes 0 ("zero", the default here and in the oracle).  Process-wide: ISPLIB_EMPTY_ROW=init|zero
// read once, or isplib_hip_set_empty_row.  The positions are nnz either way; sum / mean are 0 either way.
namespace isplib {
static int g_empty_row = -1;
int empty_row_init() {
   if (g_empty_row < 0) {
      const char *e = getenv("ISPLIB_EMPTY_ROW");
      g_empty_row = (e && (strcmp(e, "init") == 0 || strcmp(e, "1") == 0)) ? 1 : 0;
   }
   return g_empty_row;
}
}  // namespace isplib

extern "C" int isplib_hip_set_empty_row(int init) {
   isplib::clear_error();
   if (init != 0 && init != 1) return isplib::fail(ISPLIB_FAIL, "isplib_hip_set_empty_row: 0 (zero) or 1 (the launcher's init value)");
   isplib::g_empty_row = init;
   return ISPLIB_SUCCESS;
}
extern "C" int isplib_hip_get_empty_row(void) { return isplib::empty_row_init(); }

extern "C" int isplib_hip_abi_version(void) { return ISPLIB_HIP_ABI_VERSION; }
extern "C" const char *isplib_hip_last_error(void) { return isplib::error_buffer(); }
