// Error text + version entry points.
#include <stdarg.h>

#include "common.h"

namespace mojo {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace mojo

// MOJO_SRC_HASH: sha256 (first 16 hex digits) over csrc/*.hip, csrc/*.h, csrc/experiments/*.h and include/*.h, given
// by csrc/build.py to THIS file only; the Python loader recomputes it from the tree beside the library and refuses a
// library built from other sources (backends/hip/lib.py).
#ifndef MOJO_SRC_HASH
#define MOJO_SRC_HASH "unstamped"
#endif
#ifdef MOJO_HIP_BUILD_EXPERIMENTS
#define MOJO_BUILD_KIND "+experiments"
#else
#define MOJO_BUILD_KIND ""
#endif
extern "C" const char* mojo_hip_version(void) {
  return "mojo_hip 0.2.0 (gfx950" MOJO_BUILD_KIND ") src=" MOJO_SRC_HASH;
}
extern "C" const char* mojo_hip_last_error(void) { return mojo::g_err; }
