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

extern "C" const char* mojo_hip_version(void) { return "mojo_hip 0.1.0 (gfx950)"; }
extern "C" const char* mojo_hip_last_error(void) { return mojo::g_err; }
