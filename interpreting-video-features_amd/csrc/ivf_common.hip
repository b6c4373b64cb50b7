// Error reporting and version for libivf_hip.so.
#include <cstring>

#include "ivf_common.h"

namespace ivf {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace ivf

extern "C" const char* ivf_last_error(void) { return ivf::g_err; }
extern "C" int ivf_version(void) { return 100; }
