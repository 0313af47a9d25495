// Library-level entry points: version and per-thread error text.
#include "svae_internal.h"

namespace svae {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace svae

extern "C" int svae_version(void) { return 100; }

extern "C" void svae_last_error(char* buf, size_t n) {
  if (!buf || n == 0) return;
  strncpy(buf, svae::g_err, n - 1);
  buf[n - 1] = '\0';
}
