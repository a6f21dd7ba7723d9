// Library-level entry points and the thread-local error message.
#include "common.h"

namespace advhip {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace advhip

extern "C" int advhip_abi_version(void) { return ADVHIP_ABI_VERSION; }
extern "C" const char* advhip_last_error(void) { return advhip::g_err; }
extern "C" const char* advhip_target_arch(void) { return "gfx950"; }
