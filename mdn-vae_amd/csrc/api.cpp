// Error-string plumbing and version query of the C ABI.
#include "common.h"
#include "../../include/mdnvae_hip.h"
#include <cstring>

namespace mdnvae {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace mdnvae

extern "C" {
int mdnvae_version(void) { return 100; }
const char* mdnvae_last_error(void) { return mdnvae::g_err; }
}
