// Error plumbing + ABI version for libzeroshape_hip.so.
#include "zs_common.h"
#include "../../include/zeroshape_hip.h"

namespace zs {

char *err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}

void set_err(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
}

}  // namespace zs

extern "C" int zs_abi_version(void) { return ZS_ABI_VERSION; }
extern "C" const char *zs_last_error(void) { return zs::err_buf(); }
