// Shared helpers for libzeroshape_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

namespace zs {

// thread-local last-error text behind zs_last_error()
char *err_buf();
void set_err(const char *fmt, ...);

inline bool check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_err("%s: %s", what, hipGetErrorString(e));
        return false;
    }
    return true;
}

}  // namespace zs
