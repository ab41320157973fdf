// Library-wide C-ABI plumbing: version and thread-local last-error text.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

extern "C" void miq_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* miq_last_error(void) { return g_err; }

extern "C" int miq_version(void) { return 303; }  // major*10000 + minor*100 + patch
