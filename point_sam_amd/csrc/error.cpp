// Last-error string + version for libpointsam_hip.so (host only).
#include "common.h"
#include <string.h>

static thread_local char g_err[256] = "";

void psam_set_error(const char* msg) {
    strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = 0;
}

PSAM_API const char* psam_last_error_string(void) { return g_err; }
PSAM_API int32_t psam_version(void) { return 100; }  // 0.1.0
