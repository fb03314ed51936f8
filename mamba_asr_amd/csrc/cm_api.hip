// cm_api.hip — ABI version, thread-local error string.
#include "cm_common.h"

static thread_local char g_cm_error[512] = "";

void cm_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_cm_error, sizeof(g_cm_error), fmt, ap);
    va_end(ap);
}

extern "C" int cm_abi_version(void) { return CM_ABI_VERSION; }
extern "C" const char *cm_last_error(void) { return g_cm_error; }
extern "C" int cm_scan_num_chunks(int seqlen) { return (seqlen + CM_SCAN_CHUNK - 1) / CM_SCAN_CHUNK; }
