// Library-wide state: ABI version and the last-error message.
#include "common.h"
#include <string.h>

static thread_local char g_err[256] = "";

extern "C" void rsdf_set_error(const char *msg)
{
    strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = 0;
}

extern "C" int rsdf_abi_version(void) { return RSDF_ABI_VERSION; }
extern "C" const char *rsdf_last_error(void) { return g_err; }
