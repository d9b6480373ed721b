// Library-wide state: ABI version and the last-error message.
#include "common.h"
#include <stdlib.h>
#include <string.h>

static thread_local char g_err[256] = "";

extern "C" void rsdf_set_error(const char *msg)
{
    strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = 0;
}

int rsdf_func_lds(const void *kernel, size_t bytes)
{
    constexpr int SLOTS = 64;
    static thread_local const void *done[SLOTS] = {};
    static thread_local size_t done_key[SLOTS] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const size_t key = bytes * 64 + (size_t)(dev & 63) + 1;
    int slot = -1;
    for (int i = 0; i < SLOTS; ++i) {
        if (done[i] == kernel && done_key[i] == key) return 0;
        if (slot < 0 && (done[i] == nullptr || done[i] == kernel)) slot = i;
    }
    hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) { rsdf_set_error(hipGetErrorString(e)); return (int)e; }
    if (slot >= 0) {       // (a full table only means the attribute is set again next time)
        done[slot] = kernel;
        done_key[slot] = key;
    }
    return 0;
}

bool rsdf_env_is(const char *name, const char *value)
{
    // a handful of fixed names: cache the first lookup of each
    struct Entry { const char *name; const char *val; };
    constexpr int SLOTS = 16;
    static Entry cache[SLOTS] = {};
    static int n = 0;
    const char *v = nullptr;
    bool found = false;
    for (int i = 0; i < n; ++i)
        if (strcmp(cache[i].name, name) == 0) { v = cache[i].val; found = true; break; }
    if (!found) {
        v = getenv(name);
        if (n < SLOTS) { cache[n].name = name; cache[n].val = v; ++n; }   // benign race: same value either way
    }
    return v != nullptr && strcmp(v, value) == 0;
}

extern "C" int rsdf_abi_version(void) { return RSDF_ABI_VERSION; }
extern "C" const char *rsdf_last_error(void) { return g_err; }
