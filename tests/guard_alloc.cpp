// An "electric fence" device allocator for the GPU test suite (test infrastructure; not part of the product library).
//
//     RSDF_GUARD_ALLOC=1 python -m pytest tests -m gpu        (tests/conftest.py installs it before the first allocation)
//
// torch's caching allocator carves tensors out of large mapped segments, so a kernel that reads or writes a little past the
// end of a tensor normally lands in mapped memory and nothing happens -- until the tensor happens to be the last piece of a
// segment (round 5: an over-read of 256 bytes in the per-wave SDF forward aborted the process only after another test file
// had shaped the free lists).  This allocator gives EVERY allocation its own mapping through HIP's virtual memory
// management API, placed so that the allocation ends < 16 bytes before a reserved, never-mapped granule: any access past the
// end of any tensor is a GPU memory fault on the spot, in every test that runs.
//
// Frees are deferred (the caching allocator's stream-ordered reuse is what the product relies on; a pluggable allocator's free
// arrives while kernels may still be in flight): a freed block is parked, and parked blocks are unmapped after a device
// synchronisation once enough of them have collected.
//
// build:  hipcc -O2 -shared -fPIC tests/guard_alloc.cpp -o tests/_guard_alloc.so      (__graft_entry__.build() does it)
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <unordered_map>
#include <vector>

namespace {
struct Block {
    void *base;
    size_t mapped, reserved;
};
std::mutex g_mu;
std::unordered_map<void *, Block> g_live;
std::vector<Block> g_parked;
size_t g_parked_bytes = 0, g_gran = 0;

void die(const char *what, hipError_t e)
{
    std::fprintf(stderr, "guard_alloc: %s failed: %s\n", what, hipGetErrorString(e));
    std::abort();
}

hipMemAllocationProp prop_of(int device)
{
    hipMemAllocationProp p = {};
    p.type = hipMemAllocationTypePinned;
    p.requestedHandleType = hipMemHandleTypeNone;
    p.location.type = hipMemLocationTypeDevice;
    p.location.id = device;
    return p;
}

void release(const Block &b)
{
    hipError_t e;
    if ((e = hipMemUnmap(b.base, b.mapped)) != hipSuccess) die("hipMemUnmap", e);
    // the address range stays reserved for the life of the process (RSDF_GUARD_REUSE_VA=1 gives it back): a later allocation
    // never lands on a freed tensor's addresses, so a use after free faults as well
    static const bool reuse = std::getenv("RSDF_GUARD_REUSE_VA") != nullptr;
    if (reuse && (e = hipMemAddressFree(b.base, b.reserved)) != hipSuccess) die("hipMemAddressFree", e);
}
}  // namespace

extern "C" void *guard_malloc(ssize_t size, int device, hipStream_t)
{
    if (size <= 0) return nullptr;
    std::lock_guard<std::mutex> lock(g_mu);
    hipError_t e;
    const hipMemAllocationProp prop = prop_of(device);
    if (g_gran == 0 &&
        (e = hipMemGetAllocationGranularity(&g_gran, &prop, hipMemAllocationGranularityMinimum)) != hipSuccess)
        die("hipMemGetAllocationGranularity", e);
    Block b;
    static const char *slack_env0 = std::getenv("RSDF_GUARD_SLACK");
    const size_t slack0 = slack_env0 ? (size_t)std::atol(slack_env0) : 0;
    b.mapped = ((size_t)size + slack0 + 15 + g_gran - 1) / g_gran * g_gran;
    b.reserved = b.mapped + g_gran;                     // the last granule stays reserved and unmapped: the fence
    if ((e = hipMemAddressReserve(&b.base, b.reserved, g_gran, nullptr, 0)) != hipSuccess) die("hipMemAddressReserve", e);
    hipMemGenericAllocationHandle_t h;
    if ((e = hipMemCreate(&h, b.mapped, &prop, 0)) != hipSuccess) {
        // out of device memory: give the parked blocks back and try once more, then report it torch's way (nullptr)
        hipDeviceSynchronize();
        for (const Block &p : g_parked) release(p);
        g_parked.clear();
        g_parked_bytes = 0;
        if ((e = hipMemCreate(&h, b.mapped, &prop, 0)) != hipSuccess) {
            hipMemAddressFree(b.base, b.reserved);
            return nullptr;
        }
    }
    if ((e = hipMemMap(b.base, b.mapped, 0, h, 0)) != hipSuccess) die("hipMemMap", e);
    if ((e = hipMemRelease(h)) != hipSuccess) die("hipMemRelease", e);      // the mapping keeps the memory alive
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    if ((e = hipMemSetAccess(b.base, b.mapped, &acc, 1)) != hipSuccess) die("hipMemSetAccess", e);
    // RSDF_GUARD_SLACK=<bytes>: mapped bytes left behind every allocation.  torch's own IndexBackward0 (index_put_ with
    // accumulate=True, the sort-based kernel) reads past the end of its operands (tools/debug/guard_torch_index_repro.py: pure
    // torch, faults at slack 0), so test files whose models index with torch run with a slack that covers torch's over-read
    // and still catch every larger one.
    static const char *slack_env = std::getenv("RSDF_GUARD_SLACK");
    static const size_t slack = slack_env ? (size_t)std::atol(slack_env) : 0;
    const uintptr_t end = (uintptr_t)b.base + b.mapped - slack;
    void *p = (void *)((end - (size_t)size) & ~(uintptr_t)15);          // 16-byte aligned, flush against the fence
    g_live.emplace(p, b);
    return p;
}

extern "C" void guard_free(void *p, ssize_t, int, hipStream_t)
{
    if (p == nullptr) return;
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_live.find(p);
    if (it == g_live.end()) {
        std::fprintf(stderr, "guard_alloc: free of an unknown pointer %p\n", p);
        std::abort();
    }
    g_parked.push_back(it->second);
    g_parked_bytes += it->second.mapped;
    g_live.erase(it);
    static const char *park_env = std::getenv("RSDF_GUARD_PARK");     // blocks parked before a release round (0: never release)
    static const size_t park = park_env ? (size_t)std::atol(park_env) : 512;
    static const char *gib_env = std::getenv("RSDF_GUARD_PARK_GIB");  // ... or this much parked memory (default 16 GiB)
    static const size_t park_bytes = (size_t)(gib_env ? std::atol(gib_env) : 16) << 30;
    if ((park != 0 && g_parked.size() >= park) || g_parked_bytes > park_bytes) {
        hipDeviceSynchronize();
        for (const Block &b : g_parked) release(b);
        g_parked.clear();
        g_parked_bytes = 0;
    }
}
