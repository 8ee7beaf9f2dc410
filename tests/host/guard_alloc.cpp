// guard_alloc.cpp — test infrastructure (tests/test_oob_guard.py): device buffers that end (or start) exactly at the edge of MAPPED
// address space, built with HIP's virtual-memory API.  A reservation of [granule | n granules | granule] is made, only the middle is
// backed by memory: an access one byte past the buffer's end (before its start) is a GPU page fault — the process dies — instead of
// a silent read of whatever torch's caching allocator placed next to the tensor.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

namespace {
hipMemAllocationProp props() {
    hipMemAllocationProp p = {};
    p.type = hipMemAllocationTypePinned;
    p.location.type = hipMemLocationTypeDevice;
    int dev = 0;
    (void)hipGetDevice(&dev);
    p.location.id = dev;
    return p;
}
}  // namespace

extern "C" {

int guard_granularity(size_t* g) {
    const hipMemAllocationProp p = props();
    return (int)hipMemGetAllocationGranularity(g, &p, hipMemAllocationGranularityMinimum);
}

// bytes > 0.  tail != 0: the buffer's last byte is the last mapped byte; tail == 0: its first byte is the first mapped byte.
// user: the buffer (16-byte aligned when bytes is a multiple of 16 or tail == 0); base / reserved / mapped / handle: for guard_free.
int guard_alloc(size_t bytes, int tail, void** user, void** base, size_t* reserved, size_t* mapped, unsigned long long* handle) {
    size_t g = 0;
    int rc = guard_granularity(&g);
    if (rc != 0 || g == 0) return rc ? rc : -1;
    const size_t m = (bytes + g - 1) / g * g, r = m + 2 * g;
    void* va = nullptr;
    if ((rc = (int)hipMemAddressReserve(&va, r, g, nullptr, 0)) != 0) return rc;
    const hipMemAllocationProp p = props();
    hipMemGenericAllocationHandle_t h;
    if ((rc = (int)hipMemCreate(&h, m, &p, 0)) != 0) { (void)hipMemAddressFree(va, r); return rc; }
    char* mid = static_cast<char*>(va) + g;
    if ((rc = (int)hipMemMap(mid, m, 0, h, 0)) != 0) { (void)hipMemRelease(h); (void)hipMemAddressFree(va, r); return rc; }
    hipMemAccessDesc acc = {};
    acc.location = p.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    if ((rc = (int)hipMemSetAccess(mid, m, &acc, 1)) != 0) { (void)hipMemUnmap(mid, m); (void)hipMemRelease(h); (void)hipMemAddressFree(va, r); return rc; }
    *user = tail ? mid + (m - bytes) : mid;
    *base = va; *reserved = r; *mapped = m; *handle = (unsigned long long)(uintptr_t)h;
    return 0;
}

int guard_free(void* base, size_t reserved, size_t mapped, unsigned long long handle) {
    size_t g = (reserved - mapped) / 2;
    int rc = (int)hipMemUnmap(static_cast<char*>(base) + g, mapped);
    rc |= (int)hipMemRelease((hipMemGenericAllocationHandle_t)(uintptr_t)handle);
    rc |= (int)hipMemAddressFree(base, reserved);
    return rc;
}

// synchronous device-to-device copy and byte fill (the buffers are no torch tensors)
int guard_copy(void* dst, const void* src, size_t bytes) { return (int)hipMemcpy(dst, src, bytes, hipMemcpyDeviceToDevice); }
int guard_fill(void* dst, int byte, size_t bytes) { return (int)hipMemset(dst, byte, bytes); }
const char* guard_error(int rc) { return hipGetErrorString((hipError_t)rc); }

}  // extern "C"
