// api_common.cpp -- error state, device selection and raw memory helpers of the C ABI.
#include "common.h"
#include <string.h>
#include <atomic>

namespace adas {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
int hip_fail(hipError_t e, const char* what, const char* file, int line) {
    set_error("HIP error %d (%s) at %s:%d: %s", (int)e, hipGetErrorString(e), file, line, what);
    (void)hipGetLastError();
    return ADAS_ERR_HIP;
}
static std::atomic<unsigned long long> g_cfg_gen{1};
unsigned long long config_generation() { return g_cfg_gen.load(std::memory_order_acquire); }
void bump_config_generation() { g_cfg_gen.fetch_add(1, std::memory_order_acq_rel); }
}  // namespace adas

extern "C" {
const char* adas_last_error(void) { return adas::g_err; }
int adas_version(void) { return 100; }
int adas_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}
int adas_set_device(int index) {
    ADAS_REQUIRE(adas_device_count() > 0, ADAS_ERR_NO_DEVICE, "no HIP device visible; this library has no CPU fallback");
    ADAS_HIP_TRY(hipSetDevice(index));
    return ADAS_OK;
}
int adas_device_pci_bus_id(int index, char* out, int out_len) {
    ADAS_REQUIRE(out && out_len >= 16, ADAS_ERR_INVALID, "adas_device_pci_bus_id: buffer of at least 16 bytes needed");
    ADAS_REQUIRE(adas_device_count() > 0, ADAS_ERR_NO_DEVICE, "no HIP device visible; this library has no CPU fallback");
    ADAS_HIP_TRY(hipDeviceGetPCIBusId(out, out_len, index));
    return ADAS_OK;
}
int adas_malloc(void** d_ptr, size_t bytes) {
    ADAS_REQUIRE(d_ptr, ADAS_ERR_INVALID, "adas_malloc: null out pointer");
    ADAS_REQUIRE(adas_device_count() > 0, ADAS_ERR_NO_DEVICE, "no HIP device visible; this library has no CPU fallback");
    ADAS_HIP_TRY(hipMalloc(d_ptr, bytes ? bytes : 16));
    return ADAS_OK;
}
int adas_free(void* d_ptr) {
    if (d_ptr) ADAS_HIP_TRY(hipFree(d_ptr));
    return ADAS_OK;
}
int adas_memcpy_h2d(void* d, const void* h, size_t n) {
    ADAS_HIP_TRY(hipMemcpy(d, h, n, hipMemcpyHostToDevice));
    return ADAS_OK;
}
int adas_memcpy_d2h(void* h, const void* d, size_t n) {
    ADAS_HIP_TRY(hipMemcpy(h, d, n, hipMemcpyDeviceToHost));
    return ADAS_OK;
}
int adas_host_alloc(void** h_ptr, size_t bytes) {
    ADAS_REQUIRE(h_ptr, ADAS_ERR_INVALID, "adas_host_alloc: null out pointer");
    ADAS_REQUIRE(adas_device_count() > 0, ADAS_ERR_NO_DEVICE, "no HIP device visible; this library has no CPU fallback");
    ADAS_HIP_TRY(hipHostMalloc(h_ptr, bytes ? bytes : 16, hipHostMallocDefault));
    return ADAS_OK;
}
int adas_host_free(void* h_ptr) {
    if (h_ptr) ADAS_HIP_TRY(hipHostFree(h_ptr));
    return ADAS_OK;
}
int adas_synchronize(void) {
    ADAS_HIP_TRY(hipDeviceSynchronize());
    return ADAS_OK;
}
}

struct adas_timer {
    hipEvent_t a = nullptr, b = nullptr;
};
extern "C" {
int adas_timer_create(adas_timer** out) {
    ADAS_REQUIRE(out, ADAS_ERR_INVALID, "adas_timer_create: null out pointer");
    ADAS_REQUIRE(adas_device_count() > 0, ADAS_ERR_NO_DEVICE, "no HIP device visible; this library has no CPU fallback");
    adas_timer* t = new adas_timer();
    if (hipEventCreate(&t->a) != hipSuccess || hipEventCreate(&t->b) != hipSuccess) {
        adas_timer_destroy(t);
        return adas::hip_fail(hipGetLastError(), "hipEventCreate", __FILE__, __LINE__);
    }
    *out = t;
    return ADAS_OK;
}
int adas_timer_destroy(adas_timer* t) {
    if (!t) return ADAS_OK;
    if (t->a) (void)hipEventDestroy(t->a);
    if (t->b) (void)hipEventDestroy(t->b);
    delete t;
    return ADAS_OK;
}
int adas_timer_start(adas_timer* t, void* stream) {
    ADAS_REQUIRE(t, ADAS_ERR_INVALID, "null timer");
    ADAS_HIP_TRY(hipEventRecord(t->a, (hipStream_t)stream));
    return ADAS_OK;
}
int adas_timer_stop(adas_timer* t, void* stream) {
    ADAS_REQUIRE(t, ADAS_ERR_INVALID, "null timer");
    ADAS_HIP_TRY(hipEventRecord(t->b, (hipStream_t)stream));
    return ADAS_OK;
}
int adas_timer_elapsed_ms(adas_timer* t, float* ms) {
    ADAS_REQUIRE(t && ms, ADAS_ERR_INVALID, "null argument");
    ADAS_HIP_TRY(hipEventSynchronize(t->b));
    ADAS_HIP_TRY(hipEventElapsedTime(ms, t->a, t->b));
    return ADAS_OK;
}
}
