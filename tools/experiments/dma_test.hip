// scratch: semantics of buffer_load_dwordx4 ... lds on gfx950 (lane i -> M0 base + 16*i ? out-of-range -> zeros ?)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
__global__ void k(const uint16_t* in, uint16_t* out, int n) {
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    for (int i = threadIdx.x; i < 4 * 512; i += 256) lds[i] = 0xAAAA;
    __syncthreads();
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, n * 2, 0x00020000);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t off = (uint32_t)((wave * 64 + (lane ^ 1)) * 16);
    if (lane == 5) off = 0x80000000u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(lds + wave * 512), 16, off, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = 0; i < 8; ++i) out[threadIdx.x * 8 + i] = lds[threadIdx.x * 8 + i];
}
int main() {
    const int n = 256 * 8;
    std::vector<uint16_t> h(n), o(n);
    for (int i = 0; i < n; ++i) h[i] = (uint16_t)i;
    uint16_t *di, *dout;
    hipMalloc(&di, n * 2); hipMalloc(&dout, n * 2);
    hipMemcpy(di, h.data(), n * 2, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 4 * 1024, 0, di, dout, n);
    hipMemcpy(o.data(), dout, n * 2, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 256; ++t) {
        int wave = t >> 6, lane = t & 63;
        for (int e = 0; e < 8; ++e) {
            uint16_t want = lane == 5 ? 0 : (uint16_t)((wave * 64 + (lane ^ 1)) * 8 + e);
            if (o[t * 8 + e] != want) { if (bad < 8) printf("t %d e %d got %u want %u\n", t, e, o[t * 8 + e], want); ++bad; }
        }
    }
    printf("dma test: %d mismatches (%s)\n", bad, hipGetErrorString(hipGetLastError()));
    return bad != 0;
}
