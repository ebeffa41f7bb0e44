// Scratch: do two HIP streams (and the two branches of a captured graph) run small kernels CONCURRENTLY on this box?
//   hipcc --offload-arch=gfx950 -O2 tools/stream_overlap_test.hip -o /tmp/sot && /tmp/sot
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
__global__ void spin(long long cycles, int* out) {
    long long t0 = clock64();
    while (clock64() - t0 < cycles) {}
    if (out) out[blockIdx.x] = 1;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipStream_t a, b;
    hipStreamCreateWithFlags(&a, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
    int* d; hipMalloc(&d, 4096);
    const long long cyc = 2000000;   // ~1 ms at 2 GHz
    for (int blocks : {1, 64, 256, 512}) {
        for (int mode = 0; mode < 3; ++mode) {
            hipDeviceSynchronize();
            double t0 = now();
            if (mode == 0) {   // one stream, two launches
                hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, a, cyc, d);
                hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, a, cyc, d);
            } else if (mode == 1) {   // two streams
                hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, a, cyc, d);
                hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, b, cyc, d);
            } else {   // captured graph with a fork / join
                hipEvent_t e1, e2; hipEventCreate(&e1); hipEventCreate(&e2);
                hipGraph_t g; hipGraphExec_t ge;
                hipStreamBeginCapture(a, hipStreamCaptureModeGlobal);
                hipEventRecord(e1, a); hipStreamWaitEvent(b, e1, 0);
                hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, a, cyc, d);
                hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, b, cyc, d);
                hipEventRecord(e2, b); hipStreamWaitEvent(a, e2, 0);
                hipStreamEndCapture(a, &g);
                hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
                hipGraphLaunch(ge, a); hipStreamSynchronize(a);
                t0 = now();
                hipGraphLaunch(ge, a);
            }
            hipDeviceSynchronize();
            printf("blocks %4d  %-28s %.3f ms\n", blocks, mode == 0 ? "one stream x2" : mode == 1 ? "two streams" : "graph fork/join", (now() - t0) * 1e3);
        }
    }
    return 0;
}
