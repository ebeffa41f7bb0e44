// bt_bench.hip -- standalone latency harness for the device ByteTrack update (track_core.h).
// Scratch tool: synthetic random-walk boxes, S streams, F frames; prints average kernel time of the last frames and a
// checksum of every frame's output messages so two builds of track_core.h can be compared for identical results.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I vehicle-cv-adas_amd/csrc tools/experiments/bt_bench.hip -o tools/experiments/bt_bench
//   tools/experiments/bt_bench [streams=16] [objects=60] [frames=120] [MT=256] [MD=256]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#ifdef ADAS_BT_PROF
__device__ unsigned long long g_bt_prof[32];
#endif
#include "track_core.h"

using namespace adas;

struct BtDev {
    BtParams P;
    unsigned char* base;
    size_t stream_bytes;
    const double* xyxy;
    const double* score;
    const int* cls;
    const int* counts;
    int det_stride;
};

static inline size_t bt_align(size_t x) { return (x + 63) & ~(size_t)63; }
__host__ __device__ inline size_t bt_align_d(size_t x) { return (x + 63) & ~(size_t)63; }
static size_t bt_stream_bytes(int MT, int MD) {
    return bt_align(sizeof(BtHeader)) + bt_align((size_t)MT * sizeof(BtTrack)) + bt_align((size_t)MT * MD * 8) +
           bt_align((size_t)2 * MT * sizeof(BtOut)) + 2 * bt_align((size_t)MT * 4) + bt_align((size_t)MT * ADAS_BT_TRAJ * 32);
}
__host__ __device__ inline BtStream bt_view(unsigned char* p, int MT, int MD) {
    BtStream S;
    S.hdr = (BtHeader*)p; p += bt_align_d(sizeof(BtHeader));
    S.slots = (BtTrack*)p; p += bt_align_d((size_t)MT * sizeof(BtTrack));
    S.cost = (double*)p; p += bt_align_d((size_t)MT * MD * 8);
    S.out = (BtOut*)p; p += bt_align_d((size_t)2 * MT * sizeof(BtOut));
    S.tracked = (int*)p; p += bt_align_d((size_t)MT * 4);
    S.lost = (int*)p; p += bt_align_d((size_t)MT * 4);
    S.traj = (double*)p;
    return S;
}

__global__ __launch_bounds__(256) void bt_kernel(BtDev d) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int s = blockIdx.x;
    BtStream S = bt_view(d.base + (size_t)s * d.stream_bytes, d.P.MT, d.P.MD);
    BtDet det;
    det.tlbr = d.xyxy + (size_t)s * d.det_stride * 4;
    det.score = d.score + (size_t)s * d.det_stride;
    det.cls = d.cls + (size_t)s * d.det_stride;
    det.nd = d.counts[s];
    Ctx c{(int)threadIdx.x, (int)blockDim.x};
    bytetrack_update(c, d.P, S, det, smem);
}

static unsigned long long rng_state = 0x9E3779B97F4A7C15ull;
static double urand() {
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
    return (double)(rng_state >> 11) / 9007199254740992.0;
}

struct Obj { double cx, cy, w, h, vx, vy, base; int cls; int hidden; };

int main(int argc, char** argv) {
    int S = argc > 1 ? atoi(argv[1]) : 16, NOBJ = argc > 2 ? atoi(argv[2]) : 60, F = argc > 3 ? atoi(argv[3]) : 120;
    int MT = argc > 4 ? atoi(argv[4]) : 256, MD = argc > 5 ? atoi(argv[5]) : 256;
    BtDev d;
    d.P = BtParams{0.5, 0.6, 0.8, 30, MT, MD};
    d.stream_bytes = bt_stream_bytes(MT, MD);
    unsigned char* arena;
    hipMalloc(&arena, d.stream_bytes * S);
    hipMemset(arena, 0, d.stream_bytes * S);
    d.base = arena;
    double *dx, *ds; int *dc, *dn;
    hipMalloc(&dx, (size_t)S * MD * 32); hipMalloc(&ds, (size_t)S * MD * 8); hipMalloc(&dc, (size_t)S * MD * 4); hipMalloc(&dn, S * 4);
    d.xyxy = dx; d.score = ds; d.cls = dc; d.counts = dn; d.det_stride = MD;
    size_t lds = BtLds::bytes(MT, MD, 256);
    if (hipFuncSetAttribute((const void*)bt_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { printf("lds attr failed (%zu)\n", lds); return 1; }

    std::vector<Obj> objs((size_t)S * NOBJ);
    for (auto& o : objs) {
        o.cx = 100 + urand() * 1720; o.cy = 100 + urand() * 880; o.w = 30 + urand() * 120; o.h = 30 + urand() * 120;
        o.vx = (urand() - 0.5) * 8; o.vy = (urand() - 0.5) * 4; o.base = 0.35 + urand() * 0.6; o.cls = (int)(urand() * 6); o.hidden = 0;
    }
    std::vector<double> hx((size_t)S * MD * 4), hs((size_t)S * MD);
    std::vector<int> hc((size_t)S * MD), hn(S);
    std::vector<unsigned char> hout;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    double tsum = 0; int tcnt = 0;
    unsigned long long checksum = 1469598103934665603ull;
    long tot_tracked = 0, tot_lost = 0;
    std::vector<unsigned char> hstream(d.stream_bytes);
    for (int f = 0; f < F; ++f) {
        for (int s = 0; s < S; ++s) {
            int n = 0;
            for (int k = 0; k < NOBJ; ++k) {
                Obj& o = objs[(size_t)s * NOBJ + k];
                o.cx += o.vx + (urand() - 0.5) * 2; o.cy += o.vy + (urand() - 0.5) * 2;
                if (o.cx < 50 || o.cx > 1870) o.vx = -o.vx;
                if (o.cy < 50 || o.cy > 1030) o.vy = -o.vy;
                if (o.hidden > 0) { o.hidden--; continue; }
                if (urand() < 0.02) { o.hidden = 5 + (int)(urand() * 60); continue; }
                double sc = o.base + (urand() - 0.5) * 0.2;
                if (sc < 0.05) sc = 0.05;
                if (sc > 0.99) sc = 0.99;
                if (n >= MD) break;
                size_t q = (size_t)s * MD + n;
                hx[q * 4 + 0] = o.cx - o.w / 2; hx[q * 4 + 1] = o.cy - o.h / 2; hx[q * 4 + 2] = o.cx + o.w / 2; hx[q * 4 + 3] = o.cy + o.h / 2;
                hs[q] = sc; hc[q] = (urand() < 0.9) ? o.cls : (int)(urand() * 6);
                ++n;
            }
            hn[s] = n;
        }
        hipMemcpy(dx, hx.data(), hx.size() * 8, hipMemcpyHostToDevice);
        hipMemcpy(ds, hs.data(), hs.size() * 8, hipMemcpyHostToDevice);
        hipMemcpy(dc, hc.data(), hc.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dn, hn.data(), hn.size() * 4, hipMemcpyHostToDevice);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(bt_kernel, dim3(S), dim3(256), lds, 0, d);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        if (hipGetLastError() != hipSuccess) { printf("launch failed\n"); return 1; }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (f >= F / 2) { tsum += ms; tcnt++; }
        for (int s = 0; s < S; ++s) {
            hipMemcpy(hstream.data(), arena + (size_t)s * d.stream_bytes, d.stream_bytes, hipMemcpyDeviceToHost);
            BtStream V = bt_view(hstream.data(), MT, MD);
            int a = V.hdr->n_tracked, b = V.hdr->n_lost;
            if (f == F - 1) { tot_tracked += a; tot_lost += b; }
            const unsigned char* p = (const unsigned char*)V.out;
            for (size_t i = 0; i < (size_t)(a + b) * sizeof(BtOut); ++i) { checksum ^= p[i]; checksum *= 1099511628211ull; }
            const unsigned char* ph = (const unsigned char*)V.hdr;
            for (size_t i = 0; i < 20; ++i) { checksum ^= ph[i]; checksum *= 1099511628211ull; }
        }
    }
    printf("streams %d objects %d frames %d MT %d MD %d lds %zu : avg %.1f us/update  tracked/stream %.1f lost/stream %.1f  checksum %016llx\n", S, NOBJ, F,
           MT, MD, lds, tsum / tcnt * 1000.0, (double)tot_tracked / S, (double)tot_lost / S, checksum);
#ifdef ADAS_BT_PROF
    unsigned long long prof[32];
    hipMemcpyFromSymbol(prof, HIP_SYMBOL(g_bt_prof), sizeof(prof));
    static const char* names[] = {"prep", "predict", "cost1", "lap1", "post1", "apply1", "cost2", "lap2", "apply2", "lostnew", "cost3", "lap3", "apply3", "algebra", "dup", "final", "out"};
    unsigned long long tot = 0;
    for (int i = 0; i < 17; ++i) tot += prof[i];
    for (int i = 0; i < 17; ++i) printf("  %-8s %6.2f %%  %8.2f us/update\n", names[i], 100.0 * prof[i] / tot, (double)prof[i] / 100.0 / ((double)F * S));
#endif
    return 0;
}
