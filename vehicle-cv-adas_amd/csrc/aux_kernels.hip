// aux_kernels.hip -- the non-GEMM ops of the detector / lane graphs, all HBM-bound streaming kernels:
//   input_nchw   NCHW fp32 (the coreEngine.py seam layout) -> NHWC compute type, channels padded to 8
//   maxpool      k x k / stride s over a channel-sliced NHWC view (ResNet stem 3x3 s2, SPPF 5x5 s1, YOLOv7 MP 2x2 s2 / SP 5, 9, 13)
//   upsample2    nearest x2, written straight into the consumer's concat slice
//   detect_v8    DFL softmax-expectation + dist2bbox + sigmoid -> (N, 4+nc, A)   (yoloDetector.py:110-122 layout)
//   detect_v5    sigmoid + grid/anchor decode -> (N, A, 5+nc)                    (yoloDetector.py:23,111)
//   layernorm    model_culane.py:34 (fc_norm), one workgroup per frame
//   nhwc_to_nchw parity tap
#include "kernels.h"
#include "elem16.h"

namespace adas {

__device__ __forceinline__ float a_bf2f(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }
__device__ __forceinline__ uint16_t a_f2bf(float f) {
    uint32_t u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
template <typename T> __device__ __forceinline__ float ld(const T* p);
template <> __device__ __forceinline__ float ld<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld<uint16_t>(const uint16_t* p) { return a_bf2f(*p); }
template <> __device__ __forceinline__ float ld<f16s>(const f16s* p) { return Fp16::to_f32(p->v); }
template <typename T> __device__ __forceinline__ void st(T* p, float v);
template <> __device__ __forceinline__ void st<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void st<uint16_t>(uint16_t* p, float v) { *p = a_f2bf(v); }
template <> __device__ __forceinline__ void st<f16s>(f16s* p, float v) { p->v = Fp16::from_f32(v); }
// split precision (elem16.h x3s): a channel slot of the G8 layout; the group follows from the address
template <> __device__ __forceinline__ float ld<x3s>(const x3s* p) { return x3_ld(p); }
template <> __device__ __forceinline__ void st<x3s>(x3s* p, float v) { x3_st(p, v); }

// launch `KERNEL<T>` with T = the storage type of an engine precision
#define ADAS_DISPATCH_STORAGE(prec, T, ...) \
    do {                                    \
        if ((prec) == PREC_FP32) {          \
            using T = float;                \
            __VA_ARGS__;                    \
        } else if ((prec) == PREC_X3) {     \
            using T = x3s;                  \
            __VA_ARGS__;                    \
        } else if ((prec) == PREC_FP16) {   \
            using T = f16s;                 \
            __VA_ARGS__;                    \
        } else {                            \
            using T = uint16_t;             \
            __VA_ARGS__;                    \
        }                                   \
    } while (0)

// ------------------------------------------------------------------------------------- input
template <typename T>
__global__ void input_nchw_kernel(const float* __restrict__ src, T* __restrict__ dst, int n, int c_true, int hw, int cs) {
    size_t total = (size_t)n * hw;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        size_t b = i / hw, p = i - b * hw;
        T* o = dst + i * cs;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float v = (c < c_true) ? src[(b * c_true + c) * hw + p] : 0.0f;
            st<T>(o + c, v);
        }
    }
}
// split precision: the pixel's 8 channel slots are one G8 group -- 16 bytes of hi + 16 bytes of lo, two stores
__global__ void input_nchw_x3_kernel(const float* __restrict__ src, x3s* __restrict__ dst, int n, int c_true, int hw, int cs) {
    Fp16::enter();
    size_t total = (size_t)n * hw;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        size_t b = i / hw, p = i - b * hw;
        float v[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = (c < c_true) ? src[(b * c_true + c) * hw + p] : 0.0f;
        x3_store8(dst + i * cs, v);
    }
}
hipError_t launch_input_nchw(const float* nchw, TView out, int n, int c_true, int prec, hipStream_t st_) {
    int hw = out.h * out.w;
    size_t total = (size_t)n * hw;
    int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    if (out.c != 8 || out.coff != 0 || c_true > 8) return hipErrorInvalidValue;
    if (prec == PREC_X3) {
        if (out.cs & 7) return hipErrorInvalidValue;
        hipLaunchKernelGGL(input_nchw_x3_kernel, dim3(blocks), dim3(256), 0, st_, nchw, (x3s*)out.p, n, c_true, hw, out.cs);
        return hipGetLastError();
    }
    ADAS_DISPATCH_STORAGE(prec, T, hipLaunchKernelGGL(input_nchw_kernel<T>, dim3(blocks), dim3(256), 0, st_, nchw, (T*)out.p, n, c_true, hw, out.cs));
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------- maxpool / upsample
struct PoolDev {
    const void* in;
    void* out;
    int in_cs, in_coff, out_cs, out_coff, c, H, W, Ho, Wo, k, s, p, n;
};
// thread = (pixel, 8-channel group); padding behaves as -inf (torch.nn.MaxPool2d).  bf16: one 16-byte load per tap and
// one 16-byte store (views are 8-channel aligned); the max is taken on the exact f32 images of the bf16 values.
__device__ __forceinline__ void max8(float m[8], const uint16_t* ip) {
    const uint4 q = *reinterpret_cast<const uint4*>(ip);
    const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        m[2 * k] = fmaxf(m[2 * k], __uint_as_float(w[k] << 16));
        m[2 * k + 1] = fmaxf(m[2 * k + 1], __uint_as_float(w[k] & 0xffff0000u));
    }
}
__device__ __forceinline__ void max8(float m[8], const f16s* ip) {
    const uint4 q = *reinterpret_cast<const uint4*>(ip);
    const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        m[2 * k] = fmaxf(m[2 * k], Fp16::lo(w[k]));
        m[2 * k + 1] = fmaxf(m[2 * k + 1], Fp16::hi(w[k]));
    }
}
__device__ __forceinline__ void put8(f16s* op, const float m[8]) {   // the maxima are half values: the conversion is exact
    *reinterpret_cast<uint4*>(op) = make_uint4(Fp16::pack2(m[0], m[1]), Fp16::pack2(m[2], m[3]), Fp16::pack2(m[4], m[5]), Fp16::pack2(m[6], m[7]));
}
__device__ __forceinline__ void max8(float m[8], const float* ip) {
#pragma unroll
    for (int q = 0; q < 8; ++q) m[q] = fmaxf(m[q], ip[q]);
}
__device__ __forceinline__ void put8(uint16_t* op, const float m[8]) {
    uint4 q;  // the maxima are bf16 values: truncation is exact
    q.x = (__float_as_uint(m[0]) >> 16) | (__float_as_uint(m[1]) & 0xffff0000u);
    q.y = (__float_as_uint(m[2]) >> 16) | (__float_as_uint(m[3]) & 0xffff0000u);
    q.z = (__float_as_uint(m[4]) >> 16) | (__float_as_uint(m[5]) & 0xffff0000u);
    q.w = (__float_as_uint(m[6]) >> 16) | (__float_as_uint(m[7]) & 0xffff0000u);
    *reinterpret_cast<uint4*>(op) = q;
}
__device__ __forceinline__ void put8(float* op, const float m[8]) {
#pragma unroll
    for (int q = 0; q < 8; ++q) op[q] = m[q];
}
__device__ __forceinline__ void max8(float m[8], const x3s* ip) {   // the maximum of the joined values (hi alone can tie)
    float v[8];
    x3_load8(ip, v);
#pragma unroll
    for (int q = 0; q < 8; ++q) m[q] = fmaxf(m[q], v[q]);
}
__device__ __forceinline__ void put8(x3s* op, const float m[8]) { x3_store8(op, m); }
template <typename T>
__global__ void maxpool_kernel(PoolDev d) {
    const int c8n = d.c >> 3;
    size_t total = (size_t)d.n * d.Ho * d.Wo * c8n;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        int c8 = (int)(i % c8n);
        size_t pix = i / c8n;
        int ox = (int)(pix % d.Wo);
        size_t t = pix / d.Wo;
        int oy = (int)(t % d.Ho), b = (int)(t / d.Ho);
        float m[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) m[q] = -3.0e38f;
        for (int r = 0; r < d.k; ++r) {
            int iy = oy * d.s - d.p + r;
            if ((unsigned)iy >= (unsigned)d.H) continue;
            for (int s = 0; s < d.k; ++s) {
                int ix = ox * d.s - d.p + s;
                if ((unsigned)ix >= (unsigned)d.W) continue;
                max8(m, (const T*)d.in + ((size_t)(b * d.H + iy) * d.W + ix) * d.in_cs + d.in_coff + c8 * 8);
            }
        }
        put8((T*)d.out + pix * d.out_cs + d.out_coff + c8 * 8, m);
    }
}
// 16-bit elements, compile-time window: every tap is loaded unconditionally from a clamped (always valid) address and out-of-image
// taps are turned into -inf afterwards, so all K*K 16-byte loads of a thread are in flight together (a branch around a
// load makes hipcc wait for it at the join: the generic kernel above pays K*K sequential L2 round trips).
template <typename E, int K>
__global__ void maxpool16_kernel(PoolDev d) {
    typedef typename E::storage T;
    const int c8n = d.c >> 3;
    size_t total = (size_t)d.n * d.Ho * d.Wo * c8n;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        int c8 = (int)(i % c8n);
        size_t pix = i / c8n;
        int ox = (int)(pix % d.Wo);
        size_t t = pix / d.Wo;
        int oy = (int)(t % d.Ho), b = (int)(t / d.Ho);
        const uint16_t* base = (const uint16_t*)d.in + (size_t)b * d.H * d.W * d.in_cs + d.in_coff + c8 * 8;
        uint4 v[K * K];
#pragma unroll
        for (int r = 0; r < K; ++r) {
            const int iy = oy * d.s - d.p + r, iyc = iy < 0 ? 0 : (iy >= d.H ? d.H - 1 : iy);
#pragma unroll
            for (int q = 0; q < K; ++q) {
                const int ix = ox * d.s - d.p + q, ixc = ix < 0 ? 0 : (ix >= d.W ? d.W - 1 : ix);
                v[r * K + q] = *reinterpret_cast<const uint4*>(base + ((size_t)iyc * d.W + ixc) * d.in_cs);
            }
        }
        float m[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) m[q] = -3.0e38f;
#pragma unroll
        for (int r = 0; r < K; ++r) {
            const int iy = oy * d.s - d.p + r;
#pragma unroll
            for (int q = 0; q < K; ++q) {
                const int ix = ox * d.s - d.p + q;
                const bool in = (unsigned)iy < (unsigned)d.H && (unsigned)ix < (unsigned)d.W;
                const uint4 u = v[r * K + q];
                const uint32_t w[4] = {in ? u.x : E::kNegInf2, in ? u.y : E::kNegInf2, in ? u.z : E::kNegInf2, in ? u.w : E::kNegInf2};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    m[2 * k] = fmaxf(m[2 * k], E::lo(w[k]));
                    m[2 * k + 1] = fmaxf(m[2 * k + 1], E::hi(w[k]));
                }
            }
        }
        put8((T*)d.out + pix * d.out_cs + d.out_coff + c8 * 8, m);
    }
}

// split precision, compile-time window: like maxpool16_kernel every tap's group (16 B hi + 16 B lo) is loaded unconditionally from a
// clamped address and out-of-image taps are masked afterwards, so the K*K load pairs of a thread are in flight together; the maximum is
// taken on the joined values.  K = 5 (SPPF, 100 registers of loads) stays on the generic kernel.
template <int K>
__global__ void maxpool_x3_kernel(PoolDev d) {
    Fp16::enter();
    const int c8n = d.c >> 3;
    size_t total = (size_t)d.n * d.Ho * d.Wo * c8n;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        int c8 = (int)(i % c8n);
        size_t pix = i / c8n;
        int ox = (int)(pix % d.Wo);
        size_t t = pix / d.Wo;
        int oy = (int)(t % d.Ho), b = (int)(t / d.Ho);
        const x3s* base = (const x3s*)d.in + (size_t)b * d.H * d.W * d.in_cs + d.in_coff + c8 * 8;
        uint4 vh[K * K], vl[K * K];
#pragma unroll
        for (int r = 0; r < K; ++r) {
            const int iy = oy * d.s - d.p + r, iyc = iy < 0 ? 0 : (iy >= d.H ? d.H - 1 : iy);
#pragma unroll
            for (int q = 0; q < K; ++q) {
                const int ix = ox * d.s - d.p + q, ixc = ix < 0 ? 0 : (ix >= d.W ? d.W - 1 : ix);
                const uint4* g = reinterpret_cast<const uint4*>(base + ((size_t)iyc * d.W + ixc) * d.in_cs);
                vh[r * K + q] = g[0];
                vl[r * K + q] = g[1];
            }
        }
        float m[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) m[q] = -3.0e38f;
#pragma unroll
        for (int r = 0; r < K; ++r) {
            const int iy = oy * d.s - d.p + r;
#pragma unroll
            for (int q = 0; q < K; ++q) {
                const int ix = ox * d.s - d.p + q;
                const bool in = (unsigned)iy < (unsigned)d.H && (unsigned)ix < (unsigned)d.W;
                const uint32_t hw[4] = {vh[r * K + q].x, vh[r * K + q].y, vh[r * K + q].z, vh[r * K + q].w};
                const uint32_t lw[4] = {vl[r * K + q].x, vl[r * K + q].y, vl[r * K + q].z, vl[r * K + q].w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const e_f16x2 hh = __builtin_bit_cast(e_f16x2, hw[k]), ll = __builtin_bit_cast(e_f16x2, lw[k]);
                    const float v0 = x3_join(hh[0], ll[0]), v1 = x3_join(hh[1], ll[1]);
                    m[2 * k] = fmaxf(m[2 * k], in ? v0 : -3.0e38f);
                    m[2 * k + 1] = fmaxf(m[2 * k + 1], in ? v1 : -3.0e38f);
                }
            }
        }
        x3_store8((x3s*)d.out + pix * d.out_cs + d.out_coff + c8 * 8, m);
    }
}

hipError_t launch_maxpool(TView in, TView out, int n, int k, int s, int p, int prec, hipStream_t st_) {
    if (in.c != out.c || (in.c & 7) || ((in.cs | in.coff | out.cs | out.coff) & 7)) return hipErrorInvalidValue;
    PoolDev d{in.p, out.p, in.cs, in.coff, out.cs, out.coff, in.c, in.h, in.w, out.h, out.w, k, s, p, n};
    size_t total = (size_t)n * out.h * out.w * (in.c >> 3);
    int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    if (prec == PREC_X3 && (k == 3 || k == 2)) {
        if (k == 3) hipLaunchKernelGGL((maxpool_x3_kernel<3>), dim3(blocks), dim3(256), 0, st_, d);
        else hipLaunchKernelGGL((maxpool_x3_kernel<2>), dim3(blocks), dim3(256), 0, st_, d);
    } else if (!prec_is16(prec) || (k != 5 && k != 3 && k != 2)) {
        ADAS_DISPATCH_STORAGE(prec, T, hipLaunchKernelGGL(maxpool_kernel<T>, dim3(blocks), dim3(256), 0, st_, d));
    } else {
        ADAS_DISPATCH_E16(prec == PREC_FP16, E, {
            if (k == 5) hipLaunchKernelGGL((maxpool16_kernel<E, 5>), dim3(blocks), dim3(256), 0, st_, d);
            else if (k == 2) hipLaunchKernelGGL((maxpool16_kernel<E, 2>), dim3(blocks), dim3(256), 0, st_, d);   // YOLOv7's MP
            else hipLaunchKernelGGL((maxpool16_kernel<E, 3>), dim3(blocks), dim3(256), 0, st_, d);
        });
    }
    return hipGetLastError();
}

// F.avg_pool2d(x, k, s, p, ceil_mode=False, count_include_pad=True): YOLOv9's AConv / ADown (2x2, stride 1, no padding).  thread = (output
// pixel, 8 channels); the sum runs over the window in row-major order in fp32 and is divided by k*k (padding counts as zeros).
template <typename T>
__global__ void avgpool_kernel(PoolDev d) {
    const int c8n = d.c >> 3;
    const size_t total = (size_t)d.n * d.Ho * d.Wo * c8n;
    const float inv = 1.0f / (float)(d.k * d.k);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % c8n);
        const size_t pix = i / c8n;
        const int ox = (int)(pix % d.Wo);
        const size_t t = pix / d.Wo;
        const int oy = (int)(t % d.Ho);
        const size_t b = t / d.Ho;
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < d.k; ++r) {
            const int iy = oy * d.s - d.p + r;
            if ((unsigned)iy >= (unsigned)d.H) continue;
            for (int q = 0; q < d.k; ++q) {
                const int ix = ox * d.s - d.p + q;
                if ((unsigned)ix >= (unsigned)d.W) continue;
                const T* ip = (const T*)d.in + ((b * d.H + iy) * d.W + ix) * d.in_cs + d.in_coff + c8 * 8;
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += ld<T>(ip + e);
            }
        }
        T* op = (T*)d.out + pix * d.out_cs + d.out_coff + c8 * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) st<T>(op + e, acc[e] * inv);
    }
}
hipError_t launch_avgpool(TView in, TView out, int n, int k, int s, int p, int prec, hipStream_t st_) {
    if (in.c != out.c || (in.c & 7) || ((in.cs | in.coff | out.cs | out.coff) & 7) || k < 1 || k > 7) return hipErrorInvalidValue;
    if (out.h != (in.h + 2 * p - k) / s + 1 || out.w != (in.w + 2 * p - k) / s + 1) return hipErrorInvalidValue;
    PoolDev d{in.p, out.p, in.cs, in.coff, out.cs, out.coff, in.c, in.h, in.w, out.h, out.w, k, s, p, n};
    const size_t total = (size_t)n * out.h * out.w * (in.c >> 3);
    const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    ADAS_DISPATCH_STORAGE(prec, T, hipLaunchKernelGGL(avgpool_kernel<T>, dim3(blocks), dim3(256), 0, st_, d));
    return hipGetLastError();
}

// SPPF (ultralytics SPPF, YOLOv5 / v8 model.9): three chained 5x5 stride-1 pad-2 max-pools whose outputs are concatenated.  One
// workgroup holds an 8-channel slab of one frame's map in LDS and produces all three (row maximum, then column maximum, per pool:
// out-of-image taps are skipped, which is max-pooling's -inf padding); three launches of 25 loads per output become one of one.
// Maximum is exact, so the values are the separate launches' values.
struct Pool3Dev {
    const void* in;
    void* out[3];
    int in_cs, in_coff, out_cs[3], out_coff[3];
    int c, H, W, n;
};
template <typename E>
__global__ __launch_bounds__(256) void sppf_pool3_kernel(Pool3Dev d) {
    extern __shared__ __attribute__((aligned(16))) uint4 pl[];   // [2][H * W]: current map, row maxima
    const int c8n = d.c >> 3, b = blockIdx.x / c8n, c8 = blockIdx.x - b * c8n;
    const int HW = d.H * d.W;
    uint4* cur = pl;
    uint4* row = pl + HW;
    const uint16_t* ip = (const uint16_t*)d.in + (size_t)b * HW * d.in_cs + d.in_coff + c8 * 8;
    for (int p = threadIdx.x; p < HW; p += blockDim.x) cur[p] = *reinterpret_cast<const uint4*>(ip + (size_t)p * d.in_cs);
    __syncthreads();
    auto vmax = [](uint4 a, uint4 b) { return uint4{E::max2(a.x, b.x), E::max2(a.y, b.y), E::max2(a.z, b.z), E::max2(a.w, b.w)}; };
    for (int k = 0; k < 3; ++k) {
        for (int p = threadIdx.x; p < HW; p += blockDim.x) {
            const int y = p / d.W, x = p - y * d.W;
            uint4 m = cur[p];
#pragma unroll
            for (int dx = -2; dx <= 2; ++dx)
                if (dx != 0 && (unsigned)(x + dx) < (unsigned)d.W) m = vmax(m, cur[p + dx]);
            row[p] = m;
        }
        __syncthreads();
        uint16_t* op = (uint16_t*)d.out[k] + (size_t)b * HW * d.out_cs[k] + d.out_coff[k] + c8 * 8;
        for (int p = threadIdx.x; p < HW; p += blockDim.x) {
            const int y = p / d.W;
            uint4 m = row[p];
#pragma unroll
            for (int dy = -2; dy <= 2; ++dy)
                if (dy != 0 && (unsigned)(y + dy) < (unsigned)d.H) m = vmax(m, row[p + dy * d.W]);
            *reinterpret_cast<uint4*>(op + (size_t)p * d.out_cs[k]) = m;
            cur[p] = m;   // the next pool's input (cur is only read by the row pass, which has finished)
        }
        __syncthreads();
    }
}
bool sppf_pool3_applicable(int prec, const TView& in, const TView out[3]) {
    if (!prec_is16(prec) || in.f32 || (in.c & 7) || ((in.cs | in.coff) & 7)) return false;
    for (int k = 0; k < 3; ++k)
        if (out[k].f32 || out[k].c != in.c || out[k].h != in.h || out[k].w != in.w || ((out[k].cs | out[k].coff) & 7)) return false;
    return (size_t)in.h * in.w * 32 <= 96 * 1024;   // two 16-byte planes of the map in LDS
}
hipError_t launch_sppf_pool3(const TView& in, const TView out[3], int n, int prec, hipStream_t st_) {
    if (!sppf_pool3_applicable(prec, in, out)) return hipErrorNotSupported;
    Pool3Dev d;
    d.in = in.p; d.in_cs = in.cs; d.in_coff = in.coff; d.c = in.c; d.H = in.h; d.W = in.w; d.n = n;
    for (int k = 0; k < 3; ++k) { d.out[k] = out[k].p; d.out_cs[k] = out[k].cs; d.out_coff[k] = out[k].coff; }
    const size_t lds = (size_t)in.h * in.w * 32;
    const dim3 grid((unsigned)(n * (in.c >> 3)));
    ADAS_DISPATCH_E16(prec == PREC_FP16, E, {
        static bool attr_done = false;
        if (!attr_done) {
            (void)hipFuncSetAttribute((const void*)sppf_pool3_kernel<E>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
            attr_done = true;
        }
        hipLaunchKernelGGL((sppf_pool3_kernel<E>), grid, dim3(256), lds, st_, d);
    });
    return hipGetLastError();
}

template <typename T>
__global__ void upsample2_kernel(PoolDev d) {
    const int c8n = d.c >> 3;
    size_t total = (size_t)d.n * d.Ho * d.Wo * c8n;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        int c8 = (int)(i % c8n);
        size_t pix = i / c8n;
        int ox = (int)(pix % d.Wo);
        size_t t = pix / d.Wo;
        int oy = (int)(t % d.Ho), b = (int)(t / d.Ho);
        const T* ip = (const T*)d.in + ((size_t)(b * d.H + (oy >> 1)) * d.W + (ox >> 1)) * d.in_cs + d.in_coff + c8 * 8;
        T* op = (T*)d.out + pix * d.out_cs + d.out_coff + c8 * 8;
        if (sizeof(T) == 2) {  // 8 bf16 = one 16-byte move (views are 8-channel aligned)
            *reinterpret_cast<uint4*>(op) = *reinterpret_cast<const uint4*>(ip);
        } else {               // 8 four-byte slots (fp32, or one G8 group of the split precision): two 16-byte moves
            const uint4 q0 = reinterpret_cast<const uint4*>(ip)[0], q1 = reinterpret_cast<const uint4*>(ip)[1];
            reinterpret_cast<uint4*>(op)[0] = q0;
            reinterpret_cast<uint4*>(op)[1] = q1;
        }
    }
}
hipError_t launch_upsample2(TView in, TView out, int n, int prec, hipStream_t st_) {
    if (in.c != out.c || (in.c & 7) || ((in.cs | in.coff | out.cs | out.coff) & 7) || out.h != 2 * in.h || out.w != 2 * in.w) return hipErrorInvalidValue;
    PoolDev d{in.p, out.p, in.cs, in.coff, out.cs, out.coff, in.c, in.h, in.w, out.h, out.w, 0, 0, 0, n};
    size_t total = (size_t)n * out.h * out.w * (in.c >> 3);
    int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    if (!prec_is16(prec))   // fp32, or the split precision: a G8 group of 8 channels is 32 bytes moved as a whole
        hipLaunchKernelGGL(upsample2_kernel<float>, dim3(blocks), dim3(256), 0, st_, d);
    else
        hipLaunchKernelGGL(upsample2_kernel<uint16_t>, dim3(blocks), dim3(256), 0, st_, d);
    return hipGetLastError();
}

// depth-to-space, block 2: out[2y + dy][2x + dx][c] = in[y][x][(2 dy + dx) * C + c].  With a 1x1 conv to 4 C channels in front (rows
// ordered (dy, dx, c)) this is ConvTranspose2d(kernel 2, stride 2): YOLOv6's BiFusion up-sampling (yolov6/layers/common.py Transpose).
template <typename T>
__global__ void depth2space_kernel(PoolDev d) {
    const int c8n = d.c >> 3;      // d.c: OUTPUT channels C
    size_t total = (size_t)d.n * d.Ho * d.Wo * c8n;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        int c8 = (int)(i % c8n);
        size_t pix = i / c8n;
        int ox = (int)(pix % d.Wo);
        size_t t = pix / d.Wo;
        int oy = (int)(t % d.Ho), b = (int)(t / d.Ho);
        const int q = ((oy & 1) << 1) | (ox & 1);
        const T* ip = (const T*)d.in + ((size_t)(b * d.H + (oy >> 1)) * d.W + (ox >> 1)) * d.in_cs + d.in_coff + q * d.c + c8 * 8;
        T* op = (T*)d.out + pix * d.out_cs + d.out_coff + c8 * 8;
        if (sizeof(T) == 2) {
            *reinterpret_cast<uint4*>(op) = *reinterpret_cast<const uint4*>(ip);
        } else {
            const uint4 q0 = reinterpret_cast<const uint4*>(ip)[0], q1 = reinterpret_cast<const uint4*>(ip)[1];
            reinterpret_cast<uint4*>(op)[0] = q0;
            reinterpret_cast<uint4*>(op)[1] = q1;
        }
    }
}
bool depth2space_supported(const TView& in, const TView& out) {
    return in.c == 4 * out.c && (out.c & 7) == 0 && ((in.cs | in.coff | out.cs | out.coff) & 7) == 0 && out.h == 2 * in.h && out.w == 2 * in.w &&
           !in.f32 && !out.f32;
}
hipError_t launch_depth2space(TView in, TView out, int n, int prec, hipStream_t st_) {
    if (!depth2space_supported(in, out)) return hipErrorInvalidValue;
    PoolDev d{in.p, out.p, in.cs, in.coff, out.cs, out.coff, out.c, in.h, in.w, out.h, out.w, 0, 0, 0, n};
    size_t total = (size_t)n * out.h * out.w * (out.c >> 3);
    int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    if (!prec_is16(prec))   // (the split precision's 32-byte groups move as wholes)
        hipLaunchKernelGGL(depth2space_kernel<float>, dim3(blocks), dim3(256), 0, st_, d);
    else
        hipLaunchKernelGGL(depth2space_kernel<uint16_t>, dim3(blocks), dim3(256), 0, st_, d);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------- Detect (v8)
struct DetV8Dev {
    const float* box[3];
    const float* cls[3];
    int box_cs[3], cls_cs[3];
    int hw[3], w[3], stride[3], a_off[3];
    float* out;
    int nc, A, n;
};
// workgroup = 64 consecutive anchors of one level of one frame.
__global__ __launch_bounds__(256) void detect_v8_kernel(DetV8Dev d) {
    __shared__ float s_box[64][65];
    __shared__ float s_dist[64][4];
    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    // find level
    int blk = blockIdx.x, lvl = 0;
    for (; lvl < 3; ++lvl) {
        int nb = (d.hw[lvl] + 63) / 64;
        if (blk < nb) break;
        blk -= nb;
    }
    const int p0 = blk * 64;
    const int np = min(64, d.hw[lvl] - p0);
    const float* box = d.box[lvl] + ((size_t)b * d.hw[lvl] + p0) * d.box_cs[lvl];
    const float* cls = d.cls[lvl] + ((size_t)b * d.hw[lvl] + p0) * d.cls_cs[lvl];
    for (int i = tid; i < np * 64; i += 256) s_box[i >> 6][i & 63] = box[(size_t)(i >> 6) * d.box_cs[lvl] + (i & 63)];
    __syncthreads();
    {   // DFL: softmax over 16 bins, expectation with arange(16)
        int p = tid >> 2, side = tid & 3;
        if (p < np) {
            const float* v = &s_box[p][side * 16];
            float mx = v[0];
#pragma unroll
            for (int k = 1; k < 16; ++k) mx = fmaxf(mx, v[k]);
            float se = 0.f, sw = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                float e = expf(v[k] - mx);
                se += e;
                sw += e * (float)k;
            }
            s_dist[p][side] = sw / se;
        }
    }
    __syncthreads();
    float* out = d.out + (size_t)b * (4 + d.nc) * d.A + d.a_off[lvl] + p0;
    if (tid < 64 && tid < np) {
        int p = p0 + tid;
        float ax = (float)(p % d.w[lvl]) + 0.5f, ay = (float)(p / d.w[lvl]) + 0.5f;
        float x1 = ax - s_dist[tid][0], y1 = ay - s_dist[tid][1];
        float x2 = ax + s_dist[tid][2], y2 = ay + s_dist[tid][3];
        float s = (float)d.stride[lvl];
        out[(size_t)0 * d.A + tid] = (x1 + x2) / 2 * s;
        out[(size_t)1 * d.A + tid] = (y1 + y2) / 2 * s;
        out[(size_t)2 * d.A + tid] = (x2 - x1) * s;
        out[(size_t)3 * d.A + tid] = (y2 - y1) * s;
    }
    // class probabilities: out[(4+c)*A + anchor] = sigmoid(cls[anchor][c]); stage through LDS for coalesced stores
    for (int c0 = 0; c0 < d.nc; c0 += 64) {
        __syncthreads();
        int cw = min(64, d.nc - c0);
        for (int i = tid; i < np * cw; i += 256) {
            int p = i / cw, c = i - p * cw;
            float v = cls[(size_t)p * d.cls_cs[lvl] + c0 + c];
            s_box[c][p] = 1.0f / (1.0f + expf(-v));
        }
        __syncthreads();
        for (int i = tid; i < cw * 64; i += 256) {
            int c = i >> 6, p = i & 63;
            if (p < np) out[(size_t)(4 + c0 + c) * d.A + p] = s_box[c][p];
        }
    }
}
hipError_t launch_detect_v8(const TView* ins, float* out, int n, int nc, int A, const int strides[3], hipStream_t st_) {
    DetV8Dev d;
    int off = 0, blocks = 0;
    for (int l = 0; l < 3; ++l) {
        const TView& b = ins[2 * l];
        const TView& c = ins[2 * l + 1];
        if (!b.f32 || !c.f32 || b.c != 64 || c.c != nc || b.coff || c.coff) return hipErrorInvalidValue;
        d.box[l] = (const float*)b.p; d.cls[l] = (const float*)c.p;
        d.box_cs[l] = b.cs; d.cls_cs[l] = c.cs;
        d.hw[l] = b.h * b.w; d.w[l] = b.w; d.stride[l] = strides[l]; d.a_off[l] = off;
        off += d.hw[l];
        blocks += (d.hw[l] + 63) / 64;
    }
    if (off != A) return hipErrorInvalidValue;
    d.out = out; d.nc = nc; d.A = A; d.n = n;
    hipLaunchKernelGGL(detect_v8_kernel, dim3(blocks, n), dim3(256), 0, st_, d);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------- Detect (v8), fused with its last 1x1 convs
// model.22.cv2.i.2 (cb -> 64 DFL logits) and model.22.cv3.i.2 (cc -> nc class logits) feed nothing but the decode: run here as
// MFMA GEMMs (fragment-ordered weights in LDS, activations straight from HBM into B registers, as in conv_pw.hip) they keep
// 310 MB of fp32 logits per 64-frame step out of HBM and save six launches.  A workgroup owns 64 anchors of one level of one
// frame; wave w owns anchors 16w..16w+15.  With weights as the A operand a lane ends with 4 consecutive outputs of one
// anchor, so the 16 DFL bins of a box side sit in the 4 lanes {lrow, lrow+16, lrow+32, lrow+48} x 4 registers.
typedef __attribute__((ext_vector_type(4))) float df32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t du32x4;
#define ADAS_DETF_MAXKS 12  // hidden widths up to 384 channels (YOLOv8x: 320)

struct DetFuseDev {
    const uint16_t* hb[3];  // inputs of cv2.i.2 (bf16 NHWC views)
    const uint16_t* hc[3];  // inputs of cv3.i.2
    int hb_cs[3], hb_coff[3], hc_cs[3], hc_coff[3];
    const uint16_t* wb[3];  // [4][KSb][64][8] fragment order
    const uint16_t* wc[3];  // [NTc][KSc][64][8]
    const float* bb[3];
    const float* bc[3];
    int cb, cc;             // hidden widths (same on every level)
    int hw[3], w[3], stride[3], a_off[3];
    float* out;
    int nc, A, n;
    float* sink_conf;       // SINK: per-anchor best class probability / first arg-max class ([n][A] each), written instead of the class rows
    int* sink_cls;
};

__device__ __forceinline__ float detf_quad(float v, int m) { return __shfl_xor(v, m, 64); }

// SINK (pipeline steps): the consumer is the post-processing, which reads the four box rows of candidate anchors and, per anchor, the
// best class probability and its first arg-max (yoloDetector.py:120-127) -- what yolo_scan_v8 derives from the 80 class rows.  The
// probabilities are in this kernel's registers: the per-anchor maximum is taken here (same float values, same first-maximum rule) and the
// class rows (181 MB per 64 frames, written here and read back by the scan) never exist.
template <typename E, bool SINK>
__global__ __launch_bounds__(256) void detect_v8_fused_kernel(DetFuseDev d) {
    extern __shared__ __attribute__((aligned(16))) uint16_t wl[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lrow = lane & 15, kg = lane >> 4;
    const int b = blockIdx.y;
    int blk = blockIdx.x, lvl = 0;
    for (; lvl < 2; ++lvl) {
        const int nb = (d.hw[lvl] + 63) / 64;
        if (blk < nb) break;
        blk -= nb;
    }
    const int KSb = (d.cb + 31) >> 5, KSc = (d.cc + 31) >> 5, NTc = (d.nc + 15) >> 4;
    uint16_t* wbox = wl;                                  // [4][KSb][512]
    uint16_t* wcls = wl + (size_t)4 * KSb * 512;          // [NTc][KSc][512]
    float* bias = reinterpret_cast<float*>(wcls + (size_t)NTc * KSc * 512);  // [64 + NTc*16]
    stage_lds16<256, 4>(wbox, d.wb[lvl], 4 * KSb * 64, tid);
    stage_lds16<256, 8>(wcls, d.wc[lvl], NTc * KSc * 64, tid);
    for (int i = tid; i < 64 + NTc * 16; i += 256) bias[i] = i < 64 ? d.bb[lvl][i] : d.bc[lvl][i - 64];
    __syncthreads();

    const int p = blk * 64 + wave * 16 + lrow;
    const bool ok = p < d.hw[lvl];
    const size_t pix = (size_t)b * d.hw[lvl] + (ok ? p : 0);
    float* out = d.out + (size_t)b * (4 + d.nc) * d.A + d.a_off[lvl];

    // ---- box branch: 64 DFL logits = 4 feature tiles (one per box side)
    {
        const uint16_t* ip = d.hb[lvl] + pix * d.hb_cs[lvl] + d.hb_coff[lvl] + kg * 8;
        df32x4 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = df32x4{0.f, 0.f, 0.f, 0.f};
        for (int ks = 0; ks < KSb; ++ks) {
            du32x4 xb = du32x4{0u, 0u, 0u, 0u};
            if (ok && ks * 32 + kg * 8 < d.cb) xb = *reinterpret_cast<const du32x4*>(ip + ks * 32);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const du32x4 wf = *reinterpret_cast<const du32x4*>(wbox + ((size_t)(t * KSb + ks) * 64 + lane) * 8);
                acc[t] = E::mfma(wf, xb, acc[t]);
            }
        }
        float dist[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {  // DFL: softmax over the side's 16 bins, expectation with arange(16)
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[t][r] + bias[t * 16 + kg * 4 + r];
            float mx = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
            mx = fmaxf(mx, detf_quad(mx, 16));
            mx = fmaxf(mx, detf_quad(mx, 32));
            float se = 0.f, sw = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = __expf(v[r] - mx);   // (16-bit modes only: this kernel does not exist in the parity modes)
                se += e;
                sw += e * (float)(kg * 4 + r);
            }
            se += detf_quad(se, 16); sw += detf_quad(sw, 16);
            se += detf_quad(se, 32); sw += detf_quad(sw, 32);
            dist[t] = sw / se;
        }
        if (ok) {
            const float ax = (float)(p % d.w[lvl]) + 0.5f, ay = (float)(p / d.w[lvl]) + 0.5f;
            const float x1 = ax - dist[0], y1 = ay - dist[1], x2 = ax + dist[2], y2 = ay + dist[3];
            const float s = (float)d.stride[lvl];
            const float comp = kg == 0 ? (x1 + x2) / 2 * s : kg == 1 ? (y1 + y2) / 2 * s : kg == 2 ? (x2 - x1) * s : (y2 - y1) * s;
            out[(size_t)kg * d.A + p] = comp;
        }
    }
    // ---- class branch: sigmoid(cv3.i.2)
    {
        const uint16_t* ip = d.hc[lvl] + pix * d.hc_cs[lvl] + d.hc_coff[lvl] + kg * 8;
        du32x4 xc[ADAS_DETF_MAXKS];
#pragma unroll
        for (int ks = 0; ks < ADAS_DETF_MAXKS; ++ks) {
            xc[ks] = du32x4{0u, 0u, 0u, 0u};
            if (ks < KSc && ok && ks * 32 + kg * 8 < d.cc) xc[ks] = *reinterpret_cast<const du32x4*>(ip + ks * 32);
        }
        float bv = 0.f;
        int bi = -1;
        for (int nt = 0; nt < NTc; ++nt) {
            df32x4 acc = df32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < ADAS_DETF_MAXKS; ++ks) {
                if (ks < KSc) {
                    const du32x4 wf = *reinterpret_cast<const du32x4*>(wcls + ((size_t)(nt * KSc + ks) * 64 + lane) * 8);
                    acc = E::mfma(wf, xc[ks], acc);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = nt * 16 + kg * 4 + r;
                if (SINK) {
                    if (c < d.nc) {
                        const float v = fast_rcp(1.0f + __expf(-(acc[r] + bias[64 + c])));   // v_exp + v_rcp (elem16.h fast_rcp): 80 sigmoids per anchor, was expf + IEEE division
                        if (bi < 0 || v > bv) {   // classes ascend within a lane: strict > keeps the first maximum
                            bv = v;
                            bi = c;
                        }
                    }
                } else if (ok && c < d.nc) {
                    out[(size_t)(4 + c) * d.A + p] = fast_rcp(1.0f + __expf(-(acc[r] + bias[64 + c])));   // the same expression as the SINK branch: identical values
                }
            }
        }
        if (SINK) {
            // the four lanes of a pixel (kg = 0..3) hold interleaved class quads: larger value wins, equal values -> smaller class
#pragma unroll
            for (int m = 16; m <= 32; m <<= 1) {
                const float ov = __shfl_xor(bv, m, 64);
                const int oi = __shfl_xor(bi, m, 64);
                if (oi >= 0 && (bi < 0 || ov > bv || (ov == bv && oi < bi))) {
                    bv = ov;
                    bi = oi;
                }
            }
            if (ok && kg == 0) {
                const size_t o = (size_t)b * d.A + d.a_off[lvl] + p;
                d.sink_conf[o] = bv;
                d.sink_cls[o] = bi < 0 ? 0 : bi;
            }
        }
    }
}

// hidden[2l] / hidden[2l+1]: inputs of cv2.l.2 / cv3.l.2; wfrag/bias: their packed (CONV_PW order) weights and biases
hipError_t launch_detect_v8_fused(const TView* hidden, const void* const* wfrag, const float* const* bias, float* out, int n, int nc, int A,
                                  const int strides[3], int prec, hipStream_t st_, float* sink_conf, int* sink_cls) {
    DetFuseDev d;
    int off = 0, blocks = 0;
    d.cb = hidden[0].c; d.cc = hidden[1].c;
    for (int l = 0; l < 3; ++l) {
        const TView& hb = hidden[2 * l];
        const TView& hc = hidden[2 * l + 1];
        if (hb.f32 || hc.f32 || hb.c != d.cb || hc.c != d.cc || hb.h != hc.h || hb.w != hc.w) return hipErrorInvalidValue;
        if (((hb.cs | hb.coff | hc.cs | hc.coff | hb.c | hc.c) & 7) != 0) return hipErrorInvalidValue;
        d.hb[l] = (const uint16_t*)hb.p; d.hc[l] = (const uint16_t*)hc.p;
        d.hb_cs[l] = hb.cs; d.hb_coff[l] = hb.coff; d.hc_cs[l] = hc.cs; d.hc_coff[l] = hc.coff;
        d.wb[l] = (const uint16_t*)wfrag[2 * l]; d.wc[l] = (const uint16_t*)wfrag[2 * l + 1];
        d.bb[l] = bias[2 * l]; d.bc[l] = bias[2 * l + 1];
        d.hw[l] = hb.h * hb.w; d.w[l] = hb.w; d.stride[l] = strides[l]; d.a_off[l] = off;
        off += d.hw[l];
        blocks += (d.hw[l] + 63) / 64;
    }
    if (off != A || (d.cc + 31) / 32 > ADAS_DETF_MAXKS) return hipErrorInvalidValue;
    d.out = out; d.nc = nc; d.A = A; d.n = n;
    d.sink_conf = sink_conf; d.sink_cls = sink_cls;
    const int KSb = (d.cb + 31) / 32, KSc = (d.cc + 31) / 32, NTc = (nc + 15) / 16;
    const size_t lds = ((size_t)4 * KSb + (size_t)NTc * KSc) * 1024 + (64 + (size_t)NTc * 16) * 4;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)detect_v8_fused_kernel<Bf16, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        (void)hipFuncSetAttribute((const void*)detect_v8_fused_kernel<Fp16, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        (void)hipFuncSetAttribute((const void*)detect_v8_fused_kernel<Bf16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        (void)hipFuncSetAttribute((const void*)detect_v8_fused_kernel<Fp16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        attr_done = true;
    }
    if (lds > 150 * 1024) return hipErrorNotSupported;
    if (sink_conf && sink_cls)
        ADAS_DISPATCH_E16(prec == PREC_FP16, E, hipLaunchKernelGGL((detect_v8_fused_kernel<E, true>), dim3(blocks, n), dim3(256), lds, st_, d));
    else
        ADAS_DISPATCH_E16(prec == PREC_FP16, E, hipLaunchKernelGGL((detect_v8_fused_kernel<E, false>), dim3(blocks, n), dim3(256), lds, st_, d));
    return hipGetLastError();
}

// ---- the same fusion in the split precision (ADAS_PREC_FP16X3; round 6).  The six 1x1 convs read G8 activations (a lane's 8 channels:
// 16 bytes of hi halves, 16 bytes of lo halves) and conv_pw_x3's weight packing as it is ([16-feature tile][32-channel K step][hi 1 KB |
// lo 1 KB]); every product is three MFMAs (main += w_hi x_hi; cross += w_lo x_hi + w_hi x_lo; logit = main + 2^-11 cross + bias); DFL
// softmax and sigmoid in their exact forms (expf, IEEE division: detect_v8_kernel's expressions).  The exact mode ran this as six
// conv_pwx3 launches + the decode (0.27 ms per 64 frames); with SINK the pipeline's scan launch and the class rows go as well.
struct DetFuseX3Dev {
    const unsigned char* hb[3];   // inputs of cv2.i.2 (G8 NHWC views: 4 bytes per channel slot)
    const unsigned char* hc[3];   // inputs of cv3.i.2
    int hb_cs[3], hb_coff[3], hc_cs[3], hc_coff[3];
    const uint16_t* wb[3];        // [4][KTb][hi | lo][64][8]
    const uint16_t* wc[3];        // [NTc..][KTc][hi | lo][64][8]
    const float* bb[3];
    const float* bc[3];
    int cb, cc, ktb, ktc;         // hidden widths; K steps of the two packings (kpad / 32)
    int hw[3], w[3], stride[3], a_off[3];
    float* out;
    int nc, A, n;
    float* sink_conf;
    int* sink_cls;
};

template <bool SINK>
__global__ __launch_bounds__(256) void detect_v8_fused_x3_kernel(DetFuseX3Dev d) {
    extern __shared__ __attribute__((aligned(16))) uint16_t wl[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lrow = lane & 15, kg = lane >> 4;
    const int b = blockIdx.y;
    int blk = blockIdx.x, lvl = 0;
    for (; lvl < 2; ++lvl) {
        const int nb = (d.hw[lvl] + 63) / 64;
        if (blk < nb) break;
        blk -= nb;
    }
    const int KTb = d.ktb, KTc = d.ktc, NTc = (d.nc + 15) >> 4;
    uint16_t* wbox = wl;                                   // [4][KTb][2][512]
    uint16_t* wcls = wl + (size_t)4 * KTb * 1024;          // [NTc][KTc][2][512]
    float* bias = reinterpret_cast<float*>(wcls + (size_t)NTc * KTc * 1024);   // [64 + NTc * 16]
    stage_lds16<256, 4>(wbox, d.wb[lvl], 4 * KTb * 128, tid);
    stage_lds16<256, 8>(wcls, d.wc[lvl], NTc * KTc * 128, tid);
    for (int i = tid; i < 64 + NTc * 16; i += 256) bias[i] = i < 64 ? d.bb[lvl][i] : (i - 64 < d.nc ? d.bc[lvl][i - 64] : 0.0f);
    __syncthreads();

    const int p = blk * 64 + wave * 16 + lrow;
    const bool ok = p < d.hw[lvl];
    const size_t pix = (size_t)b * d.hw[lvl] + (ok ? p : 0);
    float* out = d.out + (size_t)b * (4 + d.nc) * d.A + d.a_off[lvl];
    const du32x4 zero4 = du32x4{0u, 0u, 0u, 0u};

    // ---- box branch: 64 DFL logits = 4 feature tiles (one per box side)
    {
        const unsigned char* ip = d.hb[lvl] + (pix * d.hb_cs[lvl] + d.hb_coff[lvl]) * 4 + kg * 32;   // this lane's 8-channel group of K step 0
        df32x4 am[4], ac[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) am[t] = ac[t] = df32x4{0.f, 0.f, 0.f, 0.f};
        for (int ks = 0; ks < KTb; ++ks) {
            du32x4 xh = zero4, xl = zero4;
            if (ok && ks * 32 + kg * 8 < d.cb) {
                xh = *reinterpret_cast<const du32x4*>(ip + ks * 128);
                xl = *reinterpret_cast<const du32x4*>(ip + ks * 128 + 16);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const du32x4 wh = *reinterpret_cast<const du32x4*>(wbox + ((size_t)(t * KTb + ks) * 2 * 64 + lane) * 8);
                const du32x4 wlo = *reinterpret_cast<const du32x4*>(wbox + ((size_t)((t * KTb + ks) * 2 + 1) * 64 + lane) * 8);
                am[t] = Fp16::mfma(wh, xh, am[t]);
                ac[t] = Fp16::mfma(wlo, xh, ac[t]);
                ac[t] = Fp16::mfma(wh, xl, ac[t]);
            }
        }
        float dist[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {  // DFL: softmax over the side's 16 bins (4 lanes x 4 registers), expectation with arange(16)
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = am[t][r] + ac[t][r] * kX3Down + bias[t * 16 + kg * 4 + r];
            float mx = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
            mx = fmaxf(mx, detf_quad(mx, 16));
            mx = fmaxf(mx, detf_quad(mx, 32));
            float se = 0.f, sw = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = expf(v[r] - mx);
                se += e;
                sw += e * (float)(kg * 4 + r);
            }
            se += detf_quad(se, 16); sw += detf_quad(sw, 16);
            se += detf_quad(se, 32); sw += detf_quad(sw, 32);
            dist[t] = sw / se;
        }
        if (ok) {
            const float ax = (float)(p % d.w[lvl]) + 0.5f, ay = (float)(p / d.w[lvl]) + 0.5f;
            const float x1 = ax - dist[0], y1 = ay - dist[1], x2 = ax + dist[2], y2 = ay + dist[3];
            const float s = (float)d.stride[lvl];
            const float comp = kg == 0 ? (x1 + x2) / 2 * s : kg == 1 ? (y1 + y2) / 2 * s : kg == 2 ? (x2 - x1) * s : (y2 - y1) * s;
            out[(size_t)kg * d.A + p] = comp;
        }
    }
    // ---- class branch: sigmoid(cv3.i.2), exact form
    {
        const unsigned char* ip = d.hc[lvl] + (pix * d.hc_cs[lvl] + d.hc_coff[lvl]) * 4 + kg * 32;
        du32x4 xh[ADAS_DETF_MAXKS], xl[ADAS_DETF_MAXKS];
#pragma unroll
        for (int ks = 0; ks < ADAS_DETF_MAXKS; ++ks) {
            xh[ks] = xl[ks] = zero4;
            if (ks < KTc && ok && ks * 32 + kg * 8 < d.cc) {
                xh[ks] = *reinterpret_cast<const du32x4*>(ip + ks * 128);
                xl[ks] = *reinterpret_cast<const du32x4*>(ip + ks * 128 + 16);
            }
        }
        float bv = 0.f;
        int bi = -1;
        for (int nt = 0; nt < NTc; ++nt) {
            df32x4 am = df32x4{0.f, 0.f, 0.f, 0.f}, ac = am;
#pragma unroll
            for (int ks = 0; ks < ADAS_DETF_MAXKS; ++ks) {
                if (ks < KTc) {
                    const du32x4 wh = *reinterpret_cast<const du32x4*>(wcls + ((size_t)(nt * KTc + ks) * 2 * 64 + lane) * 8);
                    const du32x4 wlo = *reinterpret_cast<const du32x4*>(wcls + ((size_t)((nt * KTc + ks) * 2 + 1) * 64 + lane) * 8);
                    am = Fp16::mfma(wh, xh[ks], am);
                    ac = Fp16::mfma(wlo, xh[ks], ac);
                    ac = Fp16::mfma(wh, xl[ks], ac);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = nt * 16 + kg * 4 + r;
                if (c < d.nc) {
                    const float v = 1.0f / (1.0f + expf(-(am[r] + ac[r] * kX3Down + bias[64 + c])));   // detect_v8_kernel's expression
                    if (SINK) {
                        if (bi < 0 || v > bv) {   // classes ascend within a lane: strict > keeps the first maximum
                            bv = v;
                            bi = c;
                        }
                    } else if (ok) {
                        out[(size_t)(4 + c) * d.A + p] = v;
                    }
                }
            }
        }
        if (SINK) {
            // the four lanes of a pixel (kg = 0..3) hold interleaved class quads: larger value wins, equal values -> smaller class
#pragma unroll
            for (int m = 16; m <= 32; m <<= 1) {
                const float ov = __shfl_xor(bv, m, 64);
                const int oi = __shfl_xor(bi, m, 64);
                if (oi >= 0 && (bi < 0 || ov > bv || (ov == bv && oi < bi))) {
                    bv = ov;
                    bi = oi;
                }
            }
            if (ok && kg == 0) {
                const size_t o = (size_t)b * d.A + d.a_off[lvl] + p;
                d.sink_conf[o] = bv;
                d.sink_cls[o] = bi < 0 ? 0 : bi;
            }
        }
    }
}

// hidden[2l] / hidden[2l+1]: inputs of cv2.l.2 / cv3.l.2 (G8 views); wfrag / bias: their conv_pw_x3 packings and biases; kt_box / kt_cls: K steps of
// the two packings (the layers' kpad / 32)
hipError_t launch_detect_v8_fused_x3(const TView* hidden, const void* const* wfrag, const float* const* bias, int kt_box, int kt_cls, float* out, int n, int nc,
                                     int A, const int strides[3], hipStream_t st_, float* sink_conf, int* sink_cls) {
    DetFuseX3Dev d;
    int off = 0, blocks = 0;
    d.cb = hidden[0].c; d.cc = hidden[1].c; d.ktb = kt_box; d.ktc = kt_cls;
    for (int l = 0; l < 3; ++l) {
        const TView& hb = hidden[2 * l];
        const TView& hc = hidden[2 * l + 1];
        if (hb.f32 || hc.f32 || hb.c != d.cb || hc.c != d.cc || hb.h != hc.h || hb.w != hc.w) return hipErrorInvalidValue;
        if (((hb.cs | hb.coff | hc.cs | hc.coff | hb.c | hc.c) & 7) != 0) return hipErrorInvalidValue;
        d.hb[l] = (const unsigned char*)hb.p; d.hc[l] = (const unsigned char*)hc.p;
        d.hb_cs[l] = hb.cs; d.hb_coff[l] = hb.coff; d.hc_cs[l] = hc.cs; d.hc_coff[l] = hc.coff;
        d.wb[l] = (const uint16_t*)wfrag[2 * l]; d.wc[l] = (const uint16_t*)wfrag[2 * l + 1];
        d.bb[l] = bias[2 * l]; d.bc[l] = bias[2 * l + 1];
        d.hw[l] = hb.h * hb.w; d.w[l] = hb.w; d.stride[l] = strides[l]; d.a_off[l] = off;
        off += d.hw[l];
        blocks += (d.hw[l] + 63) / 64;
    }
    if (off != A || kt_cls > ADAS_DETF_MAXKS || kt_box * 32 < d.cb || kt_cls * 32 < d.cc) return hipErrorInvalidValue;
    d.out = out; d.nc = nc; d.A = A; d.n = n;
    d.sink_conf = sink_conf; d.sink_cls = sink_cls;
    const int NTc = (nc + 15) / 16;
    const size_t lds = ((size_t)4 * kt_box + (size_t)NTc * kt_cls) * 2048 + (64 + (size_t)NTc * 16) * 4;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)detect_v8_fused_x3_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        (void)hipFuncSetAttribute((const void*)detect_v8_fused_x3_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        attr_done = true;
    }
    if (lds > 150 * 1024) return hipErrorNotSupported;
    if (sink_conf && sink_cls) hipLaunchKernelGGL(detect_v8_fused_x3_kernel<true>, dim3(blocks, n), dim3(256), lds, st_, d);
    else hipLaunchKernelGGL(detect_v8_fused_x3_kernel<false>, dim3(blocks, n), dim3(256), lds, st_, d);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------- Detect (v5)
struct DetV5Dev {
    const float* in[3];
    int cs[3], ny[3], nx[3], stride[3], row_off[3];
    const float* anchors;
    float* out;
    int nc, A, n;
};
__global__ void detect_v5_kernel(DetV5Dev d) {
    const int no = d.nc + 5;
    const int b = blockIdx.y;
    size_t per_frame = (size_t)d.A * no;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < per_frame; i += (size_t)gridDim.x * blockDim.x) {
        int row = (int)(i / no), c = (int)(i - (size_t)row * no);
        int l = (row >= d.row_off[2]) ? 2 : (row >= d.row_off[1] ? 1 : 0);
        int r = row - d.row_off[l];
        int hw = d.ny[l] * d.nx[l];
        int a = r / hw, p = r - a * hw;
        int y = p / d.nx[l], x = p - y * d.nx[l];
        float v = d.in[l][((size_t)b * hw + p) * d.cs[l] + a * no + c];
        float s = 1.0f / (1.0f + expf(-v));
        float o;
        if (c == 0) o = (s * 2.0f - 0.5f + (float)x) * (float)d.stride[l];
        else if (c == 1) o = (s * 2.0f - 0.5f + (float)y) * (float)d.stride[l];
        else if (c < 4) { float t = s * 2.0f; o = t * t * d.anchors[l * 6 + a * 2 + (c - 2)]; }
        else o = s;
        d.out[(size_t)b * per_frame + i] = o;
    }
}
hipError_t launch_detect_v5(const TView* ins, float* out, int n, int nc, int A, const int strides[3], const float* d_anchors,
                            hipStream_t st_) {
    DetV5Dev d;
    int off = 0;
    for (int l = 0; l < 3; ++l) {
        if (!ins[l].f32 || ins[l].c != 3 * (nc + 5) || ins[l].coff) return hipErrorInvalidValue;
        d.in[l] = (const float*)ins[l].p; d.cs[l] = ins[l].cs; d.ny[l] = ins[l].h; d.nx[l] = ins[l].w;
        d.stride[l] = strides[l]; d.row_off[l] = off;
        off += 3 * ins[l].h * ins[l].w;
    }
    if (off != A) return hipErrorInvalidValue;
    d.anchors = d_anchors; d.out = out; d.nc = nc; d.A = A; d.n = n;
    hipLaunchKernelGGL(detect_v5_kernel, dim3(2048, n), dim3(256), 0, st_, d);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------- Detect (v5 layout), fused with its 1x1 convs
// model.24.m.l (YOLOv5) / model.77.m.l (YOLOv7 IDetect, deploy form) produce 3 * (5 + nc) raw logits per cell that feed nothing but the
// decode: as separate launches the fp32 logits are written once and read once (2 x 4 * A * (5 + nc) bytes per frame, 1.1 GB per 64-frame
// step at 640x640) around a Cout = 255 GEMM that fits no 16-channel-tiled kernel.  Here a workgroup owns 128 consecutive cells of one
// anchor of one level of one frame: the anchor's 5 + nc weight rows (zero-padded to 16-row MFMA A tiles, streamed through LDS in
// 256-channel K chunks) times the cells' activations (B fragments straight from HBM), then sigmoid + grid / anchor decode in the
// accumulators, staged through LDS so that every wave writes its 32 rows of the (A, 5 + nc) output as one contiguous run.
// The sigmoid is v_exp + v_rcp (~3e-7 relative, far below the 16-bit activations' 1e-3): with IEEE expf and division the 255 sigmoids
// per cell (411 M per 64-frame step) made the launch VALU-bound at 0.35 ms, twice its 548 MB of output at HBM rate.
#define ADAS_DET5_NT 6     // 16-row tiles per anchor: 5 + nc <= 96
#define ADAS_DET5_KC 8     // K steps (of 32 channels) per LDS weight chunk
struct Det5Dev {
    const uint16_t* x[3];   // level inputs (16-bit NHWC views)
    int x_cs[3], x_coff[3], cin[3];
    const uint16_t* w[3];   // [anchor][k step][tile][lane][8] MFMA A fragments (launch_pack_weights_det5)
    const float* bias[3];   // [3 * (5 + nc)]
    int hw[3], nx[3], stride[3], row_off[3], blk_off[3];
    const float* anchors;   // [3 levels][3 anchors][2]
    float* out;
    int nc, A, n, lds_bias_off;   // lds_bias_off: in floats
    float* sink_conf;       // SINK: per-row best class confidence (class x objectness, fp32) / first arg-max class, [n][A] each
    int* sink_cls;
};

// SINK (pipeline steps): per row the post-processing reads the box and, from the scan, max_k(cls_k * obj) with its first arg-max
// (yoloDetector.py:123-127).  The decoded row sits in this wave's LDS staging area: the pair is taken there (same fp32 products, same
// first-maximum rule as yolo_scan_v5) and only the five leading values of the row go to HBM -- no class values written, no scan launch.
template <typename E, bool SINK>
__global__ __launch_bounds__(256, 3) void detect_v5_fused_kernel(Det5Dev d) {
    extern __shared__ __attribute__((aligned(16))) uint16_t wl[];      // weight chunk, then the output staging
    E::enter();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lrow = lane & 15, kg = lane >> 4;
    const int b = blockIdx.y, no = d.nc + 5, NT = (no + 15) >> 4;
    const int lvl = (int)blockIdx.x >= d.blk_off[2] ? 2 : ((int)blockIdx.x >= d.blk_off[1] ? 1 : 0);
    const int hw = d.hw[lvl], nblk = (hw + 127) >> 7;
    const int rel = blockIdx.x - d.blk_off[lvl], a = rel / nblk, p0 = (rel - a * nblk) * 128 + wave * 32;
    const int KS = d.cin[lvl] >> 5;
    const uint16_t* wsrc = d.w[lvl] + (size_t)a * KS * NT * 512;
    float* bl = reinterpret_cast<float*>(wl) + d.lds_bias_off;     // the anchor's 5 + nc biases, behind the weight / staging region
    if (tid < no) bl[tid] = d.bias[lvl][a * no + tid];              // (visible after the K loop's barriers)

    const uint16_t* ip[2];
    bool ok[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int p = p0 + t * 16 + lrow;
        ok[t] = p < hw;
        ip[t] = d.x[lvl] + ((size_t)b * hw + (ok[t] ? p : 0)) * d.x_cs[lvl] + d.x_coff[lvl] + kg * 8;
    }
    df32x4 acc[2][ADAS_DET5_NT];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int nt = 0; nt < ADAS_DET5_NT; ++nt) acc[t][nt] = df32x4{0.f, 0.f, 0.f, 0.f};

    for (int ks0 = 0; ks0 < KS; ks0 += ADAS_DET5_KC) {
        const int kc = min(ADAS_DET5_KC, KS - ks0);
        du32x4 xb[2][ADAS_DET5_KC];                         // the chunk's activations: in flight while the weights are staged
#pragma unroll
        for (int k = 0; k < ADAS_DET5_KC; ++k)
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                xb[t][k] = du32x4{0u, 0u, 0u, 0u};
                if (k < kc && ok[t]) xb[t][k] = *reinterpret_cast<const du32x4*>(ip[t] + (ks0 + k) * 32);
            }
        __syncthreads();
        const uint16_t* src = wsrc + (size_t)ks0 * NT * 512;
        stage_lds16<256, 6>(wl, src, kc * NT * 64, tid);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < ADAS_DET5_KC; ++k) {
            if (k < kc) {
#pragma unroll
                for (int nt = 0; nt < ADAS_DET5_NT; ++nt) {
                    if (nt < NT) {
                        const du32x4 wf = *reinterpret_cast<const du32x4*>(wl + ((size_t)(k * NT + nt) * 64 + lane) * 8);
                        acc[0][nt] = E::mfma(wf, xb[0][k], acc[0][nt]);
                        acc[1][nt] = E::mfma(wf, xb[1][k], acc[1][nt]);
                    }
                }
            }
        }
    }
    __syncthreads();
    // ---- sigmoid + decode (yolov5 models/yolo.py Detect.forward, inference branch) into this wave's staging rows [32][no]
    float* stage = reinterpret_cast<float*>(wl) + (size_t)wave * 32 * no;
    const float sl = (float)d.stride[lvl];
    const float aw = d.anchors[lvl * 6 + a * 2], ah = d.anchors[lvl * 6 + a * 2 + 1];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int p = p0 + t * 16 + lrow;
        const float gx = (float)(p % d.nx[lvl]), gy = (float)(p / d.nx[lvl]);
#pragma unroll
        for (int nt = 0; nt < ADAS_DET5_NT; ++nt) {
            if (nt < NT) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = nt * 16 + kg * 4 + r;
                    if (c < no) {
                        const float sg = fast_rcp(1.0f + __expf(-(acc[t][nt][r] + bl[c])));   // (elem16.h: the hardware reciprocal)
                        float o = sg;
                        if (c == 0) o = (sg * 2.0f - 0.5f + gx) * sl;
                        else if (c == 1) o = (sg * 2.0f - 0.5f + gy) * sl;
                        else if (c < 4) { const float q = sg * 2.0f; o = q * q * (c == 2 ? aw : ah); }
                        stage[(t * 16 + lrow) * no + c] = o;
                    }
                }
            }
        }
    }
    __syncthreads();
    const int rows = min(32, hw - p0);
    if (SINK) {
        if (rows > 0) {
            const size_t row0 = (size_t)b * d.A + d.row_off[lvl] + (size_t)a * hw + p0;
            if (lane < rows) {
                const float* r = stage + lane * no;
                const float obj = r[4];
                float bv = 0.f;
                int bi = -1;
                for (int c = 0; c < d.nc; ++c) {
                    const float pr = r[5 + c] * obj;
                    if (bi < 0 || pr > bv) {
                        bv = pr;
                        bi = c;
                    }
                }
                d.sink_conf[row0 + lane] = bv;
                d.sink_cls[row0 + lane] = bi < 0 ? 0 : bi;
            }
            float* dst = d.out + row0 * no;
            for (int i = lane; i < rows * 5; i += 64) {
                const int rr = i / 5, cc = i - rr * 5;
                dst[(size_t)rr * no + cc] = stage[rr * no + cc];
            }
        }
    } else if (rows > 0) {
        float* dst = d.out + ((size_t)b * d.A + d.row_off[lvl] + (size_t)a * hw + p0) * no;
        const int nel = rows * no;
        for (int i0 = 0; i0 < nel; i0 += 64 * 8) {           // eight LDS reads in flight, then eight coalesced 256-byte stores
            float tmp[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = i0 + j * 64 + lane;
                tmp[j] = stage[i < nel ? i : 0];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = i0 + j * 64 + lane;
                if (i < nel) dst[i] = tmp[j];
            }
        }
    }
}

// fp32 [3 * no][cin] (1x1 conv) -> per anchor MFMA A fragments [a][k step][tile][lane][8]: row = a * no + 16 * tile + (lane & 15) (zero
// for rows past the anchor's no), K = 32 ks + 8 (lane >> 4) + e
template <typename T>
__global__ void pack_weights_det5_kernel(const float* __restrict__ src, T* __restrict__ dst, int no, int cin, int nt_n, int total) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int nks = cin >> 5;
    const int e = idx & 7, lane = (idx >> 3) & 63, f = idx >> 9;
    const int nt = f % nt_n, ks = (f / nt_n) % nks, a = f / (nt_n * nks);
    const int c = nt * 16 + (lane & 15), k = ks * 32 + (lane >> 4) * 8 + e;
    const float v = c < no ? src[(size_t)(a * no + c) * cin + k] : 0.0f;
    st(dst + idx, v);
}
size_t det5_weight_bytes(int no, int cin) { return (size_t)3 * ((no + 15) / 16) * (cin / 32) * 512 * 2; }
bool det5_applicable(int prec, int nc, const TView& in, const TView& logits) {
    static int off = -1;
    if (off < 0) { const char* e = getenv("ADAS_NO_DETECT_FUSE"); off = (e && e[0] == '1') ? 1 : 0; }
    if (off || !prec_is16(prec) || nc + 5 > 16 * ADAS_DET5_NT) return false;
    if (in.f32 || !logits.f32 || logits.c != 3 * (nc + 5) || logits.coff) return false;
    return (in.c & 31) == 0 && ((in.cs | in.coff) & 7) == 0 && in.h == logits.h && in.w == logits.w;
}
hipError_t launch_pack_weights_det5(const float* src, void* dst, int no, int cin, int prec, hipStream_t st) {
    const int nt_n = (no + 15) / 16, total = 3 * nt_n * (cin / 32) * 512;
    if (prec == PREC_FP16) hipLaunchKernelGGL(pack_weights_det5_kernel<f16s>, dim3((total + 255) / 256), dim3(256), 0, st, src, (f16s*)dst, no, cin, nt_n, total);
    else hipLaunchKernelGGL(pack_weights_det5_kernel<uint16_t>, dim3((total + 255) / 256), dim3(256), 0, st, src, (uint16_t*)dst, no, cin, nt_n, total);
    return hipGetLastError();
}
// hidden[l]: input of the level's 1x1 conv; wfrag / bias: its det5-packed weights and its 3 * (5 + nc) biases
hipError_t launch_detect_v5_fused(const TView* hidden, const void* const* wfrag, const float* const* bias, float* out, int n, int nc, int A,
                                  const int strides[3], const float* d_anchors, int prec, hipStream_t st_, float* sink_conf, int* sink_cls) {
    Det5Dev d;
    int off = 0, blocks = 0;
    const int no = nc + 5, NT = (no + 15) / 16;
    if (NT > ADAS_DET5_NT) return hipErrorInvalidValue;
    for (int l = 0; l < 3; ++l) {
        const TView& h = hidden[l];
        if (h.f32 || (h.c & 31) || ((h.cs | h.coff) & 7)) return hipErrorInvalidValue;
        d.x[l] = (const uint16_t*)h.p; d.x_cs[l] = h.cs; d.x_coff[l] = h.coff; d.cin[l] = h.c;
        d.w[l] = (const uint16_t*)wfrag[l]; d.bias[l] = bias[l];
        d.hw[l] = h.h * h.w; d.nx[l] = h.w; d.stride[l] = strides[l]; d.row_off[l] = off; d.blk_off[l] = blocks;
        off += 3 * d.hw[l];
        blocks += 3 * ((d.hw[l] + 127) / 128);
    }
    if (off != A) return hipErrorInvalidValue;
    d.anchors = d_anchors; d.out = out; d.nc = nc; d.A = A; d.n = n;
    const size_t lw = (size_t)ADAS_DET5_KC * NT * 1024, ls = (size_t)4 * 32 * no * 4;
    const size_t region = ((lw > ls ? lw : ls) + 15) & ~(size_t)15;
    d.lds_bias_off = (int)(region / 4);
    const size_t lds = region + (size_t)no * 4;
    d.sink_conf = sink_conf; d.sink_cls = sink_cls;
    if (sink_conf && sink_cls)
        ADAS_DISPATCH_E16(prec == PREC_FP16, E, hipLaunchKernelGGL((detect_v5_fused_kernel<E, true>), dim3(blocks, n), dim3(256), lds, st_, d));
    else
        ADAS_DISPATCH_E16(prec == PREC_FP16, E, hipLaunchKernelGGL((detect_v5_fused_kernel<E, false>), dim3(blocks, n), dim3(256), lds, st_, d));
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------- Detect (v6: anchor-free, v5 output layout)
// YOLOv6 EffiDeHead.forward, inference branch without DFL (yolov6/models/effidehead.py, yolov6/assigners/anchor_generator.py,
// yolov6/utils/general.py dist2bbox 'xywh'): per level raw (l, t, r, b) distances in grid units and class logits ->
// row = [cx, cy, w, h] * stride with anchor point (x + 0.5, y + 0.5), objectness 1, sigmoid(class logits): (A, 5 + nc), A = sum of cells.
struct DetV6Dev {
    const float* reg[3];
    const float* cls[3];
    int reg_cs[3], cls_cs[3], hw[3], nx[3], stride[3], row_off[3];
    float* out;
    int nc, A, n;
};
__global__ __launch_bounds__(256) void detect_v6_kernel(DetV6Dev d) {   // workgroup = 32 consecutive rows of one frame (contiguous in and out)
    const int no = d.nc + 5;
    const int b = blockIdx.y, row0 = blockIdx.x * 32;
    const int nel = min(32, d.A - row0) * no;
    float* out = d.out + ((size_t)b * d.A + row0) * no;
    for (int e = threadIdx.x; e < nel; e += 256) {
        const int r = e / no, c = e - r * no;               // 32-bit, e < 32 * no (a 64-bit division per element tripled the launch time)
        const int row = row0 + r;
        const int l = (row >= d.row_off[2]) ? 2 : (row >= d.row_off[1] ? 1 : 0);
        const int p = row - d.row_off[l];
        const size_t pix = (size_t)b * d.hw[l] + p;
        float o;
        if (c < 4) {
            const float4 rg = *reinterpret_cast<const float4*>(d.reg[l] + pix * d.reg_cs[l]);
            const float ax = (float)(p % d.nx[l]) + 0.5f, ay = (float)(p / d.nx[l]) + 0.5f;
            const float x1 = ax - rg.x, y1 = ay - rg.y, x2 = ax + rg.z, y2 = ay + rg.w;
            const float s = (float)d.stride[l];
            o = (c == 0 ? (x1 + x2) / 2 : c == 1 ? (y1 + y2) / 2 : c == 2 ? x2 - x1 : y2 - y1) * s;
        } else if (c == 4) {
            o = 1.0f;
        } else {
            o = 1.0f / (1.0f + expf(-d.cls[l][pix * d.cls_cs[l] + (c - 5)]));
        }
        out[e] = o;
    }
}
// ins[2l] = reg_preds.l (4 fp32 channels), ins[2l + 1] = cls_preds.l (nc fp32 channels)
hipError_t launch_detect_v6(const TView* ins, float* out, int n, int nc, int A, const int strides[3], hipStream_t st_) {
    DetV6Dev d;
    int off = 0;
    for (int l = 0; l < 3; ++l) {
        const TView& r = ins[2 * l];
        const TView& c = ins[2 * l + 1];
        if (!r.f32 || !c.f32 || r.c != 4 || c.c != nc || r.coff || c.coff || r.h != c.h || r.w != c.w) return hipErrorInvalidValue;
        d.reg[l] = (const float*)r.p; d.cls[l] = (const float*)c.p; d.reg_cs[l] = r.cs; d.cls_cs[l] = c.cs;
        d.hw[l] = r.h * r.w; d.nx[l] = r.w; d.stride[l] = strides[l]; d.row_off[l] = off;
        off += d.hw[l];
    }
    if (off != A) return hipErrorInvalidValue;
    d.out = out; d.nc = nc; d.A = A; d.n = n;
    if (((ins[0].cs | ins[2].cs | ins[4].cs) & 3) != 0) return hipErrorInvalidValue;   // float4 loads of the (l, t, r, b) distances
    hipLaunchKernelGGL(detect_v6_kernel, dim3((A + 31) / 32, n), dim3(256), 0, st_, d);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------- LayerNorm
template <typename T>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ in, T* __restrict__ out,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        int len, float eps) {
    __shared__ float red[8];
    const float* x = in + (size_t)blockIdx.x * len;
    T* y = out + (size_t)blockIdx.x * len;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    float s = 0.f;
    for (int i = tid; i < len; i += 256) s += x[i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if (lane == 0) red[wv] = s;
    __syncthreads();
    const float mean = (red[0] + red[1] + red[2] + red[3]) / (float)len;
    __syncthreads();
    float v = 0.f;
    for (int i = tid; i < len; i += 256) {
        float dlt = x[i] - mean;
        v += dlt * dlt;
    }
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if (lane == 0) red[4 + wv] = v;
    __syncthreads();
    const float var = (red[4] + red[5] + red[6] + red[7]) / (float)len;
    const float rstd = 1.0f / sqrtf(var + eps);
    for (int i = tid; i < len; i += 256) st<T>(y + i, (x[i] - mean) * rstd * gamma[i] + beta[i]);
}
hipError_t launch_layernorm(const float* in, void* out, const float* gamma, const float* beta, int n, int len, float eps,
                            int prec, hipStream_t st_) {
    ADAS_DISPATCH_STORAGE(prec, T, hipLaunchKernelGGL(layernorm_kernel<T>, dim3(n), dim3(256), 0, st_, in, (T*)out, gamma, beta, len, eps));
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------- parity tap
template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ in, float* __restrict__ out, int n, int hw, int c, int cs, int coff) {
    size_t total = (size_t)n * c * hw;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        size_t p = i % hw;
        size_t t = i / hw;
        int ch = (int)(t % c);
        size_t b = t / c;
        out[i] = ld<T>(in + (b * hw + p) * cs + coff + ch);
    }
}
hipError_t launch_nhwc_to_nchw(TView in, float* out, int n, int prec, hipStream_t st_) {
    int hw = in.h * in.w;
    size_t total = (size_t)n * in.c * hw;
    int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    ADAS_DISPATCH_STORAGE(in.f32 ? PREC_FP32 : prec, T,
                          hipLaunchKernelGGL(nhwc_to_nchw_kernel<T>, dim3(blocks), dim3(256), 0, st_, (const T*)in.p, out, n, hw, in.c, in.cs, in.coff));
    return hipGetLastError();
}

}  // namespace adas
