// dw_attn.hip -- the two YOLOv10 operators the v8-family kernels do not cover (the reference's shipped default detector is
// yolov10n: demo.py:24-30; its head is decoded as a v8-layout tensor, yoloDetector.py:114,121):
//   dwconv     depth-wise k x k convolution (k = 3, 5 or 7, stride 1 or 2, groups = channels) + bias [+ SiLU] [+ residual add]:
//              SCDown.cv2, CIB's three depth-wise layers (the fused RepVGGDW is one 7x7), v10Detect's class-branch 3x3s and
//              the positional encoding of PSA's attention.  HBM-bound streaming: thread = (output pixel, 8-channel group), one
//              16-byte (16-bit modes) load per tap from a window that lives in L1/L2, fp32 weights [tap][C], fp32 accumulate.
//   attention  PSA's softmax attention (ultralytics Attention: per head q, k of key_dim and v of head_dim channels cut out of one
//              qkv tensor; out[d, i] = sum_j v[d, j] * softmax_j(q_i . k_j * scale)) on the 20x20 (N = 400 tokens) P5 map.  One
//              thread per query token: keys / values stream through LDS in chunks of 64 tokens (fp32), scores of a chunk sit in
//              registers, the running maximum / denominator are rescaled once per chunk (online softmax).  Latency-sized work
//              (31 MFLOP per frame and head): no MFMA.
// All three precisions run the same code on the storage type (elem16.h / float).
#include "kernels.h"
#include "elem16.h"
#include <stdlib.h>

namespace adas {

struct DwDev {
    const void* in;
    void* out;
    const void* res;
    const float* wgt;   // [k*k][C] fp32
    const float* bias;  // [C]
    int in_cs, in_coff, out_cs, out_coff, res_cs, res_coff;
    int c, H, W, Ho, Wo, k, s, p, n, act, has_res, small;
};

template <typename T, int K>
__global__ __launch_bounds__(256) void dwconv_kernel(DwDev d) {
    if constexpr (sizeof(T) == 2 && !__is_same(T, uint16_t)) Fp16::enter();
    const int c8n = d.c >> 3;
    const size_t total = (size_t)d.n * d.Ho * d.Wo * c8n;
    const T* __restrict__ in = (const T*)d.in;
    // XCD-aware traversal: workgroup b runs on XCD b % 8 (each XCD has its own L2).  A k x k window shares its rows with the outputs one
    // row up and down, i.e. with workgroups ~one image row of threads away -- under a plain grid-stride loop those sit on OTHER XCDs and
    // every input row is fetched into three (k = 3) L2s.  Here each XCD owns one contiguous eighth of the index space and its workgroups
    // stride inside it: vertical neighbours share an L2 (measured on EfficientNet-B0's depth-wise layers: DESIGN 9.7).
    const unsigned xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const size_t chunk = (total + 7) / 8, lo = (size_t)xcd * chunk, hi = lo + chunk < total ? lo + chunk : total;
    for (size_t i = lo + (size_t)slot * blockDim.x + threadIdx.x; i < hi; i += (size_t)per_xcd * blockDim.x) {
        int c8, ox, oy;
        size_t pix, b;
        if (d.small) {   // the whole index space fits 32 bits: 32-bit divisions (a 64-bit one costs ~10x; uniform branch)
            const unsigned iu = (unsigned)i, pu = iu / (unsigned)c8n, tu = pu / (unsigned)d.Wo;
            c8 = (int)(iu - pu * (unsigned)c8n);
            ox = (int)(pu - tu * (unsigned)d.Wo);
            const unsigned bu = tu / (unsigned)d.Ho;
            oy = (int)(tu - bu * (unsigned)d.Ho);
            pix = pu; b = bu;
        } else {
            c8 = (int)(i % c8n);
            pix = i / c8n;
            ox = (int)(pix % d.Wo);
            const size_t t = pix / d.Wo;
            oy = (int)(t % d.Ho);
            b = t / d.Ho;
        }
        const int c = c8 * 8;
        float acc[8];
        {
            const float4 b0 = *reinterpret_cast<const float4*>(d.bias + c), b1 = *reinterpret_cast<const float4*>(d.bias + c + 4);
            acc[0] = b0.x; acc[1] = b0.y; acc[2] = b0.z; acc[3] = b0.w; acc[4] = b1.x; acc[5] = b1.y; acc[6] = b1.z; acc[7] = b1.w;
        }
        // K is a compile-time constant (3 / 5 / 7): both tap loops unroll, the row's address arithmetic is hoisted and the K * K
        // input loads of an output are independent of each other (same taps in the same (r, q) order: identical sums)
        const int iy0 = oy * d.s - d.p, ix0 = ox * d.s - d.p;
        const T* base = in + (b * d.H * d.W) * (size_t)d.in_cs + d.in_coff + c;
        const float* wbase = d.wgt + c;
#pragma unroll
        for (int r = 0; r < K; ++r) {
            const int iy = iy0 + r;
            if ((unsigned)iy >= (unsigned)d.H) continue;
            const T* rowp = base + (size_t)iy * d.W * d.in_cs;
#pragma unroll
            for (int q = 0; q < K; ++q) {
                const int ix = ix0 + q;
                if ((unsigned)ix >= (unsigned)d.W) continue;
                float x[8];
                Vec8<T>::load(rowp + (size_t)ix * d.in_cs, x);
                const float* wp = wbase + (size_t)(r * K + q) * d.c;
                const float4 w0 = *reinterpret_cast<const float4*>(wp), w1 = *reinterpret_cast<const float4*>(wp + 4);
                acc[0] = fmaf(x[0], w0.x, acc[0]); acc[1] = fmaf(x[1], w0.y, acc[1]); acc[2] = fmaf(x[2], w0.z, acc[2]); acc[3] = fmaf(x[3], w0.w, acc[3]);
                acc[4] = fmaf(x[4], w1.x, acc[4]); acc[5] = fmaf(x[5], w1.y, acc[5]); acc[6] = fmaf(x[6], w1.z, acc[6]); acc[7] = fmaf(x[7], w1.w, acc[7]);
            }
        }
        if (d.act == ACT_SILU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = silu_for<T>(acc[e]);
        } else if (d.act == ACT_RELU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = fmaxf(acc[e], 0.0f);
        }
        if (d.has_res) {   // x + block(x): added after the activation (RES_AFTER_ACT)
            float rv[8];
            Vec8<T>::load((const T*)d.res + pix * d.res_cs + d.res_coff + c, rv);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += rv[e];
        }
        Vec8<T>::store((T*)d.out + pix * d.out_cs + d.out_coff + c, acc);
    }
}

// Strip form: a thread owns P consecutive outputs of one row and one 8-channel group.  The plain kernel above issues, per output, K*K
// 16-byte activation loads and 2*K*K 16-byte weight loads through the 64 B/clk L1 -- 27 KB per wave at K = 3, which bounds
// EfficientNet's 128x128x144 layer at ~240 us where HBM needs 120 (measured 386; unrolling, 32-bit index arithmetic and an XCD-aware
// traversal changed nothing: round 4).  Here one tap ROW of weights (K x 8 fp32) sits in registers while the strip's
// (P - 1) * S + K input columns stream past it: per output K * ((P-1)S + K) / P activation loads and 2K*K / P weight loads (K = 3,
// S = 1, P = 4: 4.5 + 4.5 instead of 9 + 18).  Every output still accumulates its taps in (row, column) order with fmaf: results are
// bit-identical to the plain kernel's.
template <typename T, int K, int S, int P>
__global__ __launch_bounds__(256) void dwconv_strip_kernel(DwDev d) {
    if constexpr (sizeof(T) == 2 && !__is_same(T, uint16_t)) Fp16::enter();
    constexpr int NC = (P - 1) * S + K;   // input columns a strip touches
    const unsigned c8n = (unsigned)(d.c >> 3), strips = (unsigned)((d.Wo + P - 1) / P);
    const unsigned total = (unsigned)d.n * (unsigned)d.Ho * strips * c8n;   // host guarantees < 2^31
    const T* __restrict__ in = (const T*)d.in;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const unsigned u = i / c8n, c8 = i - u * c8n;
        const unsigned v = u / strips, sx = u - v * strips;
        const unsigned b = v / (unsigned)d.Ho, oy = v - b * (unsigned)d.Ho;
        const int c = (int)c8 * 8, ox0 = (int)sx * P;
        float acc[P][8];
        {
            const float4 b0 = *reinterpret_cast<const float4*>(d.bias + c), b1 = *reinterpret_cast<const float4*>(d.bias + c + 4);
#pragma unroll
            for (int t = 0; t < P; ++t) {
                acc[t][0] = b0.x; acc[t][1] = b0.y; acc[t][2] = b0.z; acc[t][3] = b0.w;
                acc[t][4] = b1.x; acc[t][5] = b1.y; acc[t][6] = b1.z; acc[t][7] = b1.w;
            }
        }
        const int iy0 = (int)oy * S - d.p, ix0 = ox0 * S - d.p;
        const T* base = in + ((size_t)b * d.H * d.W) * (size_t)d.in_cs + d.in_coff + c;
#pragma unroll
        for (int r = 0; r < K; ++r) {
            const int iy = iy0 + r;
            if ((unsigned)iy >= (unsigned)d.H) continue;
            float w[K][8];
#pragma unroll
            for (int q = 0; q < K; ++q) {
                const float* wp = d.wgt + (size_t)(r * K + q) * d.c + c;
                const float4 w0 = *reinterpret_cast<const float4*>(wp), w1 = *reinterpret_cast<const float4*>(wp + 4);
                w[q][0] = w0.x; w[q][1] = w0.y; w[q][2] = w0.z; w[q][3] = w0.w; w[q][4] = w1.x; w[q][5] = w1.y; w[q][6] = w1.z; w[q][7] = w1.w;
            }
            const T* rowp = base + (size_t)iy * d.W * d.in_cs;
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                const int ix = ix0 + j;
                if ((unsigned)ix >= (unsigned)d.W) continue;
                float x[8];
                Vec8<T>::load(rowp + (size_t)ix * d.in_cs, x);
#pragma unroll
                for (int t = 0; t < P; ++t) {
                    const int q = j - t * S;            // compile-time after unrolling: the tap this column is for output t
                    if (q >= 0 && q < K) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) acc[t][e] = fmaf(x[e], w[q][e], acc[t][e]);
                    }
                }
            }
        }
#pragma unroll
        for (int t = 0; t < P; ++t) {
            const int ox = ox0 + t;
            if (ox >= d.Wo) break;
            if (d.act == ACT_SILU) {
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[t][e] = silu_for<T>(acc[t][e]);
            } else if (d.act == ACT_RELU) {
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[t][e] = fmaxf(acc[t][e], 0.0f);
            }
            const size_t pix = ((size_t)b * d.Ho + oy) * d.Wo + ox;
            if (d.has_res) {
                float rv[8];
                Vec8<T>::load((const T*)d.res + pix * d.res_cs + d.res_coff + c, rv);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[t][e] += rv[e];
            }
            Vec8<T>::store((T*)d.out + pix * d.out_cs + d.out_coff + c, acc[t]);
        }
    }
}

bool dwconv_supported(int k, int stride, int pad, int res_mode, const TView& in, const TView& out) {
    if ((k != 3 && k != 5 && k != 7) || (stride != 1 && stride != 2) || pad != k / 2) return false;
    if (res_mode != RES_NONE && res_mode != RES_AFTER_ACT) return false;
    if (in.c != out.c || (in.c & 7) || (in.cs & 7) || (in.coff & 7) || (out.cs & 7) || (out.coff & 7) || in.f32 || out.f32) return false;
    return out.h == (in.h + 2 * pad - k) / stride + 1 && out.w == (in.w + 2 * pad - k) / stride + 1;
}

hipError_t launch_dwconv(const TView& in, const TView& out, const TView& res, int res_mode, const float* wgt, const float* bias, int n, int k,
                         int stride, int pad, int act, int prec, hipStream_t st) {
    if (!dwconv_supported(k, stride, pad, res_mode, in, out)) return hipErrorInvalidValue;
    if (res_mode != RES_NONE && ((res.cs & 7) || (res.coff & 7) || res.f32 || res.h != out.h || res.w != out.w)) return hipErrorInvalidValue;
    DwDev d;
    d.in = in.p; d.out = out.p; d.res = res_mode != RES_NONE ? res.p : nullptr; d.wgt = wgt; d.bias = bias;
    d.in_cs = in.cs; d.in_coff = in.coff; d.out_cs = out.cs; d.out_coff = out.coff; d.res_cs = res.cs; d.res_coff = res.coff;
    d.c = in.c; d.H = in.h; d.W = in.w; d.Ho = out.h; d.Wo = out.w; d.k = k; d.s = stride; d.p = pad; d.n = n; d.act = act;
    d.has_res = res_mode != RES_NONE;
    const size_t total = (size_t)n * out.h * out.w * (in.c >> 3);
    d.small = total < ((size_t)1 << 31) ? 1 : 0;
    {   // strip form (4 outputs per thread) whenever a row has at least one whole strip and the index space fits 32 bits
        const char* e = getenv("ADAS_NO_DW_STRIP");     // read per launch (tests hold the two forms against each other in one process)
        const int use_strip = (e && e[0] == '1') ? 0 : 1;
        const size_t items = (size_t)n * out.h * ((out.w + 3) / 4) * (in.c >> 3);
        if (use_strip && out.w >= 4 && items < ((size_t)1 << 31)) {
            const int sb = (int)((items + 255) / 256 < 16384 ? (items + 255) / 256 : 16384);
#define ADAS_DWS(T, K_, S_) hipLaunchKernelGGL((dwconv_strip_kernel<T, K_, S_, 4>), dim3(sb), dim3(256), 0, st, d)
#define ADAS_DWS_K(T)                                                   \
    do {                                                                \
        if (k == 3 && stride == 1) ADAS_DWS(T, 3, 1);                   \
        else if (k == 3) ADAS_DWS(T, 3, 2);                             \
        else if (k == 5 && stride == 1) ADAS_DWS(T, 5, 1);              \
        else if (k == 5) ADAS_DWS(T, 5, 2);                             \
        else if (stride == 1) ADAS_DWS(T, 7, 1);                        \
        else ADAS_DWS(T, 7, 2);                                         \
    } while (0)
            if (prec == PREC_FP32) ADAS_DWS_K(float);
            else if (prec == PREC_X3) ADAS_DWS_K(x3s);
            else if (prec == PREC_FP16) ADAS_DWS_K(f16s);
            else ADAS_DWS_K(uint16_t);
#undef ADAS_DWS_K
#undef ADAS_DWS
            return hipGetLastError();
        }
    }
    int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    blocks = (blocks + 7) & ~7;   // a whole number of workgroups per XCD
#define ADAS_DW_LAUNCH(T)                                                                                     \
    do {                                                                                                      \
        if (k == 3) hipLaunchKernelGGL((dwconv_kernel<T, 3>), dim3(blocks), dim3(256), 0, st, d);             \
        else if (k == 5) hipLaunchKernelGGL((dwconv_kernel<T, 5>), dim3(blocks), dim3(256), 0, st, d);        \
        else hipLaunchKernelGGL((dwconv_kernel<T, 7>), dim3(blocks), dim3(256), 0, st, d);                    \
    } while (0)
    if (prec == PREC_FP32) ADAS_DW_LAUNCH(float);
    else if (prec == PREC_X3) ADAS_DW_LAUNCH(x3s);
    else if (prec == PREC_FP16) ADAS_DW_LAUNCH(f16s);
    else ADAS_DW_LAUNCH(uint16_t);
#undef ADAS_DW_LAUNCH
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------- attention
struct AttnDev {
    const void* qkv;
    void* out;
    int in_cs, in_coff, out_cs, out_coff;
    int N;          // tokens per frame (H * W)
    int nh;
    float scale;
};

constexpr int AT_KD = 32, AT_HD = 64, AT_CH = 64, AT_THR = 128;

template <typename T> __device__ __forceinline__ float at_ld(const T* p);
template <> __device__ __forceinline__ float at_ld<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float at_ld<uint16_t>(const uint16_t* p) { return Bf16::to_f32(*p); }
template <> __device__ __forceinline__ float at_ld<f16s>(const f16s* p) { return Fp16::to_f32(p->v); }
template <typename T> __device__ __forceinline__ void at_st(T* p, float v);
template <> __device__ __forceinline__ void at_st<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void at_st<uint16_t>(uint16_t* p, float v) { *p = Bf16::from_f32(v); }
template <> __device__ __forceinline__ void at_st<f16s>(f16s* p, float v) { p->v = Fp16::from_f32(v); }
template <> __device__ __forceinline__ float at_ld<x3s>(const x3s* p) { return x3_ld(p); }
template <> __device__ __forceinline__ void at_st<x3s>(x3s* p, float v) { x3_st(p, v); }

// grid (ceil(N / AT_THR), nh, batch); thread = one query token
template <typename T>
__global__ __launch_bounds__(AT_THR) void attention_kernel(AttnDev a) {
    if constexpr (sizeof(T) == 2 && !__is_same(T, uint16_t)) Fp16::enter();
    __shared__ float Ks[AT_CH][AT_KD + 1];
    __shared__ float Vs[AT_CH][AT_HD + 1];
    const int tid = threadIdx.x, h = blockIdx.y;
    const size_t b = blockIdx.z;
    const int i = blockIdx.x * AT_THR + tid;
    const int hc = h * (2 * AT_KD + AT_HD);
    const T* base = (const T*)a.qkv + b * (size_t)a.N * a.in_cs + a.in_coff + hc;
    float q[AT_KD];
    {
        const T* qp = base + (size_t)(i < a.N ? i : 0) * a.in_cs;
#pragma unroll
        for (int d = 0; d < AT_KD; ++d) q[d] = at_ld<T>(qp + d) * a.scale;
    }
    float acc[AT_HD];
#pragma unroll
    for (int d = 0; d < AT_HD; ++d) acc[d] = 0.f;
    float m = -INFINITY, l = 0.f;
    for (int j0 = 0; j0 < a.N; j0 += AT_CH) {
        const int nj = a.N - j0 < AT_CH ? a.N - j0 : AT_CH;
        __syncthreads();
        for (int e = tid; e < AT_CH * AT_KD; e += AT_THR) {
            const int j = e / AT_KD, d = e - j * AT_KD;
            Ks[j][d] = j < nj ? at_ld<T>(base + (size_t)(j0 + j) * a.in_cs + AT_KD + d) : 0.f;
        }
        for (int e = tid; e < AT_CH * AT_HD; e += AT_THR) {
            const int j = e / AT_HD, d = e - j * AT_HD;
            Vs[j][d] = j < nj ? at_ld<T>(base + (size_t)(j0 + j) * a.in_cs + 2 * AT_KD + d) : 0.f;
        }
        __syncthreads();
        float s[AT_CH];
        float cm = -INFINITY;
#pragma unroll
        for (int j = 0; j < AT_CH; ++j) {
            float v = 0.f;
#pragma unroll
            for (int d = 0; d < AT_KD; ++d) v = fmaf(q[d], Ks[j][d], v);
            s[j] = j < nj ? v : -INFINITY;
            cm = fmaxf(cm, s[j]);
        }
        const float mn = fmaxf(m, cm);
        const float corr = expf(m - mn);     // 0 on the first chunk (m = -inf)
        l *= corr;
#pragma unroll
        for (int d = 0; d < AT_HD; ++d) acc[d] *= corr;
#pragma unroll
        for (int j = 0; j < AT_CH; ++j) {
            const float p = expf(s[j] - mn);  // exp(-inf) = 0 for the padded tail
            l += p;
#pragma unroll
            for (int d = 0; d < AT_HD; ++d) acc[d] = fmaf(p, Vs[j][d], acc[d]);
        }
        m = mn;
    }
    if (i < a.N) {
        const float inv = 1.0f / l;
        T* op = (T*)a.out + (b * (size_t)a.N + i) * a.out_cs + a.out_coff + h * AT_HD;
#pragma unroll
        for (int d = 0; d < AT_HD; ++d) at_st<T>(op + d, acc[d] * inv);
    }
}

// ---- 16-bit modes: the same attention on the matrix cores (flash-attention form).  A wave owns 16 queries, a workgroup (4 waves) 64;
// keys / values stream through LDS in chunks of 128 tokens.  S = K Q^T as 8 MFMAs per chunk with K as the A operand (one 32-deep step:
// key_dim = 32), so a lane ends up with 4 consecutive KEYS of one query per 16-key tile -- which is, two tiles at a time, exactly the
// B-operand fragment of the second product out^T = V^T P (8 k-elements per lane: the sum over keys does not care about their order, so
// "k index e" is defined as tile 2u row 4g+e / tile 2u+1 row 4g+e-4 and V^T is read from LDS in that order): no lane exchange between
// the two GEMMs.  Softmax statistics per query live in the 4 lanes that share its column (two xor-shuffles), online over chunks.
constexpr int FA_CH = 128, FA_KP = AT_KD + 8, FA_VP = FA_CH + 16;   // LDS pitches in elements: K rows 80 B, V^T rows 288 B

template <typename E>
__global__ __launch_bounds__(256) void attention_mfma_kernel(AttnDev a) {
    E::enter();
    __shared__ __attribute__((aligned(16))) uint16_t Ks[FA_CH][FA_KP];
    __shared__ __attribute__((aligned(16))) uint16_t Vt[AT_HD][FA_VP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = blockIdx.y;
    const int lrow = lane & 15, g = lane >> 4;
    const size_t b = blockIdx.z;
    const int hc = h * (2 * AT_KD + AT_HD);
    const uint16_t* base = (const uint16_t*)a.qkv + b * (size_t)a.N * a.in_cs + a.in_coff + hc;
    const int qi = blockIdx.x * 64 + wave * 16 + lrow;          // this lane's query (B-operand column)
    const e_u32x4 qf = *reinterpret_cast<const e_u32x4*>(base + (size_t)(qi < a.N ? qi : 0) * a.in_cs + g * 8);
    e_f32x4 acc[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) acc[dt] = e_f32x4{0.f, 0.f, 0.f, 0.f};
    float m = -INFINITY, l = 0.f;
    for (int j0 = 0; j0 < a.N; j0 += FA_CH) {
        const int nj = a.N - j0 < FA_CH ? a.N - j0 : FA_CH;
        __syncthreads();
        // K chunk: 128 keys x 32 channels = 512 16-byte pieces; V chunk transposed: 128 keys x 64 channels = 1024 pieces, 8 b16 stores each
        for (int e = tid; e < FA_CH * 4; e += 256) {
            const int j = e >> 2, c8 = e & 3;
            e_u32x4 v{0u, 0u, 0u, 0u};
            if (j < nj) v = *reinterpret_cast<const e_u32x4*>(base + (size_t)(j0 + j) * a.in_cs + AT_KD + c8 * 8);
            *reinterpret_cast<e_u32x4*>(&Ks[j][c8 * 8]) = v;
        }
        for (int e = tid; e < FA_CH * 8; e += 256) {
            const int j = e >> 3, c8 = e & 7;
            e_u32x4 v{0u, 0u, 0u, 0u};
            if (j < nj) v = *reinterpret_cast<const e_u32x4*>(base + (size_t)(j0 + j) * a.in_cs + 2 * AT_KD + c8 * 8);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                Vt[c8 * 8 + 2 * q][j] = (uint16_t)(v[q] & 0xffffu);
                Vt[c8 * 8 + 2 * q + 1][j] = (uint16_t)(v[q] >> 16);
            }
        }
        __syncthreads();
        // S tile: keys (t * 16 + 4g + r) x query lrow
        e_f32x4 sc[FA_CH / 16];
        float cm = -INFINITY;
#pragma unroll
        for (int t = 0; t < FA_CH / 16; ++t) {
            const e_u32x4 kf = *reinterpret_cast<const e_u32x4*>(&Ks[t * 16 + lrow][g * 8]);
            sc[t] = E::mfma(kf, qf, e_f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool valid = t * 16 + g * 4 + r < nj;
                sc[t][r] = valid ? sc[t][r] * a.scale : -INFINITY;
                cm = fmaxf(cm, sc[t][r]);
            }
        }
        cm = fmaxf(cm, __shfl_xor(cm, 16));
        cm = fmaxf(cm, __shfl_xor(cm, 32));
        const float mn = fmaxf(m, cm);
        const float corr = __expf(m - mn);
        float ls = 0.f;
        uint32_t pk[FA_CH / 16][2];
#pragma unroll
        for (int t = 0; t < FA_CH / 16; ++t) {
            float p[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                p[r] = __expf(sc[t][r] - mn);
                ls += p[r];
            }
            pk[t][0] = E::pack2(p[0], p[1]);
            pk[t][1] = E::pack2(p[2], p[3]);
        }
        ls += __shfl_xor(ls, 16);
        ls += __shfl_xor(ls, 32);
        l = l * corr + ls;
        m = mn;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[dt][r] *= corr;
        }
#pragma unroll
        for (int u = 0; u < FA_CH / 32; ++u) {
            const e_u32x4 pf{pk[2 * u][0], pk[2 * u][1], pk[2 * u + 1][0], pk[2 * u + 1][1]};
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const uint16_t* vr = &Vt[dt * 16 + lrow][u * 32 + g * 4];
                const uint2 v0 = *reinterpret_cast<const uint2*>(vr), v1 = *reinterpret_cast<const uint2*>(vr + 16);
                acc[dt] = E::mfma(e_u32x4{v0.x, v0.y, v1.x, v1.y}, pf, acc[dt]);
            }
        }
    }
    if (qi < a.N) {
        const float inv = 1.0f / l;
        uint16_t* op = (uint16_t*)a.out + (b * (size_t)a.N + qi) * a.out_cs + a.out_coff + h * AT_HD + g * 4;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            uint2 q;
            q.x = E::pack2(acc[dt][0] * inv, acc[dt][1] * inv);
            q.y = E::pack2(acc[dt][2] * inv, acc[dt][3] * inv);
            *reinterpret_cast<uint2*>(op + dt * 16) = q;
        }
    }
}

bool attention_supported(int nh, int kd, int hd, const TView& qkv, const TView& out) {
    return kd == AT_KD && hd == AT_HD && nh >= 1 && qkv.c == nh * (2 * kd + hd) && out.c == nh * hd && qkv.h == out.h && qkv.w == out.w && !qkv.f32 &&
           !out.f32;
}

hipError_t launch_attention(const TView& qkv, const TView& out, int n, int nh, int kd, int hd, float scale, int prec, hipStream_t st) {
    if (!attention_supported(nh, kd, hd, qkv, out)) return hipErrorInvalidValue;
    AttnDev a;
    a.qkv = qkv.p; a.out = out.p; a.in_cs = qkv.cs; a.in_coff = qkv.coff; a.out_cs = out.cs; a.out_coff = out.coff;
    a.N = qkv.h * qkv.w; a.nh = nh; a.scale = scale;
    const dim3 grid((a.N + AT_THR - 1) / AT_THR, nh, n);
    static int mfma = -1;
    if (mfma < 0) {
        const char* e = getenv("ADAS_NO_ATTN_MFMA");
        mfma = (e && e[0] == '1') ? 0 : 1;
    }
    const bool aligned = !((qkv.cs | qkv.coff) & 7) && !((out.cs | out.coff) & 3);
    if (prec == PREC_FP32) hipLaunchKernelGGL(attention_kernel<float>, grid, dim3(AT_THR), 0, st, a);
    else if (prec == PREC_X3) hipLaunchKernelGGL(attention_kernel<x3s>, grid, dim3(AT_THR), 0, st, a);   // fp32 arithmetic on the joined values
    else if (mfma && aligned) {
        const dim3 g2((a.N + 63) / 64, nh, n);
        if (prec == PREC_FP16) hipLaunchKernelGGL(attention_mfma_kernel<Fp16>, g2, dim3(256), 0, st, a);
        else hipLaunchKernelGGL(attention_mfma_kernel<Bf16>, g2, dim3(256), 0, st, a);
    } else if (prec == PREC_FP16) hipLaunchKernelGGL(attention_kernel<f16s>, grid, dim3(AT_THR), 0, st, a);
    else hipLaunchKernelGGL(attention_kernel<uint16_t>, grid, dim3(AT_THR), 0, st, a);
    return hipGetLastError();
}

}  // namespace adas
