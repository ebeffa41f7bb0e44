// fuse_ops.hip -- the element-wise operators of the EfficientDet graph (the reference's fourth detector family:
// ObjectDetector/efficientdetDetector.py:18-111 loads an exported EfficientDet-D0; EfficientNet MBConv blocks carry a
// squeeze-and-excitation gate, BiFPN nodes are weighted sums of 2-3 maps at one resolution):
//   se_gate   gate[n][c] = sigmoid(W2 silu(W1 mean_hw(x[n]) + b1) + b2)           (OP_SE_GATE; fp32 arithmetic in every precision)
//   scale     out[n][p][c] = x[n][p][c] * gate[n][c]                                (OP_SCALE)
//   wsum      out = act(w0 a [+ w1 b [+ w2 c]]), an input of half the output's resolution is read at (y / 2, x / 2); with ONE input it is the
//             stand-alone activation layer (hard-swish / hard-sigmoid networks: those two live only here, not in the conv epilogues)
//             (nearest 2x upsample folded into the loads)                            (OP_WSUM: BiFPN fast normalised fusion, the
//             weights are constants at inference: relu(w_i) / (sum_j relu(w_j) + 1e-4))
// All HBM-bound streaming: thread = (pixel, 8-channel group), 16-byte loads (16-bit modes), fp32 arithmetic, one pass.
// se_gate is one workgroup per frame with a fixed summation order (stripes of pixels, then stripes in index order): results do not
// depend on scheduling, graph replay = eager bit for bit.
#include "kernels.h"
#include "elem16.h"

namespace adas {

struct SeDev {
    const void* in;
    float* gate;         // [n][gate_cs] (+ gate_coff) fp32
    const float* w1;     // [cr][c] then b1[cr]
    const float* w2;     // [c][cr] then b2[c]
    int in_cs, in_coff, gate_cs, gate_coff;
    int c, cr, HW, S;    // S: pixel stripes (1024 / (c / 8))
    float* part;         // two-launch form: [n][P][c] partial channel sums written by se_partial_kernel (null: this kernel sums the frame itself)
    int P;
    int act_hidden, act_gate;   // ACT_SILU | ACT_RELU; 0 (sigmoid) | ACT_HSIGMOID
};

// Partial channel sums of pixel range p of frame b (grid = n * P workgroups): one workgroup per frame leaves 64 of 256 CUs busy on a
// 64-frame batch and reads 4 MB through one CU's L1 on the large maps; P ranges per frame fill the chip.  Fixed order: stripes of a
// range are reduced in index order here, ranges in index order by the gate kernel.
template <typename T>
__global__ __launch_bounds__(256) void se_partial_kernel(SeDev d) {
    if constexpr (sizeof(T) == 2 && !__is_same(T, uint16_t)) Fp16::enter();
    extern __shared__ float sm[];   // [S][c]
    const int b = blockIdx.x / d.P, p = blockIdx.x - b * d.P, t = threadIdx.x, G = d.c >> 3;
    const int S = d.S, g = t % G, s = t / G;
    const int lo = (int)((long)d.HW * p / d.P), hi = (int)((long)d.HW * (p + 1) / d.P);
    if (s < S) {
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const T* base = (const T*)d.in + (size_t)b * d.HW * d.in_cs + d.in_coff + g * 8;
        for (int q = lo + s; q < hi; q += S) {
            float x[8];
            Vec8<T>::load(base + (size_t)q * d.in_cs, x);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += x[e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) sm[s * d.c + g * 8 + e] = acc[e];
    }
    __syncthreads();
    for (int ch = t; ch < d.c; ch += 256) {
        float m = 0.f;
        for (int q = 0; q < S; ++q) m += sm[q * d.c + ch];
        d.part[((size_t)b * d.P + p) * d.c + ch] = m;
    }
}

template <typename T>
__global__ __launch_bounds__(1024) void se_gate_kernel(SeDev d) {
    if constexpr (sizeof(T) == 2 && !__is_same(T, uint16_t)) Fp16::enter();
    extern __shared__ float sm[];
    float* part = sm;                 // [S][c]
    float* mean = sm + d.S * d.c;     // [c]
    float* hid = mean + d.c;          // [cr]
    const int n = blockIdx.x, t = threadIdx.x, G = d.c >> 3;
    const int g = t % G, s = t / G;
    if (d.part) {   // two-launch form: the ranges' partial sums, in range order
        for (int ch = t; ch < d.c; ch += 1024) {
            float m = 0.f;
            for (int q = 0; q < d.P; ++q) m += d.part[((size_t)n * d.P + q) * d.c + ch];
            mean[ch] = m / (float)d.HW;
        }
    } else {
        if (s < d.S) {
            float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            const T* base = (const T*)d.in + (size_t)n * d.HW * d.in_cs + d.in_coff + g * 8;
            for (int p = s; p < d.HW; p += d.S) {
                float x[8];
                Vec8<T>::load(base + (size_t)p * d.in_cs, x);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += x[e];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) part[s * d.c + g * 8 + e] = acc[e];
        }
        __syncthreads();
        for (int ch = t; ch < d.c; ch += 1024) {
            float m = 0.f;
            for (int q = 0; q < d.S; ++q) m += part[q * d.c + ch];
            mean[ch] = m / (float)d.HW;
        }
    }
    __syncthreads();
    const int lane = t & 63, wave = t >> 6;
    for (int j = wave; j < d.cr; j += 16) {
        float a = 0.f;
        for (int ch = lane; ch < d.c; ch += 64) a = fmaf(d.w1[(size_t)j * d.c + ch], mean[ch], a);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) a += __shfl_xor(a, o);
        if (lane == 0) {
            const float v = a + d.w1[(size_t)d.cr * d.c + j];
            hid[j] = d.act_hidden == ACT_RELU ? fmaxf(v, 0.0f) : v / (1.0f + expf(-v));
        }
    }
    __syncthreads();
    for (int ch = t; ch < d.c; ch += 1024) {
        float v = d.w2[(size_t)d.c * d.cr + ch];
        for (int j = 0; j < d.cr; ++j) v = fmaf(d.w2[(size_t)ch * d.cr + j], hid[j], v);
        d.gate[(size_t)n * d.gate_cs + d.gate_coff + ch] = d.act_gate == ACT_HSIGMOID ? fminf(fmaxf(v + 3.0f, 0.0f), 6.0f) / 6.0f : 1.0f / (1.0f + expf(-v));
    }
}

bool se_gate_supported(const TView& in, const TView& gate, int cr, uint64_t w_elems, uint64_t b_elems) {
    if (in.f32 || !gate.f32 || gate.h != 1 || gate.w != 1 || gate.c != in.c) return false;
    if ((in.c & 7) || (in.cs & 7) || (in.coff & 7) || in.c > 8192 || cr < 1 || cr > 2048) return false;
    return w_elems == (uint64_t)cr * in.c + cr && b_elems == (uint64_t)in.c * cr + in.c;
}

hipError_t launch_se_gate(const TView& in, const TView& gate, const float* w1, const float* w2, int cr, int act_hidden, int act_gate, int n, int prec, hipStream_t st,
                          const TView* scratch) {
    if ((act_hidden != 0 && act_hidden != ACT_SILU && act_hidden != ACT_RELU) || (act_gate != 0 && act_gate != ACT_HSIGMOID)) return hipErrorInvalidValue;
    SeDev d;
    d.act_hidden = act_hidden == ACT_RELU ? ACT_RELU : ACT_SILU; d.act_gate = act_gate;
    d.part = nullptr; d.P = 0;
    d.in = in.p; d.gate = (float*)gate.p; d.w1 = w1; d.w2 = w2;
    d.in_cs = in.cs; d.in_coff = in.coff; d.gate_cs = gate.cs; d.gate_coff = gate.coff;
    d.c = in.c; d.cr = cr; d.HW = in.h * in.w;
    const int G = in.c >> 3;
    if (scratch && scratch->p && scratch->f32 && scratch->cs == scratch->c && G <= 256 && scratch->c >= in.c && scratch->c % in.c == 0 &&
        d.HW >= 4 * (scratch->c / in.c)) {
        // two launches: P pixel ranges per frame (P = scratch channels / C), then the gate from the ranges' sums
        d.part = (float*)scratch->p; d.P = scratch->c / in.c;
        SeDev a = d;
        a.S = 256 / G;
        const size_t la = (size_t)a.S * d.c * 4;
        if (prec == PREC_FP32) hipLaunchKernelGGL(se_partial_kernel<float>, dim3(n * d.P), dim3(256), la, st, a);
        else if (prec == PREC_X3) hipLaunchKernelGGL(se_partial_kernel<x3s>, dim3(n * d.P), dim3(256), la, st, a);
        else if (prec == PREC_FP16) hipLaunchKernelGGL(se_partial_kernel<f16s>, dim3(n * d.P), dim3(256), la, st, a);
        else hipLaunchKernelGGL(se_partial_kernel<uint16_t>, dim3(n * d.P), dim3(256), la, st, a);
    }
    d.S = 1024 / G;
    if (d.S < 1) return hipErrorInvalidValue;
    if (d.S > d.HW) d.S = d.HW;
    const size_t lds = ((size_t)d.S * d.c + d.c + cr) * 4;
    if (lds > 64 * 1024) return hipErrorInvalidValue;   // (32 KB of striped sums + C + Cr floats: beyond 64 KB only for C > ~7,000 channels; refused, not a failed launch)
    if (prec == PREC_FP32) hipLaunchKernelGGL(se_gate_kernel<float>, dim3(n), dim3(1024), lds, st, d);
    else if (prec == PREC_X3) hipLaunchKernelGGL(se_gate_kernel<x3s>, dim3(n), dim3(1024), lds, st, d);
    else if (prec == PREC_FP16) hipLaunchKernelGGL(se_gate_kernel<f16s>, dim3(n), dim3(1024), lds, st, d);
    else hipLaunchKernelGGL(se_gate_kernel<uint16_t>, dim3(n), dim3(1024), lds, st, d);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------- scale
struct ScDev {
    const void* in;
    const float* gate;
    void* out;
    int in_cs, in_coff, out_cs, out_coff, gate_cs, gate_coff;
    int c, HW, n, small;
};

template <typename T>
__global__ __launch_bounds__(256) void scale_kernel(ScDev d) {
    if constexpr (sizeof(T) == 2 && !__is_same(T, uint16_t)) Fp16::enter();
    const int G = d.c >> 3;
    const size_t total = (size_t)d.n * d.HW * G;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        int g;
        size_t pix, b;
        if (d.small) {   // 32-bit divisions when the index space allows (a 64-bit one costs ~10x; uniform branch)
            const unsigned iu = (unsigned)i, pu = iu / (unsigned)G;
            g = (int)(iu - pu * (unsigned)G); pix = pu; b = pu / (unsigned)d.HW;
        } else {
            g = (int)(i % G); pix = i / G; b = pix / d.HW;
        }
        float x[8];
        Vec8<T>::load((const T*)d.in + pix * d.in_cs + d.in_coff + g * 8, x);
        const float* gp = d.gate + b * d.gate_cs + d.gate_coff + g * 8;
        const float4 g0 = *reinterpret_cast<const float4*>(gp), g1 = *reinterpret_cast<const float4*>(gp + 4);
        x[0] *= g0.x; x[1] *= g0.y; x[2] *= g0.z; x[3] *= g0.w; x[4] *= g1.x; x[5] *= g1.y; x[6] *= g1.z; x[7] *= g1.w;
        Vec8<T>::store((T*)d.out + pix * d.out_cs + d.out_coff + g * 8, x);
    }
}

bool scale_supported(const TView& in, const TView& gate, const TView& out) {
    if (in.f32 || out.f32 || !gate.f32 || gate.h != 1 || gate.w != 1 || gate.c != in.c || out.c != in.c || out.h != in.h || out.w != in.w) return false;
    return !((in.c & 7) || (in.cs & 7) || (in.coff & 7) || (out.cs & 7) || (out.coff & 7) || (gate.cs & 3) || (gate.coff & 3));
}

hipError_t launch_scale(const TView& in, const TView& gate, const TView& out, int n, int prec, hipStream_t st) {
    if (!scale_supported(in, gate, out)) return hipErrorInvalidValue;
    ScDev d;
    d.in = in.p; d.gate = (const float*)gate.p; d.out = out.p;
    d.in_cs = in.cs; d.in_coff = in.coff; d.out_cs = out.cs; d.out_coff = out.coff; d.gate_cs = gate.cs; d.gate_coff = gate.coff;
    d.c = in.c; d.HW = in.h * in.w; d.n = n;
    const size_t total = (size_t)n * d.HW * (in.c >> 3);
    d.small = total < ((size_t)1 << 31) ? 1 : 0;
    const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    if (prec == PREC_FP32) hipLaunchKernelGGL(scale_kernel<float>, dim3(blocks), dim3(256), 0, st, d);
    else if (prec == PREC_X3) hipLaunchKernelGGL(scale_kernel<x3s>, dim3(blocks), dim3(256), 0, st, d);
    else if (prec == PREC_FP16) hipLaunchKernelGGL(scale_kernel<f16s>, dim3(blocks), dim3(256), 0, st, d);
    else hipLaunchKernelGGL(scale_kernel<uint16_t>, dim3(blocks), dim3(256), 0, st, d);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------- channel shuffle
// torch channel_shuffle(x, g): x.view(B, g, C / g, H, W).transpose(1, 2).reshape(B, C, H, W), i.e. out[j * g + i] = in[i * (C / g) + j]
// (ShuffleNetV2 units: the YOLOv5-lite backbones of the reference's model table, README.md:95-108).  A pure permutation of storage
// elements: thread = (pixel, 8 output channels), eight element moves (2 bytes; 4 in fp32 mode; a (hi, lo) pair of halves in fp16x3).
struct ShDev {
    const void* in;
    void* out;
    int in_cs, in_coff, out_cs, out_coff;
    int c, g, npix;
};
template <int MODE>   // 0: 2-byte elements, 1: 4-byte elements, 2: fp16x3 G8 groups
__global__ __launch_bounds__(256) void shuffle_kernel(ShDev d) {
    const unsigned G = (unsigned)(d.c >> 3), total = (unsigned)d.npix * G;
    const int cpg = d.c / d.g;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const unsigned pix = i / G, g8 = i - pix * G;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int oc = (int)g8 * 8 + e, ic = (oc % d.g) * cpg + oc / d.g;
            if (MODE == 0) {
                ((uint16_t*)d.out)[(size_t)pix * d.out_cs + d.out_coff + oc] = ((const uint16_t*)d.in)[(size_t)pix * d.in_cs + d.in_coff + ic];
            } else if (MODE == 1) {
                ((uint32_t*)d.out)[(size_t)pix * d.out_cs + d.out_coff + oc] = ((const uint32_t*)d.in)[(size_t)pix * d.in_cs + d.in_coff + ic];
            } else {   // slot s of a G8 tensor: hi half at byte (s / 8) * 32 + (s % 8) * 2, lo half 16 bytes behind
                const size_t so = (size_t)pix * d.out_cs + d.out_coff + oc, si = (size_t)pix * d.in_cs + d.in_coff + ic;
                const uint16_t* ip = (const uint16_t*)d.in + (si >> 3) * 16 + (si & 7);
                uint16_t* op = (uint16_t*)d.out + (so >> 3) * 16 + (so & 7);
                op[0] = ip[0];
                op[8] = ip[8];
            }
        }
    }
}
bool shuffle_supported(const TView& in, const TView& out, int groups) {
    if (in.f32 || out.f32 || in.c != out.c || in.h != out.h || in.w != out.w || groups < 2 || in.c % groups) return false;
    return !((in.c & 7) || (in.cs & 7) || (in.coff & 7) || (out.cs & 7) || (out.coff & 7));
}
hipError_t launch_shuffle(const TView& in, const TView& out, int groups, int n, int prec, hipStream_t st) {
    if (!shuffle_supported(in, out, groups)) return hipErrorInvalidValue;
    const size_t npix = (size_t)n * in.h * in.w, total = npix * (in.c >> 3);
    if (total >= ((size_t)1 << 31)) return hipErrorInvalidValue;
    ShDev d{in.p, out.p, in.cs, in.coff, out.cs, out.coff, in.c, groups, (int)npix};
    const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    if (prec == PREC_FP32) hipLaunchKernelGGL(shuffle_kernel<1>, dim3(blocks), dim3(256), 0, st, d);
    else if (prec == PREC_X3) hipLaunchKernelGGL(shuffle_kernel<2>, dim3(blocks), dim3(256), 0, st, d);
    else hipLaunchKernelGGL(shuffle_kernel<0>, dim3(blocks), dim3(256), 0, st, d);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------- weighted sum
struct WsDev {
    const void* in[3];
    void* out;
    int cs[3], coff[3], half[3];   // half: the input has half the output's resolution
    float w[3];
    int out_cs, out_coff;
    int n_in, c, H, W, n, act, small;
};

template <typename T>
__global__ __launch_bounds__(256) void wsum_kernel(WsDev d) {
    if constexpr (sizeof(T) == 2 && !__is_same(T, uint16_t)) Fp16::enter();
    const int G = d.c >> 3;
    const size_t total = (size_t)d.n * d.H * d.W * G;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        int g, x, y;
        size_t pix, b;
        if (d.small) {
            const unsigned iu = (unsigned)i, pu = iu / (unsigned)G, tu = pu / (unsigned)d.W, bu = tu / (unsigned)d.H;
            g = (int)(iu - pu * (unsigned)G); x = (int)(pu - tu * (unsigned)d.W); y = (int)(tu - bu * (unsigned)d.H);
            pix = pu; b = bu;
        } else {
            g = (int)(i % G); pix = i / G; x = (int)(pix % d.W);
            const size_t t = pix / d.W;
            y = (int)(t % d.H); b = t / d.H;
        }
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (k >= d.n_in) break;
            const size_t sp = d.half[k] ? ((b * (d.H >> 1) + (y >> 1)) * (d.W >> 1) + (x >> 1)) : pix;
            float v[8];
            Vec8<T>::load((const T*)d.in[k] + sp * d.cs[k] + d.coff[k] + g * 8, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = fmaf(d.w[k], v[e], acc[e]);
        }
        if (d.act == ACT_SILU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = silu_for<T>(acc[e]);
        } else if (d.act == ACT_RELU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = fmaxf(acc[e], 0.0f);
        } else if (d.act == ACT_LEAKY) {   // LeakyReLU(0.1), the form conv_halo / conv_pw apply
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = fmaxf(acc[e], 0.1f * acc[e]);
        } else if (d.act == ACT_HSWISH) {  // torch.nn.Hardswish: x * relu6(x + 3) / 6
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = acc[e] * fminf(fmaxf(acc[e] + 3.0f, 0.0f), 6.0f) / 6.0f;
        } else if (d.act == ACT_HSIGMOID) {   // torch.nn.Hardsigmoid: relu6(x + 3) / 6
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = fminf(fmaxf(acc[e] + 3.0f, 0.0f), 6.0f) / 6.0f;
        } else if (d.act == ACT_RELU6) {
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = fminf(fmaxf(acc[e], 0.0f), 6.0f);
        }
        Vec8<T>::store((T*)d.out + pix * d.out_cs + d.out_coff + g * 8, acc);
    }
}

bool wsum_supported(int n_in, const TView* ins, const TView& out) {
    if (n_in < 1 || n_in > 3 || out.f32 || (out.c & 7) || (out.cs & 7) || (out.coff & 7)) return false;
    for (int k = 0; k < n_in; ++k) {
        const TView& v = ins[k];
        if (v.f32 || v.c != out.c || (v.cs & 7) || (v.coff & 7)) return false;
        const bool same = v.h == out.h && v.w == out.w, half = v.h * 2 == out.h && v.w * 2 == out.w;
        if (!same && !half) return false;
    }
    return true;
}

hipError_t launch_wsum(int n_in, const TView* ins, const float* w, const TView& out, int n, int act, int prec, hipStream_t st) {
    if (!wsum_supported(n_in, ins, out)) return hipErrorInvalidValue;
    if (act < ACT_NONE || act > ACT_RELU6) return hipErrorInvalidValue;   // never a silently dropped activation
    WsDev d;
    for (int k = 0; k < 3; ++k) {
        const TView& v = ins[k < n_in ? k : 0];
        d.in[k] = v.p; d.cs[k] = v.cs; d.coff[k] = v.coff; d.half[k] = v.h != out.h; d.w[k] = k < n_in ? w[k] : 0.0f;
    }
    d.out = out.p; d.out_cs = out.cs; d.out_coff = out.coff;
    d.n_in = n_in; d.c = out.c; d.H = out.h; d.W = out.w; d.n = n; d.act = act;
    const size_t total = (size_t)n * out.h * out.w * (out.c >> 3);
    d.small = total < ((size_t)1 << 31) ? 1 : 0;
    const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    if (prec == PREC_FP32) hipLaunchKernelGGL(wsum_kernel<float>, dim3(blocks), dim3(256), 0, st, d);
    else if (prec == PREC_X3) hipLaunchKernelGGL(wsum_kernel<x3s>, dim3(blocks), dim3(256), 0, st, d);
    else if (prec == PREC_FP16) hipLaunchKernelGGL(wsum_kernel<f16s>, dim3(blocks), dim3(256), 0, st, d);
    else hipLaunchKernelGGL(wsum_kernel<uint16_t>, dim3(blocks), dim3(256), 0, st, d);
    return hipGetLastError();
}

}  // namespace adas
