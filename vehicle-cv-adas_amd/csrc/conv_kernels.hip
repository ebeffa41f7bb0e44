// conv_kernels.hip -- NHWC implicit-GEMM convolution on the CDNA4 matrix cores.
//
//   out[m][co] = act( bias[co] + sum_{r,s,c} in[pix(m)+(r,s)][c] * w[co][r][s][c] ) (+ residual)
//
// GEMM view: M = N*Ho*Wo output pixels, N = Cout, K = kh*kw*Cin (tap-major, channel-minor, so
// every 8-element K chunk is 16 contiguous bytes of one input pixel -- no im2col buffer exists).
// A workgroup (4 waves) owns a BM x BN tile; per 32-deep K step it gathers the BM x 32 activation
// slab and the BN x 32 weight slab into padded LDS rows (80 B: conflict-free ds_read_b128), and
// each wave runs 16x16x32 bf16 MFMAs (fp32 mode: 16x16x4 f32 MFMAs) with the WEIGHTS as the MFMA
// A operand, so a lane ends up holding 4 consecutive output channels of one pixel and the fused
// epilogue (bias + SiLU/ReLU + residual + concat-by-offset) stores 8/16 contiguous bytes per lane.
// Global loads of step k+1 are issued before the MFMAs of step k (register prefetch, 2 LDS buffers,
// one barrier per step).  Roofline: MFMA-bound for the ResNet stages, HBM/L2-bound for the
// 16..64-channel YOLOv8n layers (arithmetic intensity < 312 FLOP/B).
#include "kernels.h"
#include "elem16.h"
#include <stdlib.h>

namespace adas {

typedef __attribute__((ext_vector_type(4))) float f32x4;

// Element types of this generic kernel: `uint16_t` = bf16 bits, `f16s` = IEEE-half bits (elem16.h), `float` = the fp32 parity mode.

struct alignas(16) U4 {
    uint32_t x, y, z, w;
};

template <typename T>
struct Chunk;  // 8 K-elements
template <>
struct Chunk<uint16_t> {
    U4 v;
    __device__ void zero() { v = U4{0, 0, 0, 0}; }
    __device__ void load(const uint16_t* p) { v = *reinterpret_cast<const U4*>(p); }
    __device__ void store(uint16_t* p) const { *reinterpret_cast<U4*>(p) = v; }
};
template <>
struct Chunk<f16s> {
    U4 v;
    __device__ void zero() { v = U4{0, 0, 0, 0}; }
    __device__ void load(const f16s* p) { v = *reinterpret_cast<const U4*>(p); }
    __device__ void store(f16s* p) const { *reinterpret_cast<U4*>(p) = v; }
};
template <>
struct Chunk<float> {
    U4 v[2];
    __device__ void zero() { v[0] = v[1] = U4{0, 0, 0, 0}; }
    __device__ void load(const float* p) {
        v[0] = reinterpret_cast<const U4*>(p)[0];
        v[1] = reinterpret_cast<const U4*>(p)[1];
    }
    __device__ void store(float* p) const {
        reinterpret_cast<U4*>(p)[0] = v[0];
        reinterpret_cast<U4*>(p)[1] = v[1];
    }
};

__device__ __forceinline__ f32x4 mma(const Chunk<uint16_t>& w, const Chunk<uint16_t>& x, f32x4 acc) {
    return Bf16::mfma(__builtin_bit_cast(e_u32x4, w.v), __builtin_bit_cast(e_u32x4, x.v), acc);
}
__device__ __forceinline__ f32x4 mma(const Chunk<f16s>& w, const Chunk<f16s>& x, f32x4 acc) {
    return Fp16::mfma(__builtin_bit_cast(e_u32x4, w.v), __builtin_bit_cast(e_u32x4, x.v), acc);
}
__device__ __forceinline__ f32x4 mma(const Chunk<float>& w, const Chunk<float>& x, f32x4 acc) {
    const float* a = reinterpret_cast<const float*>(w.v);
    const float* b = reinterpret_cast<const float*>(x.v);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], acc, 0, 0, 0);
    return acc;
}

__device__ __forceinline__ float bf2f(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }
__device__ __forceinline__ uint16_t f2bf(float f) {
    uint32_t u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);  // round to nearest even
    return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float ldf(const float* p) { return *p; }
__device__ __forceinline__ float ldf(const uint16_t* p) { return bf2f(*p); }
__device__ __forceinline__ float ldf(const f16s* p) { return Fp16::to_f32(p->v); }
__device__ __forceinline__ void stf(float* p, float v) { *p = v; }
__device__ __forceinline__ void stf(uint16_t* p, float v) { *p = f2bf(v); }
__device__ __forceinline__ void stf(f16s* p, float v) { p->v = Fp16::from_f32(v); }

__device__ __forceinline__ void load4(const float* p, float o[4]) {
    float4 q = *reinterpret_cast<const float4*>(p);
    o[0] = q.x; o[1] = q.y; o[2] = q.z; o[3] = q.w;
}
__device__ __forceinline__ void load4(const uint16_t* p, float o[4]) {
    uint2 q = *reinterpret_cast<const uint2*>(p);
    o[0] = __uint_as_float(q.x << 16); o[1] = __uint_as_float(q.x & 0xffff0000u);
    o[2] = __uint_as_float(q.y << 16); o[3] = __uint_as_float(q.y & 0xffff0000u);
}
__device__ __forceinline__ void load4(const f16s* p, float o[4]) {
    uint2 q = *reinterpret_cast<const uint2*>(p);
    o[0] = Fp16::lo(q.x); o[1] = Fp16::hi(q.x);
    o[2] = Fp16::lo(q.y); o[3] = Fp16::hi(q.y);
}
__device__ __forceinline__ void store4(f16s* p, const float v[4]) {
    *reinterpret_cast<uint2*>(p) = make_uint2(Fp16::pack2(v[0], v[1]), Fp16::pack2(v[2], v[3]));
}
__device__ __forceinline__ void store4(float* p, const float v[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void store4(uint16_t* p, const float v[4]) {
    uint2 q;
    q.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
    q.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
    *reinterpret_cast<uint2*>(p) = q;
}

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == ACT_SILU) return v / (1.0f + expf(-v));
    if (act == ACT_RELU) return fmaxf(v, 0.0f);
    if (act == ACT_LEAKY) return fmaxf(v, 0.1f * v);
    return v;
}

#define ADAS_MAX_Q 1152  // K/8 chunks: 3*3*1024/8 (YOLOv8m/x and YOLOv5x reach 3*3*640/8 = 720; UFLDv2 cls.1 4000/8 = 500)

struct ConvDev {
    const void* in;
    const void* wgt;
    const float* bias;
    void* out;
    const void* res;
    int in_cs, in_coff, cin, H, W;
    int out_cs, out_coff, cout, Ho, Wo;
    int res_cs, res_coff, res_mode;
    int kh, kw, stride, pad, act;
    int nq, kpad, M;
};

// T = activation/weight element (uint16_t = bf16 bits, or float); OutT = output element; ResT = residual element
template <typename T, typename OutT, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvDev a) {
    if constexpr (sizeof(OutT) == 2 && !__is_same(OutT, uint16_t)) Fp16::enter();   // half stores saturate (elem16.h)
    constexpr int PADE = 16 / (int)sizeof(T);
    constexpr int LDK = 32 + PADE;
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    constexpr int A_IT = (BM * 4 + 255) / 256, B_IT = (BN * 4 + 255) / 256;
    static_assert(WM * WN == 4 && TM >= 1 && TN >= 1, "tile");
    __shared__ __attribute__((aligned(16))) T As[2][BM][LDK];
    __shared__ __attribute__((aligned(16))) T Bs[2][BN][LDK];
    __shared__ int ktab[ADAS_MAX_Q];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const T* __restrict__ in = (const T*)a.in;
    const T* __restrict__ wgt = (const T*)a.wgt;

    // K chunk table: q -> (r, s, c8)
    const int cin8 = a.cin >> 3;
    for (int q = tid; q < a.nq; q += 256) {
        int tap = q / cin8, c8 = q - tap * cin8;
        int r = tap / a.kw, s = tap - r * a.kw;
        ktab[q] = (r << 24) | (s << 16) | c8;
    }
    // per-thread gather rows
    const int kc = tid & 3;
    int iy0[A_IT], ix0[A_IT], pb[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        int row = (tid >> 2) + 64 * i;
        int m = m0 + row;
        bool ok = (row < BM) && (m < a.M);
        int mm = ok ? m : 0;
        int hw = a.Ho * a.Wo;
        int n = mm / hw, rem = mm - n * hw;
        int oy = rem / a.Wo, ox = rem - oy * a.Wo;
        iy0[i] = oy * a.stride - a.pad;
        ix0[i] = ox * a.stride - a.pad;
        pb[i] = ok ? n * a.H * a.W : -1;
    }
    __syncthreads();

    const int KT = a.kpad >> 5;
    Chunk<T> ra[A_IT], rb[B_IT];

    auto gload = [&](int ks) {
        const int q = ks * 4 + kc;
        int r = 0, s = 0, c8 = 0;
        const bool qok = q < a.nq;
        if (qok) {
            int e = ktab[q];
            r = e >> 24;
            s = (e >> 16) & 0xff;
            c8 = e & 0xffff;
        }
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            int iy = iy0[i] + r, ix = ix0[i] + s;
            bool ok = qok && pb[i] >= 0 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            if (ok)
                ra[i].load(in + ((size_t)(pb[i] + iy * a.W + ix) * a.in_cs + a.in_coff + c8 * 8));
            else
                ra[i].zero();
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            int row = (tid >> 2) + 64 * i;
            if (row < BN) rb[i].load(wgt + ((size_t)(n0 + row) * a.kpad + ks * 32 + kc * 8));
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            int row = (tid >> 2) + 64 * i;
            if (row < BM) ra[i].store(&As[buf][row][kc * 8]);
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            int row = (tid >> 2) + 64 * i;
            if (row < BN) rb[i].store(&Bs[buf][row][kc * 8]);
        }
    };

    f32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int lrow = lane & 15, kg = lane >> 4;
    gload(0);
    lstore(0);
    __syncthreads();
    for (int ks = 0; ks < KT; ++ks) {
        const int buf = ks & 1;
        if (ks + 1 < KT) gload(ks + 1);
        Chunk<T> wf[TN], xf[TM];
#pragma unroll
        for (int i = 0; i < TN; ++i) wf[i].load(&Bs[buf][(wn * TN + i) * 16 + lrow][kg * 8]);
#pragma unroll
        for (int j = 0; j < TM; ++j) xf[j].load(&As[buf][(wm * TM + j) * 16 + lrow][kg * 8]);
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j) acc[i][j] = mma(wf[i], xf[j], acc[i][j]);
        if (ks + 1 < KT) lstore(buf ^ 1);
        __syncthreads();
    }

    // ---- fused epilogue: lane holds channels c..c+3 of pixel m
    OutT* __restrict__ out = (OutT*)a.out;
    const bool vec_ok = ((a.cout & 3) == 0) && ((a.out_cs & 3) == 0) && ((a.out_coff & 3) == 0);
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int m = m0 + (wm * TM + j) * 16 + lrow;
        if (m >= a.M) continue;
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            const int c = n0 + (wn * TN + i) * 16 + kg * 4;
            if (c >= a.cout) continue;
            float v[4], b[4];
            load4(a.bias + c, b);
#pragma unroll
            for (int t = 0; t < 4; ++t) v[t] = acc[i][j][t] + b[t];
            if (a.res_mode != RES_NONE) {
                const T* rp = (const T*)a.res + ((size_t)m * a.res_cs + a.res_coff + c);
                float rv[4];
                load4(rp, rv);
                if (a.res_mode == RES_BEFORE_ACT) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) v[t] = apply_act(v[t] + rv[t], a.act);
                } else {
#pragma unroll
                    for (int t = 0; t < 4; ++t) v[t] = apply_act(v[t], a.act) + rv[t];
                }
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) v[t] = apply_act(v[t], a.act);
            }
            OutT* op = out + ((size_t)m * a.out_cs + a.out_coff + c);
            if (vec_ok && c + 3 < a.cout) {
                store4(op, v);
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    if (c + t < a.cout) stf(op + t, v[t]);
            }
        }
    }
}

// -------------------------------------------------------------------------------------
struct Tile {
    int bm, bn;
};
int persist_slots(int which) {
    // which: 0 conv_halo8, 1 conv_halo_rw, 2 conv_stem.  ADAS_PERSIST_SLOTS sets all three, ADAS_PERSIST_SLOTS_H8 / _RW / _STEM one
    static int v[3] = {-1, -1, -1};
    if (v[0] < 0) {
        const char* all = getenv("ADAS_PERSIST_SLOTS");
        const char* names[3] = {"ADAS_PERSIST_SLOTS_H8", "ADAS_PERSIST_SLOTS_RW", "ADAS_PERSIST_SLOTS_STEM"};
        for (int i = 2; i >= 0; --i) {
            const char* e = getenv(names[i]);
            int x = e ? atoi(e) : (all ? atoi(all) : 32);
            v[i] = (x < 4 || x > 32) ? 32 : x;
        }
    }
    return v[which < 0 || which > 2 ? 0 : which];
}

static Tile pick_tile(const ConvArgs& a, int prec) {
    int bn;
    if (a.out.c <= 16) bn = 16;
    else if (a.out.c <= 32) bn = 32;
    else if (a.out.c <= 64 || prec == PREC_FP32 || (a.out.c % 128 != 0 && a.out.c < 256)) bn = 64;
    else bn = 128;
    // enough workgroups to cover 256 CUs a few times over, otherwise the smaller M tile
    long tiles128 = (long)((a.m + 127) / 128) * ((a.out.c + bn - 1) / bn);
    int bm = (tiles128 >= 512) ? 128 : 64;
    if (bn == 128 && bm == 64) bn = 64;
    return Tile{bm, bn};
}

const char* conv_tile_name(const ConvArgs& a, int prec) {
    Tile t = pick_tile(a, prec);
    static thread_local char buf[32];
    snprintf(buf, sizeof(buf), "%dx%d", t.bm, t.bn);
    return buf;
}

// Name of the kernel instantiation a conv launch resolves to (as rocprofv3 --kernel-trace prints it, minus namespaces).
const char* conv_kernel_name(const ConvArgs& a, int prec, int kernel) {
    static thread_local char buf[96];
    if (prec == PREC_X3 && kernel == CONV_STEM) {
        snprintf(buf, sizeof(buf), "conv_stem_x3_kernel<%d,%d,%s>", a.kh, (a.out.c + 15) / 16, a.act == ACT_SILU ? "SILU" : (a.act == ACT_RELU ? "RELU" : "LEAKY"));
        return buf;
    }
    if (prec == PREC_X3 && kernel == CONV_FC) return "fc_x3_kernel";
    if (prec == PREC_X3 && kernel == CONV_PW) {
        snprintf(buf, sizeof(buf), "conv_pwx3_kernel<%d>", (a.in.c + 31) / 32);
        return buf;
    }
    if (prec == PREC_X3) {
        if (a.wgt_h8x3 && halo8_x3_applicable(a.kh, a.kw, a.stride, a.pad, a.n, a.in, a.out, a.res, a.res_mode)) {
            snprintf(buf, sizeof(buf), "conv_h8x3_kernel<%s>", a.act == ACT_SILU ? "SILU" : (a.act == ACT_RELU ? "RELU" : (a.act == ACT_LEAKY ? "LEAKY" : "NONE")));
            return buf;
        }
        if (a.wgt_h8x3 && halo_s2p_x3_applicable(a.kh, a.kw, a.stride, a.pad, a.res_mode, a.n, a.in, a.out)) {
            snprintf(buf, sizeof(buf), "%s<%s>", halo_s2d_x3_applicable(a.kh, a.kw, a.stride, a.pad, a.res_mode, a.n, a.in, a.out) ? "conv_s2d_x3_kernel" : "conv_s2p_x3_kernel",
                     a.act == ACT_SILU ? "SILU" : (a.act == ACT_RELU ? "RELU" : (a.act == ACT_LEAKY ? "LEAKY" : "NONE")));
            return buf;
        }
        return conv_x3_kernel_name(a);
    }
    const char* actn = a.act == ACT_SILU ? "SILU" : (a.act == ACT_RELU ? "RELU" : (a.act == ACT_LEAKY ? "LEAKY" : "NONE"));
    if (kernel == CONV_HALO && a.halo_bn > 0 && a.halo_bn != halo_bn(a.out.c)) {
        snprintf(buf, sizeof(buf), a.stride == 1 && halo_tile_pixels(a) == 128 ? "conv_halo_kernel<%d,%s,s%d,bm128>" : "conv_halo_kernel<%d,%s,s%d>", a.halo_bn, actn, a.stride);
    } else if (kernel == CONV_HALO && halo_rw_applicable(a.kh, a.kw, a.stride, a.pad, a.n, a.in, a.out)) {
        snprintf(buf, sizeof(buf), a.out.c <= 32 ? "conv_halo_rw_kernel<%d,%s,bn32>" : "conv_halo_rw_kernel<%d,%s>", (a.in.c + 31) / 32, actn);
    } else if (kernel == CONV_HALO && halo_s2p_applicable(a.kh, a.kw, a.stride, a.pad, a.res_mode, a.n, a.in, a.out)) {
        snprintf(buf, sizeof(buf), "conv_s2p_kernel<%s>", actn);
    } else if (kernel == CONV_HALO && halo8_applicable(a.kh, a.kw, a.stride, a.pad, a.n, a.in, a.out, a.res, a.res_mode)) {
        snprintf(buf, sizeof(buf), "conv_h8_kernel<%s>", actn);
    } else if (kernel == CONV_HALO) {
        snprintf(buf, sizeof(buf), a.stride == 1 && halo_tile_pixels(a) == 128 ? "conv_halo_kernel<%d,%s,s%d,bm128>" : "conv_halo_kernel<%d,%s,s%d>",
                 halo_bn(a.out.c), actn, a.stride);
    } else if (kernel == CONV_FC) {
        snprintf(buf, sizeof(buf), "fc_kernel");
    } else if (kernel == CONV_PW) {
        snprintf(buf, sizeof(buf), "conv_pw_kernel<%d>", (a.in.c + 31) / 32);
    } else if (kernel == CONV_STEM) {
        snprintf(buf, sizeof(buf), "conv_stem_kernel<%d,%d,%s>", a.kh, (a.out.c + 15) / 16, actn);
    } else if (kernel == CONV_GATHER && pwg_applicable(prec, a.kh, a.kw, a.stride, a.pad, a.in, a.out, a.res, a.res_mode)) {
        snprintf(buf, sizeof(buf), "%s", pwg_kernel_name(a.m, a.out.c));
    } else {
        Tile t = pick_tile(a, prec);
        const bool of32 = a.out.f32 || prec == PREC_FP32;
        const char* en = prec == PREC_FP32 ? "f32" : (prec == PREC_FP16 ? "f16" : "bf16");
        snprintf(buf, sizeof(buf), "conv_igemm_kernel<%s,%s,%d,%d>", en, of32 ? "f32" : en, t.bm, t.bn);
    }
    return buf;
}

template <typename T, typename OutT>
static hipError_t launch_typed(const ConvDev& d, Tile t, hipStream_t st) {
    dim3 grid((d.M + t.bm - 1) / t.bm, (d.cout + t.bn - 1) / t.bn);
#define LAUNCH(BM_, BN_, WM_, WN_)                                                                         \
    if (t.bm == BM_ && t.bn == BN_) {                                                                       \
        hipLaunchKernelGGL((conv_igemm_kernel<T, OutT, BM_, BN_, WM_, WN_>), grid, dim3(256), 0, st, d);    \
        return hipGetLastError();                                                                           \
    }
    if constexpr (sizeof(T) == 2) { LAUNCH(128, 128, 2, 2) }
    LAUNCH(128, 64, 2, 2)
    LAUNCH(128, 32, 4, 1)
    LAUNCH(128, 16, 4, 1)
    LAUNCH(64, 64, 2, 2)
    LAUNCH(64, 32, 2, 2)
    LAUNCH(64, 16, 4, 1)
#undef LAUNCH
    return hipErrorInvalidValue;
}

hipError_t launch_conv_halo(const ConvArgs& a, hipStream_t st);  // conv_halo.hip

static bool halo_enabled() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("ADAS_NO_HALO");
        v = (e && e[0] == '1') ? 0 : 1;
    }
    return v == 1;
}

bool halo_applicable(int kh, int kw, int stride, int pad, const TView& in, const TView& out);  // conv_halo.hip
bool fc_applicable(int prec, int kh, int kw, int stride, int max_n, const TView& in, const TView& out);  // conv_fc.hip
hipError_t launch_fc(const ConvArgs& a, hipStream_t st);
bool pw_applicable(int prec, int kh, int kw, int stride, int pad, int res_mode, const TView& in, const TView& out);  // conv_pw.hip
hipError_t launch_conv_pw(const ConvArgs& a, hipStream_t st);
static bool pw_enabled() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("ADAS_NO_PW");
        v = (e && e[0] == '1') ? 0 : 1;
    }
    return v == 1;
}
static bool fc_enabled() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("ADAS_NO_FC");
        v = (e && e[0] == '1') ? 0 : 1;
    }
    return v == 1;
}

ConvPlan plan_conv(int prec, int kh, int kw, int stride, int pad, int max_n, int res_mode, const TView& in, const TView& out) {
    ConvPlan p;
    if (prec == PREC_X3) {   // split precision: the two streaming kernels (conv_pw_x3.hip), else the generic one (conv_x3.hip), K = (tap, channel) in G8 groups
        p.kernel = (fc_enabled() && fc_x3_applicable(kh, kw, stride, in, out))                    ? CONV_FC
                   : (pw_enabled() && pw_x3_applicable(kh, kw, stride, pad, res_mode, in, out)) ? CONV_PW
                                                                                                : CONV_GATHER;
        p.cin_pad = in.c;
        p.kpad = (kh * kw * p.cin_pad + 31) / 32 * 32;
        return p;
    }
    if (fc_enabled() && fc_applicable(prec, kh, kw, stride, max_n, in, out)) {
        p.kernel = CONV_FC;
        p.cin_pad = in.c;
    } else if (pw_enabled() && pw_applicable(prec, kh, kw, stride, pad, res_mode, in, out)) {
        p.kernel = CONV_PW;
        p.cin_pad = in.c;
    } else if (prec_is16(prec) && halo_enabled() && halo_applicable(kh, kw, stride, pad, in, out)) {
        p.kernel = CONV_HALO;
        p.cin_pad = (in.c + 31) / 32 * 32;
    } else {
        p.kernel = CONV_GATHER;
        p.cin_pad = in.c;
    }
    p.kpad = (kh * kw * p.cin_pad + 31) / 32 * 32;
    return p;
}

hipError_t launch_conv(const ConvArgs& a, int prec, hipStream_t st) {
    ConvPlan pl = plan_conv(prec, a.kh, a.kw, a.stride, a.pad, a.max_n, a.res_mode, a.in, a.out);
    if (pl.kpad != a.kpad) return hipErrorInvalidValue;  // weights were packed for a different plan
    if (prec == PREC_X3) {
        if ((a.up_c > 0 && pl.kernel != CONV_PW) || a.ds_w) return hipErrorInvalidValue;
        if (pl.kernel == CONV_FC) return launch_fc_x3(a, st);
        if (pl.kernel == CONV_PW) return launch_conv_pw_x3(a, st);
        if (a.wgt_h8x3 && halo8_x3_applicable(a.kh, a.kw, a.stride, a.pad, a.n, a.in, a.out, a.res, a.res_mode)) {
            hipError_t e = launch_conv_halo8_x3(a, st);
            if (e != hipErrorNotSupported) return e;
        }
        if (a.wgt_h8x3 && halo_s2p_x3_applicable(a.kh, a.kw, a.stride, a.pad, a.res_mode, a.n, a.in, a.out)) {
            hipError_t e = launch_conv_s2p_x3(a, st);
            if (e != hipErrorNotSupported) return e;
        }
        return launch_conv_x3(a, st);
    }
    if (a.up_c > 0 && pl.kernel != CONV_PW) return hipErrorInvalidValue;   // only conv_pw fetches the folded upsample's channels
    if (a.ds_w) {   // the engine dropped the projection's launch: only conv_halo8 computes it inside this conv
        if (pl.kernel != CONV_HALO) return hipErrorInvalidValue;
        hipError_t e = launch_conv_halo8(a, st);
        return e == hipErrorNotSupported ? hipErrorInvalidValue : e;
    }
    if (pl.kernel == CONV_HALO && a.halo_bn > 0 && a.halo_bn != halo_bn(a.out.c)) return launch_conv_halo(a, st);   // packed for narrower blocks
    if (pl.kernel == CONV_HALO) {
        if (halo_rw_applicable(a.kh, a.kw, a.stride, a.pad, a.n, a.in, a.out)) {
            hipError_t e = launch_conv_halo_rw(a, st);
            if (e != hipErrorNotSupported) return e;  // e.g. a residual view that is not 16-byte aligned: same packing, other kernel
        }
        if (halo_s2p_applicable(a.kh, a.kw, a.stride, a.pad, a.res_mode, a.n, a.in, a.out)) {
            hipError_t e = launch_conv_halo_s2p(a, st);
            if (e != hipErrorNotSupported) return e;
        }
        if (halo8_applicable(a.kh, a.kw, a.stride, a.pad, a.n, a.in, a.out, a.res, a.res_mode)) {
            hipError_t e = launch_conv_halo8(a, st);
            if (e != hipErrorNotSupported) return e;
        }
        return launch_conv_halo(a, st);
    }
    if (pl.kernel == CONV_PW) return launch_conv_pw(a, st);
    if (pl.kernel == CONV_FC) return a.res_mode == RES_NONE ? launch_fc(a, st) : hipErrorInvalidValue;
    if (pwg_applicable(prec, a.kh, a.kw, a.stride, a.pad, a.in, a.out, a.res, a.res_mode)) {   // wide 1x1: same packing, K-looped GEMM kernel
        hipError_t e = launch_conv_pwg(a, st);
        if (e != hipErrorNotSupported) return e;
    }
    ConvDev d;
    d.in = a.in.p; d.wgt = a.wgt; d.bias = a.bias; d.out = a.out.p; d.res = a.res.p;
    d.in_cs = a.in.cs; d.in_coff = a.in.coff; d.cin = a.in.c; d.H = a.in.h; d.W = a.in.w;
    d.out_cs = a.out.cs; d.out_coff = a.out.coff; d.cout = a.out.c; d.Ho = a.out.h; d.Wo = a.out.w;
    d.res_cs = a.res.cs; d.res_coff = a.res.coff; d.res_mode = a.res_mode;
    d.kh = a.kh; d.kw = a.kw; d.stride = a.stride; d.pad = a.pad; d.act = a.act;
    d.nq = a.k / 8; d.kpad = a.kpad; d.M = a.m;
    if (d.nq > ADAS_MAX_Q || (a.in.c & 7) || (a.in.cs & 7) || (a.in.coff & 7)) return hipErrorInvalidValue;
    if (a.in.f32 && prec != PREC_FP32) return hipErrorInvalidValue;  // conv inputs are always in the compute type
    Tile t = pick_tile(a, prec);
    const bool out_f32 = a.out.f32 || prec == PREC_FP32;
    if (prec == PREC_FP32) return launch_typed<float, float>(d, t, st);
    if (prec == PREC_FP16) return out_f32 ? launch_typed<f16s, float>(d, t, st) : launch_typed<f16s, f16s>(d, t, st);
    if (out_f32) return launch_typed<uint16_t, float>(d, t, st);
    return launch_typed<uint16_t, uint16_t>(d, t, st);
}

// -------------------------------------------------------------------------------------
// weight packing: fp32 [cout][taps][cin] -> T [cout_pad][kpad], element (row, tap*cin_pad + c), zero padded
template <typename T>
__global__ void pack_weights_kernel(const float* __restrict__ src, T* __restrict__ dst, int cout, int taps, int cin, int cin_pad,
                                    int kpad, size_t total) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        size_t row = i / kpad;
        int col = (int)(i - row * kpad);
        int tap = col / cin_pad, c = col - tap * cin_pad;
        float v = (row < (size_t)cout && tap < taps && c < cin) ? src[(row * taps + tap) * cin + c] : 0.0f;
        stf(dst + i, v);
    }
}

// fp32 [cout][9][cin] -> 16-bit [cout_pad/BN][cin_pad/32][9][BN][32], zero padded (halo kernels, BN = halo_bn(cout))
template <typename T>
__global__ void pack_weights_halo_kernel(const float* __restrict__ src, T* __restrict__ dst, int cout, int cin, int cin_pad, int bn,
                                         size_t total) {
    const int nchunk = cin_pad >> 5;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i & 31);
        size_t r = i >> 5;
        const int nl = (int)(r % bn); r /= bn;
        const int tap = (int)(r % 9); r /= 9;
        const int chunk = (int)(r % nchunk);
        const size_t tile = r / nchunk;
        const size_t n = tile * bn + nl;
        const int ch = chunk * 32 + c;
        const float v = (n < (size_t)cout && ch < cin) ? src[(n * 9 + tap) * cin + ch] : 0.0f;
        stf(dst + i, v);
    }
}
hipError_t launch_pack_weights_halo(const float* src, void* dst, int cout, int cout_pad, int cin, int cin_pad, int prec, hipStream_t st, int bn) {
    if (bn <= 0) bn = halo_bn(cout);
    const size_t total = (size_t)cout_pad * 9 * cin_pad;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    if (prec == PREC_FP16)
        hipLaunchKernelGGL(pack_weights_halo_kernel<f16s>, dim3(blocks), dim3(256), 0, st, src, (f16s*)dst, cout, cin, cin_pad, bn, total);
    else
        hipLaunchKernelGGL(pack_weights_halo_kernel<uint16_t>, dim3(blocks), dim3(256), 0, st, src, (uint16_t*)dst, cout, cin, cin_pad, bn, total);
    return hipGetLastError();
}

hipError_t launch_pack_weights(const float* src, void* dst, int cout, int cout_pad, int taps, int cin, int cin_pad, int kpad, int prec,
                               hipStream_t st) {
    if (prec == PREC_X3) return launch_pack_weights_x3(src, dst, cout, cout_pad, taps, cin, cin_pad, kpad, st);
    size_t total = (size_t)cout_pad * kpad;
    int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    if (prec == PREC_FP32)
        hipLaunchKernelGGL(pack_weights_kernel<float>, dim3(blocks), dim3(256), 0, st, src, (float*)dst, cout, taps, cin, cin_pad, kpad, total);
    else if (prec == PREC_FP16)
        hipLaunchKernelGGL(pack_weights_kernel<f16s>, dim3(blocks), dim3(256), 0, st, src, (f16s*)dst, cout, taps, cin, cin_pad, kpad, total);
    else
        hipLaunchKernelGGL(pack_weights_kernel<uint16_t>, dim3(blocks), dim3(256), 0, st, src, (uint16_t*)dst, cout, taps, cin, cin_pad, kpad, total);
    return hipGetLastError();
}

}  // namespace adas
