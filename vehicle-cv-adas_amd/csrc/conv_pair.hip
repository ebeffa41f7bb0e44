// conv_pair.hip -- two chained stride-1 3x3 convolutions on 16 or 32 channels in ONE launch (the Bottleneck of YOLOv8's C2f blocks
// at its two largest resolutions: model.2 at 160x160x16, model.4 / model.15 at 80x80x32):
//     y = SiLU(conv2(SiLU(conv1(x)))) [+ x]
// As separate launches each of these layers moves its 16 / 32-channel tensor through HBM twice for a handful of MFMAs (2 x 7.5 GFLOP
// at 64 frames against 2 x 52 MB in and out): 65 us per launch at 160x160, three times the HBM floor.  Here a workgroup owns a
// 16 x 16 output tile: it stages the 20 x 20 input window once, computes conv1 on the 18 x 18 region conv2 needs (27 % more conv1
// work, irrelevant at this arithmetic intensity), keeps that intermediate in LDS (zeroed outside the image: conv2's padding is
// applied to conv1's OUTPUT domain), runs conv2 from it and takes the shortcut operand from the window it already holds -- the
// intermediate tensor never exists in memory and x is read once.
// MFMA mapping: weights are the A operand (16 output channels x 32 K), pixels the B operand.  32 channels: one K step per tap.
// 16 channels: one K step per PAIR of taps (lanes kg 0-1 take tap 2s, kg 2-3 tap 2s+1; the tenth half-step multiplies packed
// zeros).  Both convs' weights are register-resident fragments (20 / 72 VGPRs each), loaded once per workgroup from an L2-hot
// fragment-ordered array.
#include "kernels.h"
#include "elem16.h"
#include <stdlib.h>

namespace adas {

typedef __attribute__((ext_vector_type(4))) float pf32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t pu32x4;

struct PairDev {
    const uint16_t* in;
    uint16_t* out;
    const uint16_t* w1;
    const uint16_t* w2;
    const float* b1;
    const float* b2;
    uint32_t in_bytes;
    int in_cs, in_coff, out_cs, out_coff;
    int H, W, has_res;
    int tiles_x, tiles_per_img;
};

constexpr int PAIR_T = 16;                        // output tile edge
constexpr int PAIR_IW = PAIR_T + 2;               // intermediate region edge (conv2's halo)
constexpr int PAIR_WW = PAIR_T + 4;               // input window edge
constexpr int PAIR_NI = PAIR_IW * PAIR_IW;        // 324 intermediate pixels
constexpr int PAIR_G1 = (PAIR_NI + 15) / 16;      // 21 sixteen-pixel groups of them
constexpr int PAIR_NW = PAIR_WW * PAIR_WW;        // 400 window pixels
constexpr uint32_t PAIR_OOB = 0x80000000u;

__device__ __forceinline__ float pair_silu(float v) { return v * fast_rcp(1.0f + __expf(-v)); }

// 16-byte position of channel chunk c of pixel p: 64-byte pixels (32 channels) use conv_halo's conflict-free XOR swizzle
template <int C>
__device__ __forceinline__ int pair_pos(int p, int c) {
    return C == 32 ? (c ^ (((p >> 2) & 1) << 1)) : c;
}

template <typename E, int C>
__global__ __launch_bounds__(256, C == 16 ? 6 : 3) void conv_pair_kernel(PairDev a) {
    E::enter();
    typedef typename E::vec8 vec8;
    constexpr int NT = C / 16;               // 16-channel output tiles
    constexpr int NK = C == 16 ? 5 : 9;      // K steps of 32
    constexpr int PITCH = C * 2;             // bytes per pixel
    constexpr int PPP = PITCH / 16;          // 16-byte pieces per pixel
    constexpr int NPIECE = PAIR_NW * PPP;
    constexpr int NLD = (NPIECE + 255) / 256;
    __shared__ __attribute__((aligned(16))) uint8_t win[PAIR_NW * PITCH];
    __shared__ __attribute__((aligned(16))) uint8_t inter[PAIR_G1 * 16 * PITCH];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 15, kg = lane >> 4;
    const int img = blockIdx.x / a.tiles_per_img;
    const int tl = blockIdx.x - img * a.tiles_per_img;
    const int ty0 = (tl / a.tiles_x) * PAIR_T, tx0 = (tl % a.tiles_x) * PAIR_T;

    // ---- window: 16-byte pieces, zero outside the image (out-of-range buffer offsets)
    __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, a.in_bytes, 0x00020000);
    pu32x4 ra[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int e = tid + 256 * i;
        const int pix = e / PPP, c = e - pix * PPP;
        const int wy = pix / PAIR_WW, wx = pix - wy * PAIR_WW;
        const int iy = ty0 - 2 + wy, ix = tx0 - 2 + wx;
        const bool ok = e < NPIECE && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        const uint32_t off = ok ? ((uint32_t)((img * a.H + iy) * a.W + ix) * (uint32_t)a.in_cs + (uint32_t)a.in_coff) * 2u + (uint32_t)c * 16u : PAIR_OOB;
        ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rin, off, 0, 0);
    }
    // ---- conv1 weights (fragment order [k step][tile][lane][8]) and bias
    vec8 wa[NK][NT];
#pragma unroll
    for (int ks = 0; ks < NK; ++ks)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) wa[ks][nt] = *reinterpret_cast<const vec8*>(a.w1 + ((size_t)(ks * NT + nt) * 64 + lane) * 8);
    float4 bias[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bias[nt] = *reinterpret_cast<const float4*>(a.b1 + nt * 16 + kg * 4);
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int e = tid + 256 * i;
        const int pix = e / PPP, c = e - pix * PPP;
        if (e < NPIECE) *reinterpret_cast<pu32x4*>(win + pix * PITCH + pair_pos<C>(pix, c) * 16) = ra[i];
    }
    __syncthreads();

    // per-lane tap shift within a pixel row pitch `pw` (window 20, intermediate 18) for K step ks
    auto tap_of = [&](int ks) { return C == 32 ? ks : (2 * ks + (kg >> 1) > 8 ? 8 : 2 * ks + (kg >> 1)); };
    // B fragment of pixel p (linear index at pitch pw already shifted by the tap) from a 64- or 32-byte-pixel buffer
    auto frag = [&](const uint8_t* buf, int p) {
        return C == 32 ? *reinterpret_cast<const vec8*>(buf + p * PITCH + pair_pos<C>(p, kg) * 16)
                       : *reinterpret_cast<const vec8*>(buf + p * PITCH + (kg & 1) * 16);
    };

    // ---- conv1 on the 18 x 18 region: wave w takes pixel groups w, w + 4, ...
    constexpr int G1W = (PAIR_G1 + 3) / 4;   // 6
    pf32x4 acc1[G1W][NT];
    int wp[G1W];
#pragma unroll
    for (int g = 0; g < G1W; ++g) {
        int q = (wave + 4 * g) * 16 + lrow;
        q = q < PAIR_NI ? q : PAIR_NI - 1;
        const int qy = q / PAIR_IW, qx = q - qy * PAIR_IW;
        wp[g] = qy * PAIR_WW + qx;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc1[g][nt] = pf32x4{bias[nt].x, bias[nt].y, bias[nt].z, bias[nt].w};
    }
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
        const int t = tap_of(ks);
        const int sh = (t / 3) * PAIR_WW + (t % 3);
#pragma unroll
        for (int g = 0; g < G1W; ++g) {
            if (wave + 4 * g < PAIR_G1) {   // wave-uniform
                const vec8 xf = frag(win, wp[g] + sh);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc1[g][nt] = E::mfma(wa[ks][nt], xf, acc1[g][nt]);
            }
        }
    }
    // conv2's weights replace conv1's in the same registers while the intermediate is written
#pragma unroll
    for (int ks = 0; ks < NK; ++ks)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) wa[ks][nt] = *reinterpret_cast<const vec8*>(a.w2 + ((size_t)(ks * NT + nt) * 64 + lane) * 8);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bias[nt] = *reinterpret_cast<const float4*>(a.b2 + nt * 16 + kg * 4);
#pragma unroll
    for (int g = 0; g < G1W; ++g) {
        const int q = (wave + 4 * g) * 16 + lrow;
        if (wave + 4 * g < PAIR_G1) {
            const int qy = q / PAIR_IW, qx = q - qy * PAIR_IW;
            const int iy = ty0 - 1 + qy, ix = tx0 - 1 + qx;
            const bool inside = q < PAIR_NI && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                uint2 v;
                v.x = inside ? E::pack2(pair_silu(acc1[g][nt][0]), pair_silu(acc1[g][nt][1])) : 0u;
                v.y = inside ? E::pack2(pair_silu(acc1[g][nt][2]), pair_silu(acc1[g][nt][3])) : 0u;
                *reinterpret_cast<uint2*>(inter + q * PITCH + pair_pos<C>(q, nt * 2 + (kg >> 1)) * 16 + (kg & 1) * 8) = v;
            }
        }
    }
    __syncthreads();

    // ---- conv2 on the 16 x 16 tile: wave w takes rows 4w .. 4w + 3 (one 16-pixel group per row)
    pf32x4 acc2[4][NT];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc2[g][nt] = pf32x4{bias[nt].x, bias[nt].y, bias[nt].z, bias[nt].w};
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
        const int t = tap_of(ks);
        const int sh = (t / 3) * PAIR_IW + (t % 3);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const vec8 xf = frag(inter, (wave * 4 + g) * PAIR_IW + lrow + sh);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc2[g][nt] = E::mfma(wa[ks][nt], xf, acc2[g][nt]);
        }
    }
    // ---- epilogue: SiLU, shortcut from the window's centre, 8-byte stores (lane: 4 channels of one pixel)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int oy = wave * 4 + g, y = ty0 + oy, x = tx0 + lrow;
        const int wpix = (oy + 2) * PAIR_WW + lrow + 2;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            float v[4] = {pair_silu(acc2[g][nt][0]), pair_silu(acc2[g][nt][1]), pair_silu(acc2[g][nt][2]), pair_silu(acc2[g][nt][3])};
            if (a.has_res) {
                const uint2 r = *reinterpret_cast<const uint2*>(win + wpix * PITCH + pair_pos<C>(wpix, nt * 2 + (kg >> 1)) * 16 + (kg & 1) * 8);
                v[0] += E::lo(r.x); v[1] += E::hi(r.x); v[2] += E::lo(r.y); v[3] += E::hi(r.y);
            }
            uint2 o;
            o.x = E::pack2(v[0], v[1]);
            o.y = E::pack2(v[2], v[3]);
            if (y < a.H && x < a.W)
                *reinterpret_cast<uint2*>(a.out + ((size_t)(img * a.H + y) * a.W + x) * a.out_cs + a.out_coff + nt * 16 + kg * 4) = o;
        }
    }
}

__device__ __forceinline__ void pair_store(uint16_t* p, float v) { *p = Bf16::from_f32(v); }
__device__ __forceinline__ void pair_store(f16s* p, float v) { p->v = Fp16::from_f32(v); }

// fp32 [C][9][C] (cout, tap, cin) -> 16-bit MFMA A fragments [k step][tile][lane][8]
template <typename T, int C>
__global__ void pack_weights_pair_kernel(const float* __restrict__ src, T* __restrict__ dst) {
    constexpr int NT = C / 16, NK = C == 16 ? 5 : 9;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= NK * NT * 64 * 8) return;
    const int e = idx & 7, lane = (idx >> 3) & 63, f = idx >> 9;
    const int nt = f % NT, ks = f / NT;
    const int m = nt * 16 + (lane & 15), kg = lane >> 4;
    const int tap = C == 32 ? ks : 2 * ks + (kg >> 1);
    const int cin = C == 32 ? kg * 8 + e : (kg & 1) * 8 + e;
    const float v = tap < 9 ? src[((size_t)m * 9 + tap) * C + cin] : 0.0f;
    pair_store(dst + idx, v);
}

hipError_t launch_pack_weights_pair(const float* src, void* dst, int c, int prec, hipStream_t st) {
    const int total = (c == 16 ? 5 : 18) * 512;
    const int blocks = (total + 255) / 256;
    if (c == 16) {
        if (prec == PREC_FP16) hipLaunchKernelGGL((pack_weights_pair_kernel<f16s, 16>), dim3(blocks), dim3(256), 0, st, src, (f16s*)dst);
        else hipLaunchKernelGGL((pack_weights_pair_kernel<uint16_t, 16>), dim3(blocks), dim3(256), 0, st, src, (uint16_t*)dst);
    } else {
        if (prec == PREC_FP16) hipLaunchKernelGGL((pack_weights_pair_kernel<f16s, 32>), dim3(blocks), dim3(256), 0, st, src, (f16s*)dst);
        else hipLaunchKernelGGL((pack_weights_pair_kernel<uint16_t, 32>), dim3(blocks), dim3(256), 0, st, src, (uint16_t*)dst);
    }
    return hipGetLastError();
}

static bool pair_enabled() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("ADAS_NO_PAIR_FUSE");
        v = (e && e[0] == '1') ? 0 : 1;
    }
    return v == 1;
}

// conv A (x -> t) followed by conv B (t -> y, optional shortcut x): both 3x3 stride 1 pad 1 SiLU on 16 or 32 channels
bool pair_applicable(int prec, int kh, int kw, int stride, int pad, int act, int res_mode, const TView& x, const TView& t, int kh2, int kw2,
                     int stride2, int pad2, int act2, int res_mode2, const TView& y, const TView& res2) {
    if (!pair_enabled() || !prec_is16(prec)) return false;
    if (kh != 3 || kw != 3 || stride != 1 || pad != 1 || act != ACT_SILU || res_mode != RES_NONE) return false;
    if (kh2 != 3 || kw2 != 3 || stride2 != 1 || pad2 != 1 || act2 != ACT_SILU) return false;
    if (res_mode2 != RES_NONE && res_mode2 != RES_AFTER_ACT) return false;
    if ((x.c != 16 && x.c != 32) || t.c != x.c || y.c != x.c) return false;
    if (x.f32 || t.f32 || y.f32 || x.h != y.h || x.w != y.w || t.h != x.h || t.w != x.w) return false;
    if ((x.cs & 7) || (x.coff & 7) || (y.cs & 3) || (y.coff & 3)) return false;
    if (res_mode2 == RES_AFTER_ACT && (res2.p != x.p || res2.coff != x.coff || res2.cs != x.cs || res2.c != x.c)) return false;
    return true;
}

hipError_t launch_conv_pair(const TView& x, const TView& y, const void* w1, const float* b1, const void* w2, const float* b2, int n, bool has_res,
                            int prec, hipStream_t st) {
    if ((double)n * x.h * x.w * x.cs * 2.0 >= (double)PAIR_OOB) return hipErrorNotSupported;
    PairDev d;
    d.in = (const uint16_t*)x.p; d.out = (uint16_t*)y.p;
    d.w1 = (const uint16_t*)w1; d.w2 = (const uint16_t*)w2; d.b1 = b1; d.b2 = b2;
    d.in_bytes = (uint32_t)((size_t)n * x.h * x.w * x.cs * 2);
    d.in_cs = x.cs; d.in_coff = x.coff; d.out_cs = y.cs; d.out_coff = y.coff;
    d.H = x.h; d.W = x.w; d.has_res = has_res ? 1 : 0;
    d.tiles_x = (x.w + PAIR_T - 1) / PAIR_T;
    d.tiles_per_img = d.tiles_x * ((x.h + PAIR_T - 1) / PAIR_T);
    dim3 grid((unsigned)(n * d.tiles_per_img));
    if (x.c == 16) {
        if (prec == PREC_FP16) hipLaunchKernelGGL((conv_pair_kernel<Fp16, 16>), grid, dim3(256), 0, st, d);
        else hipLaunchKernelGGL((conv_pair_kernel<Bf16, 16>), grid, dim3(256), 0, st, d);
    } else {
        if (prec == PREC_FP16) hipLaunchKernelGGL((conv_pair_kernel<Fp16, 32>), grid, dim3(256), 0, st, d);
        else hipLaunchKernelGGL((conv_pair_kernel<Bf16, 32>), grid, dim3(256), 0, st, d);
    }
    return hipGetLastError();
}

}  // namespace adas
