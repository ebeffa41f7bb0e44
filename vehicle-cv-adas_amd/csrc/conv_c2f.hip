// conv_c2f.hip -- one whole C2f block (ultralytics C2f(c1 = 32, c2 = 32, n = 1, shortcut = True): YOLOv8n / YOLOv10n `model.2`, the
// block on the networks' largest map, 160 x 160) in ONE launch:
//     (y0, y1) = split(SiLU(cv1 x));  y2 = y1 + SiLU(B(SiLU(A y1)));  out = SiLU(cv2 cat(y0, y1, y2))
// with cv1 / cv2 1x1 (32 -> 32, 48 -> 32) and A / B the Bottleneck's 3x3 convs on 16 channels.  As three launches (1x1, the fused 3x3
// pair of conv_pair.hip, 1x1) the block moves 576 MB per 64 frames for 24 GFLOP -- x in, (y0, y1) out, y1 in, y2 out, the 48-channel
// concat in, out out -- 0.18 ms at ~3 TB/s; fused, x (with a 2-pixel halo) is read once and `out` written once (~230 MB), the
// concat buffer never exists.
// A workgroup owns a 16 x 16 output tile.  Stages, all operands in LDS between them:
//   1. x window 20 x 20 x 32 ch (zero outside the image)                                    -> xwin
//   2. cv1 on all 400 window pixels (the Bottleneck needs y1 on the 20 x 20 window; y0 only on the tile).  y1 is ZEROED outside the
//      image: conv A's zero padding applies to cv1's OUTPUT domain                             -> y1win (16 ch), y0t (tile, 16 ch)
//   3. conv A on the 18 x 18 region conv B needs, SiLU, zeroed outside the image            -> inter
//   4. conv B on the tile, SiLU, + y1                                                       -> y2t
//   5. cv2 over K = 48 = [y0 | y1 | y2] (+ 16 zero channels), SiLU                          -> HBM
// MFMA mapping as in conv_pair.hip: weights are the A operand, 16 pixels the B operand, a lane ends up with 4 consecutive output
// channels of one pixel; the 16-channel 3x3 convs take two taps per 32-deep K step.  All four weight sets are register fragments.
#include "kernels.h"
#include "elem16.h"
#include <stdlib.h>

namespace adas {

typedef __attribute__((ext_vector_type(4))) float cf32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t cu32x4;

struct C2fDev {
    const uint16_t* in;
    uint16_t* out;
    const uint16_t *w_cv1, *w_a, *w_b, *w_cv2;   // fragment order (pack kernels below / conv_pair.hip)
    const float *b_cv1, *b_a, *b_b, *b_cv2;
    uint32_t in_bytes;
    int in_cs, in_coff, out_cs, out_coff;
    int H, W;
    int tiles_x, tiles_per_img;
};

constexpr int CF_T = 16, CF_IW = CF_T + 2, CF_WW = CF_T + 4;
constexpr int CF_NI = CF_IW * CF_IW, CF_G1 = (CF_NI + 15) / 16;      // 324 intermediate pixels, 21 groups
constexpr int CF_NW = CF_WW * CF_WW, CF_GW = CF_NW / 16;             // 400 window pixels, 25 groups
constexpr uint32_t CF_OOB = 0x80000000u;

__device__ __forceinline__ float cf_silu(float v) { return v * fast_rcp(1.0f + __expf(-v)); }
// 64-byte pixels (32 channels): conv_halo's conflict-free chunk swizzle
__device__ __forceinline__ int cf_pos32(int p, int c) { return c ^ (((p >> 2) & 1) << 1); }

template <typename E>
__global__ __launch_bounds__(256, 3) void conv_c2f16_kernel(C2fDev a) {
    E::enter();
    typedef typename E::vec8 vec8;
    // region A: the x window (25.6 KB), later conv A's output (21 groups x 16 px x 32 B = 10.5 KB) and y2 of the tile (8 KB)
    __shared__ __attribute__((aligned(16))) uint8_t regA[CF_NW * 64];
    __shared__ __attribute__((aligned(16))) uint8_t y1win[CF_NW * 32];   // y1 on the window, 16 channels
    __shared__ __attribute__((aligned(16))) uint8_t y0t[256 * 32];       // y0 on the tile, 16 channels
    uint8_t* const xwin = regA;
    uint8_t* const inter = regA;
    uint8_t* const y2t = regA + CF_G1 * 16 * 32;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 15, kg = lane >> 4;
    const int img = blockIdx.x / a.tiles_per_img;
    const int tl = blockIdx.x - img * a.tiles_per_img;
    const int ty0 = (tl / a.tiles_x) * CF_T, tx0 = (tl % a.tiles_x) * CF_T;

    // ---- 1. x window: 1600 16-byte pieces, zero outside the image (out-of-range buffer offsets)
    __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, a.in_bytes, 0x00020000);
    constexpr int NLD = (CF_NW * 4 + 255) / 256;   // 7
    cu32x4 ra[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int e = tid + 256 * i;
        const int pix = e >> 2, c = e & 3;
        const int wy = pix / CF_WW, wx = pix - wy * CF_WW;
        const int iy = ty0 - 2 + wy, ix = tx0 - 2 + wx;
        const bool ok = e < CF_NW * 4 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        const uint32_t off = ok ? ((uint32_t)((img * a.H + iy) * a.W + ix) * (uint32_t)a.in_cs + (uint32_t)a.in_coff) * 2u + (uint32_t)c * 16u : CF_OOB;
        ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rin, off, 0, 0);
    }
    // weights of all four convs as register fragments (L2-hot, fragment-ordered arrays)
    vec8 w1[2], wA[5], wB[5], w2[2][2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) w1[nt] = *reinterpret_cast<const vec8*>(a.w_cv1 + ((size_t)nt * 64 + lane) * 8);
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) {
        wA[ks] = *reinterpret_cast<const vec8*>(a.w_a + ((size_t)ks * 64 + lane) * 8);
        wB[ks] = *reinterpret_cast<const vec8*>(a.w_b + ((size_t)ks * 64 + lane) * 8);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) w2[ks][nt] = *reinterpret_cast<const vec8*>(a.w_cv2 + ((size_t)(ks * 2 + nt) * 64 + lane) * 8);
    const float4 b1lo = *reinterpret_cast<const float4*>(a.b_cv1 + kg * 4), b1hi = *reinterpret_cast<const float4*>(a.b_cv1 + 16 + kg * 4);
    const float4 bA = *reinterpret_cast<const float4*>(a.b_a + kg * 4), bB = *reinterpret_cast<const float4*>(a.b_b + kg * 4);
    const float4 b2lo = *reinterpret_cast<const float4*>(a.b_cv2 + kg * 4), b2hi = *reinterpret_cast<const float4*>(a.b_cv2 + 16 + kg * 4);
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int e = tid + 256 * i;
        const int pix = e >> 2, c = e & 3;
        if (e < CF_NW * 4) *reinterpret_cast<cu32x4*>(xwin + pix * 64 + cf_pos32(pix, c) * 16) = ra[i];
    }
    __syncthreads();

    // ---- 2. cv1 (1x1, 32 -> 32, SiLU) on the 25 window pixel groups: wave w takes groups w, w + 4, ...
    //         channels 0..15 = y0 (kept for tile pixels), 16..31 = y1 (kept for the whole window, zero outside the image)
    cf32x4 acc1[7][2];
#pragma unroll
    for (int g = 0; g < 7; ++g) {
        if (wave + 4 * g < CF_GW) {   // wave-uniform
            const int p = (wave + 4 * g) * 16 + lrow;
            const vec8 xf = *reinterpret_cast<const vec8*>(xwin + p * 64 + cf_pos32(p, kg) * 16);
            acc1[g][0] = E::mfma(w1[0], xf, cf32x4{b1lo.x, b1lo.y, b1lo.z, b1lo.w});
            acc1[g][1] = E::mfma(w1[1], xf, cf32x4{b1hi.x, b1hi.y, b1hi.z, b1hi.w});
        }
    }
    // (y1win / y0t are separate arrays: no barrier needed before writing them; the one behind the writes also orders every wave's
    //  x-fragment reads before stage 3 overwrites region A)
#pragma unroll
    for (int g = 0; g < 7; ++g) {
        if (wave + 4 * g < CF_GW) {
            const int p = (wave + 4 * g) * 16 + lrow;
            const int wy = p / CF_WW, wx = p - wy * CF_WW;
            const int iy = ty0 - 2 + wy, ix = tx0 - 2 + wx;
            const bool inside = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            uint2 v;
            v.x = inside ? E::pack2(cf_silu(acc1[g][1][0]), cf_silu(acc1[g][1][1])) : 0u;
            v.y = inside ? E::pack2(cf_silu(acc1[g][1][2]), cf_silu(acc1[g][1][3])) : 0u;
            *reinterpret_cast<uint2*>(y1win + p * 32 + kg * 8) = v;           // channel 4 kg .. 4 kg + 3 of y1
            const int oy = wy - 2, ox = wx - 2;
            if ((unsigned)oy < (unsigned)CF_T && (unsigned)ox < (unsigned)CF_T) {
                uint2 u;
                u.x = E::pack2(cf_silu(acc1[g][0][0]), cf_silu(acc1[g][0][1]));
                u.y = E::pack2(cf_silu(acc1[g][0][2]), cf_silu(acc1[g][0][3]));
                *reinterpret_cast<uint2*>(y0t + (oy * CF_T + ox) * 32 + kg * 8) = u;
            }
        }
    }
    __syncthreads();

    // 16-channel 3x3 convs: one K step per PAIR of taps (lanes kg 0-1 take tap 2 ks, kg 2-3 tap 2 ks + 1; the tenth half-step
    // multiplies zero weights)
    auto tap_of = [&](int ks) { return 2 * ks + (kg >> 1) > 8 ? 8 : 2 * ks + (kg >> 1); };
    auto frag16 = [&](const uint8_t* buf, int p) { return *reinterpret_cast<const vec8*>(buf + p * 32 + (kg & 1) * 16); };

    // ---- 3. conv A on the 18 x 18 region (21 groups): wave w takes groups w, w + 4, ...
    cf32x4 accA[6];
    int wp[6];
#pragma unroll
    for (int g = 0; g < 6; ++g) {
        int q = (wave + 4 * g) * 16 + lrow;
        q = q < CF_NI ? q : CF_NI - 1;
        const int qy = q / CF_IW, qx = q - qy * CF_IW;
        wp[g] = qy * CF_WW + qx;
        accA[g] = cf32x4{bA.x, bA.y, bA.z, bA.w};
    }
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) {
        const int t = tap_of(ks);
        const int sh = (t / 3) * CF_WW + (t % 3);
#pragma unroll
        for (int g = 0; g < 6; ++g)
            if (wave + 4 * g < CF_G1) accA[g] = E::mfma(wA[ks], frag16(y1win, wp[g] + sh), accA[g]);
    }
#pragma unroll
    for (int g = 0; g < 6; ++g) {
        const int q = (wave + 4 * g) * 16 + lrow;
        if (wave + 4 * g < CF_G1) {
            const int qy = q / CF_IW, qx = q - qy * CF_IW;
            const int iy = ty0 - 1 + qy, ix = tx0 - 1 + qx;
            const bool inside = q < CF_NI && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            uint2 v;
            v.x = inside ? E::pack2(cf_silu(accA[g][0]), cf_silu(accA[g][1])) : 0u;
            v.y = inside ? E::pack2(cf_silu(accA[g][2]), cf_silu(accA[g][3])) : 0u;
            *reinterpret_cast<uint2*>(inter + q * 32 + kg * 8) = v;
        }
    }
    __syncthreads();

    // ---- 4. conv B on the 16 x 16 tile: wave w takes rows 4 w .. 4 w + 3; SiLU; + y1 (shortcut)  -> y2t
    cf32x4 accB[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) accB[g] = cf32x4{bB.x, bB.y, bB.z, bB.w};
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) {
        const int t = tap_of(ks);
        const int sh = (t / 3) * CF_IW + (t % 3);
#pragma unroll
        for (int g = 0; g < 4; ++g) accB[g] = E::mfma(wB[ks], frag16(inter, (wave * 4 + g) * CF_IW + lrow + sh), accB[g]);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int oy = wave * 4 + g;
        const int wpix = (oy + 2) * CF_WW + lrow + 2;
        const uint2 r = *reinterpret_cast<const uint2*>(y1win + wpix * 32 + kg * 8);
        const float v0 = cf_silu(accB[g][0]) + E::lo(r.x), v1 = cf_silu(accB[g][1]) + E::hi(r.x);
        const float v2 = cf_silu(accB[g][2]) + E::lo(r.y), v3 = cf_silu(accB[g][3]) + E::hi(r.y);
        uint2 o;
        o.x = E::pack2(v0, v1);
        o.y = E::pack2(v2, v3);
        *reinterpret_cast<uint2*>(y2t + (oy * CF_T + lrow) * 32 + kg * 8) = o;
    }
    __syncthreads();

    // ---- 5. cv2 (1x1, K = 48 = [y0 | y1 | y2], padded to 64): two K steps; SiLU; 8-byte stores (lane: 4 channels of one pixel)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int oy = wave * 4 + g, tp = oy * CF_T + lrow;
        const int wpix = (oy + 2) * CF_WW + lrow + 2;
        // K step 0: channels 0..15 = y0 (kg 0, 1), 16..31 = y1 (kg 2, 3); K step 1: 32..47 = y2 (kg 0, 1), 48..63 = zero weights
        const vec8 f0 = kg < 2 ? *reinterpret_cast<const vec8*>(y0t + tp * 32 + kg * 16) : *reinterpret_cast<const vec8*>(y1win + wpix * 32 + (kg - 2) * 16);
        const vec8 f1 = *reinterpret_cast<const vec8*>(y2t + tp * 32 + (kg & 1) * 16);
        cf32x4 lo = E::mfma(w2[0][0], f0, cf32x4{b2lo.x, b2lo.y, b2lo.z, b2lo.w});
        cf32x4 hi = E::mfma(w2[0][1], f0, cf32x4{b2hi.x, b2hi.y, b2hi.z, b2hi.w});
        lo = E::mfma(w2[1][0], f1, lo);
        hi = E::mfma(w2[1][1], f1, hi);
        const int y = ty0 + oy, x = tx0 + lrow;
        if (y < a.H && x < a.W) {
            uint16_t* op = a.out + ((size_t)(img * a.H + y) * a.W + x) * a.out_cs + a.out_coff + kg * 4;
            uint2 o;
            o.x = E::pack2(cf_silu(lo[0]), cf_silu(lo[1]));
            o.y = E::pack2(cf_silu(lo[2]), cf_silu(lo[3]));
            *reinterpret_cast<uint2*>(op) = o;
            o.x = E::pack2(cf_silu(hi[0]), cf_silu(hi[1]));
            o.y = E::pack2(cf_silu(hi[2]), cf_silu(hi[3]));
            *reinterpret_cast<uint2*>(op + 16) = o;
        }
    }
}

__device__ __forceinline__ void cf_store(uint16_t* p, float v) { *p = Bf16::from_f32(v); }
__device__ __forceinline__ void cf_store(f16s* p, float v) { p->v = Fp16::from_f32(v); }

// fp32 [cout][cin] (1x1 conv, cout a multiple of 16, cin <= 32 * nks) -> MFMA A fragments [k step][cout tile][lane][8], K = 32 ks + 8 kg + e,
// zero beyond cin
template <typename T>
__global__ void pack_weights_c2f_pw_kernel(const float* __restrict__ src, T* __restrict__ dst, int cout, int cin, int nks) {
    const int nt_n = cout / 16;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nks * nt_n * 512) return;
    const int e = idx & 7, lane = (idx >> 3) & 63, f = idx >> 9;
    const int nt = f % nt_n, ks = f / nt_n;
    const int m = nt * 16 + (lane & 15), k = ks * 32 + (lane >> 4) * 8 + e;
    cf_store(dst + idx, k < cin ? src[(size_t)m * cin + k] : 0.0f);
}
hipError_t launch_pack_weights_c2f_pw(const float* src, void* dst, int cout, int cin, int prec, hipStream_t st) {
    const int nks = (cin + 31) / 32, total = nks * (cout / 16) * 512;
    if (prec == PREC_FP16) hipLaunchKernelGGL(pack_weights_c2f_pw_kernel<f16s>, dim3((total + 255) / 256), dim3(256), 0, st, src, (f16s*)dst, cout, cin, nks);
    else hipLaunchKernelGGL(pack_weights_c2f_pw_kernel<uint16_t>, dim3((total + 255) / 256), dim3(256), 0, st, src, (uint16_t*)dst, cout, cin, nks);
    return hipGetLastError();
}
size_t c2f_pw_weight_bytes(int cout, int cin) { return (size_t)((cin + 31) / 32) * (cout / 16) * 512 * 2; }

static bool c2f_enabled() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("ADAS_NO_C2F_FUSE");
        v = (e && e[0] == '1') ? 0 : 1;
    }
    return v == 1;
}

// cv1: x (32 ch) -> cat[0:32]; pair A / B on cat[16:32] -> cat[32:48] with shortcut; cv2: cat[0:48] -> out (32 ch)
bool c2f16_applicable(int prec, const TView& x, const TView& cat01, const TView& y1, const TView& y2, const TView& cat, const TView& out) {
    if (!c2f_enabled() || !prec_is16(prec)) return false;
    if (x.c != 32 || cat01.c != 32 || y1.c != 16 || y2.c != 16 || cat.c != 48 || out.c != 32) return false;
    if (x.f32 || cat.f32 || out.f32) return false;
    if (cat01.p != cat.p || y1.p != cat.p || y2.p != cat.p || cat01.coff != cat.coff || y1.coff != cat.coff + 16 || y2.coff != cat.coff + 32) return false;
    if (x.h != out.h || x.w != out.w || cat.h != x.h || cat.w != x.w) return false;
    if ((x.cs & 7) || (x.coff & 7) || (out.cs & 3) || (out.coff & 3)) return false;
    return true;
}

hipError_t launch_conv_c2f16(const TView& x, const TView& out, const void* w_cv1, const float* b_cv1, const void* w_a, const float* b_a, const void* w_b,
                             const float* b_b, const void* w_cv2, const float* b_cv2, int n, int prec, hipStream_t st) {
    if ((double)n * x.h * x.w * x.cs * 2.0 >= (double)CF_OOB) return hipErrorNotSupported;
    C2fDev d;
    d.in = (const uint16_t*)x.p; d.out = (uint16_t*)out.p;
    d.w_cv1 = (const uint16_t*)w_cv1; d.w_a = (const uint16_t*)w_a; d.w_b = (const uint16_t*)w_b; d.w_cv2 = (const uint16_t*)w_cv2;
    d.b_cv1 = b_cv1; d.b_a = b_a; d.b_b = b_b; d.b_cv2 = b_cv2;
    d.in_bytes = (uint32_t)((size_t)n * x.h * x.w * x.cs * 2);
    d.in_cs = x.cs; d.in_coff = x.coff; d.out_cs = out.cs; d.out_coff = out.coff;
    d.H = x.h; d.W = x.w;
    d.tiles_x = (x.w + CF_T - 1) / CF_T;
    d.tiles_per_img = d.tiles_x * ((x.h + CF_T - 1) / CF_T);
    const dim3 grid((unsigned)(n * d.tiles_per_img));
    if (prec == PREC_FP16) hipLaunchKernelGGL(conv_c2f16_kernel<Fp16>, grid, dim3(256), 0, st, d);
    else hipLaunchKernelGGL(conv_c2f16_kernel<Bf16>, grid, dim3(256), 0, st, d);
    return hipGetLastError();
}

}  // namespace adas
