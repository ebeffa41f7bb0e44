// conv_stem.hip -- the first layer of each network, fused with the layers around it:
//   engine-seam tensor (NCHW fp32, coreEngine.py:150-157) -> stride-2 conv (7x7 ResNet stem, backbone.py:50-52;
//   3x3 / 6x6 YOLO stems) + bias + ReLU/SiLU [-> 3x3 s2 p1 max-pool, backbone.py:53] -> NHWC bf16.
//
// The generic path spends three launches here (NCHW->NHWC8 conversion, an implicit GEMM whose K is padded
// from 147 to 392 because Cin = 3 is stored as 8, the pool) and moves the 16.4 MB/frame conv1 output through
// HBM twice.  This kernel reads the fp32 planes once, keeps a zero-padded (c0,c1,c2,0) bf16 window in LDS and
// feeds the MFMAs from it without im2col: with 4-channel pixels one 16x16x32 B fragment is 8 consecutive
// window pixels of one tap row (2 pixels = 16 B per lane), stride 2 makes the per-lane LDS address
// (wy*WW + 2*ox + 2*kg) * 8 B, always 16 B aligned and conflict free; a KH-row kernel is KH K-steps
// (K = 32*KH, 7x7: 224 vs 147 useful).  Weights sit in LDS in fragment order for the whole (persistent) workgroup;
// the next tile's window is fetched into registers under the current tile's MFMAs.  With POOL the conv tile (9 x 33 pixels incl. the pool halo) goes to LDS as bf16 and
// only the 4 x 16 pooled pixels are written to HBM.
// Roofline: UFLDv2 stem + pool per frame: 6.1 MB in + 4.1 MB out (HBM) vs 3.67 GFLOP padded MFMA work ->
// MFMA-bound; YOLO stems are HBM-bound (4.9 MB in, 3.3 MB out, 0.3 GFLOP).
#include "kernels.h"
#include <string.h>

namespace adas {

typedef __attribute__((ext_vector_type(8))) __bf16 sbf16x8;
typedef __attribute__((ext_vector_type(4))) float sf32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t su32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t su32x2;
typedef __attribute__((ext_vector_type(2))) float sf32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 sbf16x2;

__device__ __forceinline__ uint32_t s_pack2(float a, float b) {
    sbf16x2 r = __builtin_convertvector(sf32x2{a, b}, sbf16x2);
    return __builtin_bit_cast(uint32_t, r);
}
template <int ACT>
__device__ __forceinline__ float s_act(float v) {
    if (ACT == ACT_SILU) return v * __frcp_rn(1.0f + __expf(-v));
    if (ACT == ACT_RELU) return fmaxf(v, 0.0f);
    return v;
}
__device__ __forceinline__ float s_lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float s_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ uint32_t s_max2(uint32_t a, uint32_t b) {  // per-half bf16 max (exact: bf16 <-> f32 is lossless)
    return s_pack2(fmaxf(s_lo(a), s_lo(b)), fmaxf(s_hi(a), s_hi(b)));
}

struct StemDev {
    const float* in;        // [N][C][H][W] fp32
    const uint16_t* wfrag;  // [NT][KH][64 lanes][8] bf16, fragment order
    const float* bias;      // [>= NT*16]
    uint16_t* out;          // NHWC bf16 view
    int out_cs, out_coff, cout;
    int N, C, H, W;
    int Ho, Wo;             // conv output
    int Hp, Wp;             // pooled output (POOL)
    int pad;
    int tiles_x, tiles_y, ntiles;
};

constexpr int STEM_WW = 72;  // window row pitch in pixels (even: 16 B aligned fragment reads)

template <int KH, int NT, int ACT, bool POOL>
__global__ __launch_bounds__(256, 2) void conv_stem_kernel(StemDev a) {
    constexpr int CTH = POOL ? 9 : 8, CTW = POOL ? 33 : 32;  // conv tile (POOL: 4x16 pooled + halo)
    constexpr int NPIX = CTH * CTW;
    constexpr int NMT = (NPIX + 15) / 16, MT = (NMT + 3) / 4;
    constexpr int WW = STEM_WW, WH = 2 * (CTH - 1) + KH;
    constexpr int NQ = (WH * WW + 255) / 256;
    constexpr int CP = NT * 16 + 4;  // conv-tile pixel pitch in elements (pad: conflict-free 8 B writes, 8 B aligned)
    constexpr int WIN_E = WH * WW * 4, CT_E = POOL ? NPIX * CP : 0;
    // weights: fragment order, one contiguous 1 KB block per (channel tile, tap row) -> conflict-free ds_read_b128.
    // (Held in registers they cost 112 VGPRs at <7,4>, which put the kernel at the 256-VGPR limit with spills.)
    __shared__ __attribute__((aligned(16))) uint16_t wl[NT * KH * 512];
    // the window and the pooled-conv tile are never live at the same time: one region
    __shared__ __attribute__((aligned(16))) uint16_t wc[WIN_E > CT_E ? WIN_E : CT_E];
    uint16_t* win = wc;
    uint16_t* ctile = wc;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 15, kg = lane >> 4;
    for (int i = tid; i < NT * KH * 64; i += 256) *reinterpret_cast<su32x4*>(wl + i * 8) = *reinterpret_cast<const su32x4*>(a.wfrag + (size_t)i * 8);

    // per-lane window offsets of this wave's M tiles (tile-invariant)
    int boff[MT];
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        const int p = (wave * MT + j) * 16 + lrow;
        const int pc = p < NPIX ? p : NPIX - 1;
        const int cy = pc / CTW, cx = pc - cy * CTW;
        boff[j] = ((2 * cy) * WW + 2 * cx + 2 * kg) * 4;
    }
    const int per_img = a.tiles_x * a.tiles_y;
    const int plane = a.H * a.W;

    // window fetch of one tile into registers: fp32 planes, branch-free (buffer loads: offset 0x80000000 is out of range,
    // the hardware returns 0 = zero padding; a tile index past the end turns every lane out of range)
    float px[NQ][3];
    auto fetch = [&](int tile) {
        const bool live = tile < a.ntiles;
        const int tl = live ? tile : 0;
        const int img = tl / per_img;
        const int t2 = tl - img * per_img;
        const int ty = t2 / a.tiles_x, tx = t2 - ty * a.tiles_x;
        const int cy0 = POOL ? 2 * (ty * 4) - 1 : ty * CTH;
        const int cx0 = POOL ? 2 * (tx * 16) - 1 : tx * CTW;
        const int iy0 = 2 * cy0 - a.pad, ix0 = 2 * cx0 - a.pad;
        const float* in_img = a.in + (size_t)img * a.C * plane;
        __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)in_img, 0, a.C * plane * 4, 0x00020000);
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int q = tid + 256 * i;
            const int wy = q / WW, wx = q - wy * WW;
            const int iy = iy0 + wy, ix = ix0 + wx;
            const bool ok = live && q < WH * WW && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            const uint32_t off = (uint32_t)((iy * a.W + ix) * 4);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const uint32_t oc = (ok && c < a.C) ? off + (uint32_t)(c * plane * 4) : 0x80000000u;
                px[i][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, oc, 0, 0));
            }
        }
    };

    int tile = blockIdx.x;
    if (tile >= a.ntiles) return;
    fetch(tile);
    for (;;) {
        const int img = tile / per_img;
        const int t2 = tile - img * per_img;
        const int ty = t2 / a.tiles_x, tx = t2 - ty * a.tiles_x;
        const int cy0 = POOL ? 2 * (ty * 4) - 1 : ty * CTH;
        const int cx0 = POOL ? 2 * (tx * 16) - 1 : tx * CTW;

        __syncthreads();  // previous tile's readers of the window / conv tile are done (first trip: the weights are in LDS)
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int q = tid + 256 * i;
            if (q < WH * WW) {
                su32x2 v;
                v.x = s_pack2(px[i][0], px[i][1]);
                v.y = s_pack2(px[i][2], 0.f);
                *reinterpret_cast<su32x2*>(win + q * 4) = v;
            }
        }
        __syncthreads();
        const int next = tile + gridDim.x;
        fetch(next);  // in flight under this tile's MFMAs and pooling

        // ---- MFMA: KH K-steps.  No guard on M tiles past the end (the last wave's surplus tile reads a clamped, valid window
        // address and is dropped at store time): a branch per tile stops hipcc from overlapping LDS reads with MFMAs.
        sf32x4 acc[MT][NT];
#pragma unroll
        for (int j = 0; j < MT; ++j)
#pragma unroll
            for (int i = 0; i < NT; ++i) acc[j][i] = sf32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < KH; ++r) {
            sbf16x8 wf[NT], xf[MT];
#pragma unroll
            for (int i = 0; i < NT; ++i) wf[i] = *reinterpret_cast<const sbf16x8*>(wl + ((i * KH + r) * 64 + lane) * 8);
#pragma unroll
            for (int j = 0; j < MT; ++j) xf[j] = *reinterpret_cast<const sbf16x8*>(win + boff[j] + r * WW * 4);
#pragma unroll
            for (int j = 0; j < MT; ++j)
#pragma unroll
                for (int i = 0; i < NT; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[j][i], 0, 0, 0);
        }

        // ---- epilogue: lane holds channels i*16 + kg*4 .. +3 of conv pixel (pcy, pcx)
        float4 bias4[NT];
#pragma unroll
        for (int i = 0; i < NT; ++i) bias4[i] = *reinterpret_cast<const float4*>(a.bias + i * 16 + kg * 4);
        int pcy[MT], pcx[MT];
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            const int p = (wave * MT + j) * 16 + lrow;
            const int pc = p < NPIX ? p : NPIX - 1;
            pcy[j] = pc / CTW;
            pcx[j] = pc - pcy[j] * CTW;
        }
        if (!POOL) {
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                const int p = (wave * MT + j) * 16 + lrow;
                const int oy = cy0 + pcy[j], ox = cx0 + pcx[j];
                if (p >= NPIX || oy >= a.Ho || ox >= a.Wo) continue;
                uint16_t* op = a.out + ((size_t)(img * a.Ho + oy) * a.Wo + ox) * a.out_cs + a.out_coff + kg * 4;
#pragma unroll
                for (int i = 0; i < NT; ++i) {
                    if (i * 16 + kg * 4 >= a.cout) continue;
                    su32x2 q;
                    q.x = s_pack2(s_act<ACT>(acc[j][i][0] + bias4[i].x), s_act<ACT>(acc[j][i][1] + bias4[i].y));
                    q.y = s_pack2(s_act<ACT>(acc[j][i][2] + bias4[i].z), s_act<ACT>(acc[j][i][3] + bias4[i].w));
                    *reinterpret_cast<su32x2*>(op + i * 16) = q;
                }
            }
        } else {
            __syncthreads();  // every wave is done reading the window: the conv tile may overwrite it
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                const int p = (wave * MT + j) * 16 + lrow;
                if (p >= NPIX) continue;
                const int gy = cy0 + pcy[j], gx = cx0 + pcx[j];
                const bool valid = (unsigned)gy < (unsigned)a.Ho && (unsigned)gx < (unsigned)a.Wo;  // else: pool padding = -inf
#pragma unroll
                for (int i = 0; i < NT; ++i) {
                    su32x2 q;
                    q.x = s_pack2(s_act<ACT>(acc[j][i][0] + bias4[i].x), s_act<ACT>(acc[j][i][1] + bias4[i].y));
                    q.y = s_pack2(s_act<ACT>(acc[j][i][2] + bias4[i].z), s_act<ACT>(acc[j][i][3] + bias4[i].w));
                    if (!valid) q.x = q.y = 0xff80ff80u;
                    *reinterpret_cast<su32x2*>(ctile + p * CP + i * 16 + kg * 4) = q;
                }
            }
            __syncthreads();
            constexpr int CG = NT * 2;  // 8-channel groups per pixel
            for (int it = tid; it < 64 * CG; it += 256) {
                const int pp = it / CG, cg = it - pp * CG;
                const int py = pp >> 4, pxx = pp & 15;
                const int gpy = ty * 4 + py, gpx = tx * 16 + pxx;
                if (gpy >= a.Hp || gpx >= a.Wp || cg * 8 >= a.cout) continue;
                su32x2 m0{0xff80ff80u, 0xff80ff80u}, m1 = m0;
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const uint16_t* cp = ctile + ((2 * py + dy) * CTW + 2 * pxx + dx) * CP + cg * 8;
                        const su32x2 v0 = *reinterpret_cast<const su32x2*>(cp);
                        const su32x2 v1 = *reinterpret_cast<const su32x2*>(cp + 4);
                        m0.x = s_max2(m0.x, v0.x); m0.y = s_max2(m0.y, v0.y);
                        m1.x = s_max2(m1.x, v1.x); m1.y = s_max2(m1.y, v1.y);
                    }
                uint16_t* op = a.out + ((size_t)(img * a.Hp + gpy) * a.Wp + gpx) * a.out_cs + a.out_coff + cg * 8;
                *reinterpret_cast<su32x4*>(op) = su32x4{m0.x, m0.y, m1.x, m1.y};
            }
        }
        if (next >= a.ntiles) break;
        tile = next;
    }
}

// -------------------------------------------------------------------------------------
bool stem_applicable(int prec, int in_c_true, int kh, int kw, int stride, int pad, int act, int res_mode, const TView& out, bool pool,
                     const TView& pool_out) {
    if (prec != PREC_BF16 || in_c_true > 3 || stride != 2 || res_mode != RES_NONE) return false;
    if (!(kh == 3 || kh == 6 || kh == 7) || kw != kh) return false;  // kw <= 8 pixel slots per K step
    if (pad > kh / 2) return false;
    if (out.f32 || !(out.c == 16 || out.c == 32 || out.c == 64)) return false;
    const TView& o = pool ? pool_out : out;
    if ((o.cs & 7) || (o.coff & 7) || o.f32) return false;
    if (pool && (kh != 7 || out.c != 64 || act != ACT_RELU)) return false;  // the ResNet stem is the only pooled instance
    if (!pool && !(act == ACT_SILU || act == ACT_RELU)) return false;
    return true;
}

size_t stem_weight_bytes(int kh, int cout) { return (size_t)((cout + 15) / 16) * kh * 64 * 8 * 2; }

// host-side packing: w = [cout][kh][kw][cs] fp32 (OHWI, channel pitch cs) -> fragment order bf16 bits
static uint16_t h_f2bf(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
void stem_pack_weights(const float* w, int cout, int kh, int kw, int cs, int c_true, uint16_t* dst) {
    const int NT = (cout + 15) / 16;
    for (int nt = 0; nt < NT; ++nt)
        for (int r = 0; r < kh; ++r)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 8; ++e) {
                    const int co = nt * 16 + (lane & 15), kg = lane >> 4;
                    const int s = 2 * kg + e / 4, ch = e & 3;
                    float v = 0.f;
                    if (co < cout && s < kw && ch < c_true) v = w[(((size_t)co * kh + r) * kw + s) * cs + ch];
                    dst[((size_t)(nt * kh + r) * 64 + lane) * 8 + e] = h_f2bf(v);
                }
}

template <int KH, int NT, bool POOL>
static hipError_t stem_launch_act(const StemDev& d, int act, hipStream_t st) {
    const int grid = d.ntiles < 1024 ? d.ntiles : 1024;
    if (act == ACT_RELU) hipLaunchKernelGGL((conv_stem_kernel<KH, NT, ACT_RELU, POOL>), dim3(grid), dim3(256), 0, st, d);
    else hipLaunchKernelGGL((conv_stem_kernel<KH, NT, ACT_SILU, POOL>), dim3(grid), dim3(256), 0, st, d);
    return hipGetLastError();
}

hipError_t launch_conv_stem(const float* nchw, int n, int c_true, int H, int W, int kh, int pad, int act, const void* wfrag,
                            const float* bias, const TView& conv_out, bool pool, const TView& pool_out, hipStream_t st) {
    StemDev d;
    const TView& o = pool ? pool_out : conv_out;
    d.in = nchw; d.wfrag = (const uint16_t*)wfrag; d.bias = bias;
    d.out = (uint16_t*)o.p; d.out_cs = o.cs; d.out_coff = o.coff; d.cout = conv_out.c;
    d.N = n; d.C = c_true; d.H = H; d.W = W; d.Ho = conv_out.h; d.Wo = conv_out.w;
    d.Hp = pool ? pool_out.h : 0; d.Wp = pool ? pool_out.w : 0;
    d.pad = pad;
    if (pool) {
        d.tiles_x = (d.Wp + 15) / 16; d.tiles_y = (d.Hp + 3) / 4;
    } else {
        d.tiles_x = (d.Wo + 31) / 32; d.tiles_y = (d.Ho + 7) / 8;
    }
    d.ntiles = n * d.tiles_x * d.tiles_y;
    if ((size_t)c_true * H * W * 4 >= (1ull << 31)) return hipErrorInvalidValue;
    const int nt = (conv_out.c + 15) / 16;
    if (pool) return stem_launch_act<7, 4, true>(d, act, st);
#define STEM_CASE(KH_, NT_) \
    if (kh == KH_ && nt == NT_) return stem_launch_act<KH_, NT_, false>(d, act, st);
    STEM_CASE(3, 1) STEM_CASE(3, 2) STEM_CASE(3, 4)
    STEM_CASE(6, 1) STEM_CASE(6, 2) STEM_CASE(6, 4)
    STEM_CASE(7, 4)
#undef STEM_CASE
    return hipErrorInvalidValue;
}

}  // namespace adas
