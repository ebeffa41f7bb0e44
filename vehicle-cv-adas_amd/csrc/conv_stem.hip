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
#include "elem16.h"
#include <string.h>

namespace adas {

typedef __attribute__((ext_vector_type(4))) float sf32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t su32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t su32x2;

template <int ACT>
__device__ __forceinline__ float s_act(float v) {
    if (ACT == ACT_SILU) return v * fast_rcp(1.0f + __expf(-v));
    if (ACT == ACT_RELU) return fmaxf(v, 0.0f);
    if (ACT == ACT_LEAKY) return fmaxf(v, 0.1f * v);
    return v;
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
    const uint16_t* wfrag2;  // CONV2: [2][5][64 lanes][8] bf16 (3x3 s2 p1, 16 -> 32 channels), K-step = two taps x 16 channels
    const float* bias2;
};

#ifdef ADAS_STEM_PROF   // scratch instrumentation (tools/stem_prof.py): shader cycles of wave 0 per tile phase, summed over workgroups
__device__ unsigned long long g_stem_prof[16];
#define STP(i)                                      \
    if (tid == 0) {                                 \
        const unsigned long long t__ = clock64();   \
        pacc__[i] += t__ - tprev__;                 \
        tprev__ = t__;                              \
    }
#else
#define STP(i)
#endif

constexpr int STEM_WW = 72;  // window row pitch in pixels (even: 16 B aligned fragment reads)

// CONV2: the YOLO stems are followed by a 3x3 s2 p1 conv on their 16 channels (model.1); the 17 x 33 stem pixels an
// 8 x 16 tile of that conv needs are kept in LDS (geometry of the pooled case with 8 rows) and the second conv runs from
// there -- the 16-channel stem output (3.3 MB per 640^2 frame, written and read back) never reaches HBM.  Hp/Wp are then
// the second conv's output extent and `out` its view.
// PACKED: the input is the (c0, c1, c2, 0) bf16 NHWC tensor adas_preprocess_*_packed writes (8 B per pixel, one load per window
// pixel) instead of the fp32 NCHW seam tensor (three loads + a conversion); same values either way.
template <typename E, int KH, int NT, int ACT, bool POOL, bool CONV2 = false, bool PACKED = false>
__global__ __launch_bounds__(256, CONV2 ? 3 : 2) void conv_stem_kernel(StemDev a) {
    E::enter();
    typedef typename E::vec8 svec8;
    constexpr bool TILE2 = POOL || CONV2;
    constexpr int CTH = CONV2 ? 17 : (POOL ? 9 : 8), CTW = TILE2 ? 33 : 32;  // conv tile (POOL: 4x16 pooled + halo)
    constexpr int NPIX = CTH * CTW;
    constexpr int NMT = (NPIX + 15) / 16, MT = (NMT + 3) / 4;
    constexpr int WW = STEM_WW, WH = 2 * (CTH - 1) + KH;
    constexpr int NQ = (WH * WW + 255) / 256;
    constexpr int CP = CONV2 ? 24 : NT * 16 + 4;  // conv-tile pixel pitch in elements (pad: conflict-free 8 B writes, 8 B aligned;
                                                  // CONV2: 48 B so that its 16-byte fragment reads stay aligned)
    constexpr int WIN_E = WH * WW * 4, CT_E = TILE2 ? NPIX * CP : 0;
    __shared__ __attribute__((aligned(16))) uint16_t w2l[CONV2 ? 2 * 5 * 512 : 8];
    // weights: fragment order, one contiguous 1 KB block per (channel tile, tap row) -> conflict-free ds_read_b128.
    // (Held in registers they cost 112 VGPRs at <7,4>, which put the kernel at the 256-VGPR limit with spills.)
    __shared__ __attribute__((aligned(16))) uint16_t wl[NT * KH * 512];
    // the window and the pooled-conv tile are never live at the same time: one region
    __shared__ __attribute__((aligned(16))) uint16_t wc[WIN_E > CT_E ? WIN_E : CT_E];
    uint16_t* win = wc;
    uint16_t* ctile = wc;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 15, kg = lane >> 4;
    stage_lds16<256, (NT * KH * 64 + 255) / 256 < 8 ? (NT * KH * 64 + 255) / 256 : 8>(wl, a.wfrag, NT * KH * 64, tid);
    if (CONV2)
        for (int i = tid; i < 2 * 5 * 64; i += 256) *reinterpret_cast<su32x4*>(w2l + i * 8) = *reinterpret_cast<const su32x4*>(a.wfrag2 + (size_t)i * 8);

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
    // (a second register set -- tile t+2 requested while t+1 is in flight -- measured no gain: the YOLO stems are bound by VALU
    // issue, SiLU and index arithmetic, not by bytes in flight)
    uint32_t px[NQ][3];  // raw bits: three fp32 planes, or (PACKED) the pixel's two bf16x2 words
    auto fetch = [&](int tile) {
        const bool live = tile < a.ntiles;
        const int tl = live ? tile : 0;
        const int img = tl / per_img;
        const int t2 = tl - img * per_img;
        const int ty = t2 / a.tiles_x, tx = t2 - ty * a.tiles_x;
        const int cy0 = CONV2 ? 2 * (ty * 8) - 1 : (POOL ? 2 * (ty * 4) - 1 : ty * CTH);
        const int cx0 = TILE2 ? 2 * (tx * 16) - 1 : tx * CTW;
        const int iy0 = 2 * cy0 - a.pad, ix0 = 2 * cx0 - a.pad;
        const void* in_img = PACKED ? (const void*)(reinterpret_cast<const uint16_t*>(a.in) + (size_t)img * plane * 4)
                                    : (const void*)(a.in + (size_t)img * a.C * plane);
        __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)in_img, 0, PACKED ? plane * 8 : a.C * plane * 4, 0x00020000);
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int q = tid + 256 * i;
            const int wy = q / WW, wx = q - wy * WW;
            const int iy = iy0 + wy, ix = ix0 + wx;
            const bool ok = live && q < WH * WW && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            if (PACKED) {
                const su32x2 v = __builtin_bit_cast(su32x2, __builtin_amdgcn_raw_buffer_load_b64(rsrc, ok ? (uint32_t)((iy * a.W + ix) * 8) : 0x80000000u, 0, 0));
                px[i][0] = v.x;
                px[i][1] = v.y;
            } else {
                // branch-free: written with nested conditions the compiler puts every load under its own exec-masked branch, with waits between them
                // (conv_stem_pool_x3_kernel, profiles/r06/stem_pool_x3_phases.txt); outside the window / image / channel count: bit 31 -> past num_records -> 0
                const uint32_t off = ((uint32_t)((iy * a.W + ix) * 4) & 0x7FFFFFFFu) | (ok ? 0u : 0x80000000u);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const uint32_t cp = c < a.C ? (uint32_t)(c * plane * 4) : 0x80000000u;
                    px[i][c] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc, (off + cp) | ((off | cp) & 0x80000000u), 0, 0);
                }
            }
        }
    };

    const int gstride = gridDim.x;
#ifdef ADAS_STEM_PROF
    unsigned long long pacc__[16] = {0}, tprev__ = clock64();
#endif
    auto step = [&](const int tile) {
        const int img = tile / per_img;
        const int t2 = tile - img * per_img;
        const int ty = t2 / a.tiles_x, tx = t2 - ty * a.tiles_x;
        const int cy0 = CONV2 ? 2 * (ty * 8) - 1 : (POOL ? 2 * (ty * 4) - 1 : ty * CTH);
        const int cx0 = TILE2 ? 2 * (tx * 16) - 1 : tx * CTW;

        STP(0)
        __syncthreads();  // previous tile's readers of the window / conv tile are done (first trip: the weights are in LDS)
        STP(1)
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int q = tid + 256 * i;
            if (q < WH * WW) {
                su32x2 v;
                if (PACKED) {
                    v.x = px[i][0];
                    v.y = px[i][1];
                } else {
                    v.x = E::pack2(__uint_as_float(px[i][0]), __uint_as_float(px[i][1]));
                    v.y = E::pack2(__uint_as_float(px[i][2]), 0.f);
                }
                *reinterpret_cast<su32x2*>(win + q * 4) = v;
            }
        }
        STP(2)
        __syncthreads();
        STP(3)
        fetch(tile + gstride);  // in flight under this tile's MFMAs and pooling
        STP(4)

        // ---- MFMA: KH K-steps.  No guard on M tiles past the end (the last wave's surplus tile reads a clamped, valid window
        // address and is dropped at store time): a branch per tile stops hipcc from overlapping LDS reads with MFMAs.
        sf32x4 acc[MT][NT];
#pragma unroll
        for (int j = 0; j < MT; ++j)
#pragma unroll
            for (int i = 0; i < NT; ++i) acc[j][i] = sf32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < KH; ++r) {
            svec8 wf[NT], xf[MT];
#pragma unroll
            for (int i = 0; i < NT; ++i) wf[i] = *reinterpret_cast<const svec8*>(wl + ((i * KH + r) * 64 + lane) * 8);
#pragma unroll
            for (int j = 0; j < MT; ++j) xf[j] = *reinterpret_cast<const svec8*>(win + boff[j] + r * WW * 4);
#pragma unroll
            for (int j = 0; j < MT; ++j)
#pragma unroll
                for (int i = 0; i < NT; ++i) acc[j][i] = E::mfma(wf[i], xf[j], acc[j][i]);
        }

        STP(5)
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
        if (!TILE2) {
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
                    q.x = E::pack2(s_act<ACT>(acc[j][i][0] + bias4[i].x), s_act<ACT>(acc[j][i][1] + bias4[i].y));
                    q.y = E::pack2(s_act<ACT>(acc[j][i][2] + bias4[i].z), s_act<ACT>(acc[j][i][3] + bias4[i].w));
                    *reinterpret_cast<su32x2*>(op + i * 16) = q;
                }
            }
        } else {
            __syncthreads();  // every wave is done reading the window: the conv tile may overwrite it
            STP(6)
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                const int p = (wave * MT + j) * 16 + lrow;
                if (p >= NPIX) continue;
                const int gy = cy0 + pcy[j], gx = cx0 + pcx[j];
                const bool valid = (unsigned)gy < (unsigned)a.Ho && (unsigned)gx < (unsigned)a.Wo;  // else: pool padding = -inf
#pragma unroll
                for (int i = 0; i < NT; ++i) {
                    su32x2 q;
                    q.x = E::pack2(s_act<ACT>(acc[j][i][0] + bias4[i].x), s_act<ACT>(acc[j][i][1] + bias4[i].y));
                    q.y = E::pack2(s_act<ACT>(acc[j][i][2] + bias4[i].z), s_act<ACT>(acc[j][i][3] + bias4[i].w));
                    if (!valid) q.x = q.y = CONV2 ? 0u : E::kNegInf2;  // conv zero padding | pool padding
                    *reinterpret_cast<su32x2*>(ctile + p * CP + i * 16 + kg * 4) = q;
                }
            }
            STP(7)
            __syncthreads();
            STP(8)
            if (CONV2) {
                // ---- second conv from the LDS tile: K-step s2 = taps 2*s2 and 2*s2+1 x 16 channels (the tenth tap slot has zero weights)
                constexpr int MT2 = 2;  // 8 M tiles of 16 output pixels over 4 waves
                sf32x4 acc2[MT2][2];
                int py2[MT2], px2[MT2];
#pragma unroll
                for (int j = 0; j < MT2; ++j) {
                    const int p = (wave * MT2 + j) * 16 + lrow;
                    py2[j] = p >> 4;
                    px2[j] = p & 15;
                    acc2[j][0] = acc2[j][1] = sf32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int s2 = 0; s2 < 5; ++s2) {
                    const int t = 2 * s2 + (kg >> 1) < 9 ? 2 * s2 + (kg >> 1) : 8;
                    const int kh2 = t / 3, kw2 = t - kh2 * 3;
                    svec8 wf2[2], xf2[MT2];
#pragma unroll
                    for (int n = 0; n < 2; ++n) wf2[n] = *reinterpret_cast<const svec8*>(w2l + ((n * 5 + s2) * 64 + lane) * 8);
#pragma unroll
                    for (int j = 0; j < MT2; ++j)
                        xf2[j] = *reinterpret_cast<const svec8*>(ctile + ((2 * py2[j] + kh2) * CTW + 2 * px2[j] + kw2) * CP + (kg & 1) * 8);
#pragma unroll
                    for (int j = 0; j < MT2; ++j)
#pragma unroll
                        for (int n = 0; n < 2; ++n) acc2[j][n] = E::mfma(wf2[n], xf2[j], acc2[j][n]);
                }
                float4 b2[2];
#pragma unroll
                for (int n = 0; n < 2; ++n) b2[n] = *reinterpret_cast<const float4*>(a.bias2 + n * 16 + kg * 4);
#pragma unroll
                for (int j = 0; j < MT2; ++j) {
                    const int oy = ty * 8 + py2[j], ox = tx * 16 + px2[j];
                    if (oy >= a.Hp || ox >= a.Wp) continue;
                    uint16_t* op = a.out + ((size_t)(img * a.Hp + oy) * a.Wp + ox) * a.out_cs + a.out_coff + kg * 4;
#pragma unroll
                    for (int n = 0; n < 2; ++n) {
                        su32x2 q;
                        q.x = E::pack2(s_act<ACT>(acc2[j][n][0] + b2[n].x), s_act<ACT>(acc2[j][n][1] + b2[n].y));
                        q.y = E::pack2(s_act<ACT>(acc2[j][n][2] + b2[n].z), s_act<ACT>(acc2[j][n][3] + b2[n].w));
                        *reinterpret_cast<su32x2*>(op + n * 16) = q;
                    }
                }
                return;
            }
            constexpr int CG = NT * 2;  // 8-channel groups per pixel
            for (int it = tid; it < 64 * CG; it += 256) {
                const int pp = it / CG, cg = it - pp * CG;
                const int py = pp >> 4, pxx = pp & 15;
                const int gpy = ty * 4 + py, gpx = tx * 16 + pxx;
                if (gpy >= a.Hp || gpx >= a.Wp || cg * 8 >= a.cout) continue;
                su32x2 m0{E::kNegInf2, E::kNegInf2}, m1 = m0;
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const uint16_t* cp = ctile + ((2 * py + dy) * CTW + 2 * pxx + dx) * CP + cg * 8;
                        const su32x2 v0 = *reinterpret_cast<const su32x2*>(cp);
                        const su32x2 v1 = *reinterpret_cast<const su32x2*>(cp + 4);
                        m0.x = E::max2(m0.x, v0.x); m0.y = E::max2(m0.y, v0.y);
                        m1.x = E::max2(m1.x, v1.x); m1.y = E::max2(m1.y, v1.y);
                    }
                uint16_t* op = a.out + ((size_t)(img * a.Hp + gpy) * a.Wp + gpx) * a.out_cs + a.out_coff + cg * 8;
                *reinterpret_cast<su32x4*>(op) = su32x4{m0.x, m0.y, m1.x, m1.y};
            }
            STP(9)
        }
    };
    int tile = blockIdx.x;
    if (tile >= a.ntiles) return;
    fetch(tile);
    for (; tile < a.ntiles; tile += gstride) {
        step(tile);
#ifdef ADAS_STEM_PROF
        if (tid == 0) pacc__[15] += 1;
#endif
    }
#ifdef ADAS_STEM_PROF
    if (tid == 0)
        for (int i = 0; i < 16; ++i) atomicAdd(&g_stem_prof[i], pacc__[i]);
#endif
}

#ifdef ADAS_STEM_PROF
extern "C" int adas_debug_stem_prof(unsigned long long* out16, int reset) {
    static unsigned long long h[16];
    if (out16 && hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_stem_prof), sizeof(h)) != hipSuccess) return -1;
    if (reset) {
        memset(h, 0, sizeof(h));
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_stem_prof), h, sizeof(h)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

// -------------------------------------------------------------------------------------
bool stem_applicable(int prec, int in_c_true, int kh, int kw, int stride, int pad, int act, int res_mode, const TView& out, bool pool,
                     const TView& pool_out) {
    if (!prec_is16(prec) || in_c_true > 3 || stride != 2 || res_mode != RES_NONE) return false;
    if (!(kh == 3 || kh == 6 || kh == 7) || kw != kh) return false;  // kw <= 8 pixel slots per K step
    if (pad > kh / 2) return false;
    if (out.f32 || !(out.c == 16 || out.c == 32 || out.c == 48 || out.c == 64 || out.c == 80)) return false;   // n, s, m, l, x stems
    if (pool && out.c != 64) return false;
    const TView& o = pool ? pool_out : out;
    if ((o.cs & 7) || (o.coff & 7) || o.f32) return false;
    if (pool && (kh != 7 || out.c != 64 || act != ACT_RELU)) return false;  // the ResNet stem is the only pooled instance
    if (!pool && !(act == ACT_SILU || act == ACT_RELU || (act == ACT_LEAKY && kh == 3))) return false;
    return true;
}

size_t stem_weight_bytes(int kh, int cout) { return (size_t)((cout + 15) / 16) * kh * 64 * 8 * 2; }

// host-side packing: w = [cout][kh][kw][cs] fp32 (OHWI, channel pitch cs) -> fragment order bf16 bits
static uint16_t h_f2e(float f, int prec) { return prec == PREC_FP16 ? Fp16::host_from_f32(f) : Bf16::host_from_f32(f); }
void stem_pack_weights(const float* w, int cout, int kh, int kw, int cs, int c_true, uint16_t* dst, int prec) {
    const int NT = (cout + 15) / 16;
    for (int nt = 0; nt < NT; ++nt)
        for (int r = 0; r < kh; ++r)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 8; ++e) {
                    const int co = nt * 16 + (lane & 15), kg = lane >> 4;
                    const int s = 2 * kg + e / 4, ch = e & 3;
                    float v = 0.f;
                    if (co < cout && s < kw && ch < c_true) v = w[(((size_t)co * kh + r) * kw + s) * cs + ch];
                    dst[((size_t)(nt * kh + r) * 64 + lane) * 8 + e] = h_f2e(v, prec);
                }
}

// second conv of the fused YOLO stem: w = [32][3][3][16] fp32 (OHWI) -> [2][5][64 lanes][8] bf16; lane (cout row, k group),
// K index kk = 8*kgroup + e of step s: tap 2s + (kk >> 4), channel kk & 15; the tenth tap slot is zero
size_t stem2_weight_bytes() { return (size_t)2 * 5 * 64 * 8 * 2; }
void stem2_pack_weights(const float* w, uint16_t* dst, int prec) {
    for (int n = 0; n < 2; ++n)
        for (int s = 0; s < 5; ++s)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 8; ++e) {
                    const int co = n * 16 + (lane & 15), kk = (lane >> 4) * 8 + e;
                    const int tap = 2 * s + (kk >> 4), ch = kk & 15;
                    const float v = tap < 9 ? w[((size_t)co * 9 + tap) * 16 + ch] : 0.f;
                    dst[((size_t)(n * 5 + s) * 64 + lane) * 8 + e] = h_f2e(v, prec);
                }
}
bool stem2_applicable(int prec, int kh, int pad, int act, const TView& stem_out, int kh2, int kw2, int stride2, int pad2, int act2, int res_mode2,
                      const TView& out2) {
    if (!prec_is16(prec) || !(kh == 3 || kh == 6) || act != ACT_SILU || act2 != ACT_SILU || res_mode2 != RES_NONE) return false;
    if (stem_out.c != 16 || stem_out.f32 || out2.c != 32 || out2.f32 || (out2.cs & 7) || (out2.coff & 7)) return false;
    if (kh2 != 3 || kw2 != 3 || stride2 != 2 || pad2 != 1 || pad > kh / 2) return false;
    return out2.h == (stem_out.h + 2 - 3) / 2 + 1 && out2.w == (stem_out.w + 2 - 3) / 2 + 1;
}
hipError_t launch_conv_stem2(const float* nchw, int n, int c_true, int H, int W, int kh, int pad, const void* wfrag, const float* bias,
                             const TView& stem_out, const void* wfrag2, const float* bias2, const TView& out2, bool packed_in, int prec, hipStream_t st) {
    StemDev d;
    d.in = nchw; d.wfrag = (const uint16_t*)wfrag; d.bias = bias;
    d.out = (uint16_t*)out2.p; d.out_cs = out2.cs; d.out_coff = out2.coff; d.cout = stem_out.c;
    d.N = n; d.C = c_true; d.H = H; d.W = W; d.Ho = stem_out.h; d.Wo = stem_out.w;
    d.Hp = out2.h; d.Wp = out2.w;
    d.pad = pad;
    d.tiles_x = (d.Wp + 15) / 16; d.tiles_y = (d.Hp + 7) / 8;
    d.ntiles = n * d.tiles_x * d.tiles_y;
    d.wfrag2 = (const uint16_t*)wfrag2; d.bias2 = bias2;
    if ((size_t)c_true * H * W * 4 >= (1ull << 31)) return hipErrorInvalidValue;
    const int grid = d.ntiles < 768 ? d.ntiles : 768;   // persistent: three workgroups per CU (launch bounds), one generation
    if (kh != 3 && kh != 6) return hipErrorInvalidValue;
    ADAS_DISPATCH_E16(prec == PREC_FP16, E, {
        if (kh == 3 && packed_in) hipLaunchKernelGGL((conv_stem_kernel<E, 3, 1, ACT_SILU, false, true, true>), dim3(grid), dim3(256), 0, st, d);
        else if (kh == 3) hipLaunchKernelGGL((conv_stem_kernel<E, 3, 1, ACT_SILU, false, true, false>), dim3(grid), dim3(256), 0, st, d);
        else if (packed_in) hipLaunchKernelGGL((conv_stem_kernel<E, 6, 1, ACT_SILU, false, true, true>), dim3(grid), dim3(256), 0, st, d);
        else hipLaunchKernelGGL((conv_stem_kernel<E, 6, 1, ACT_SILU, false, true, false>), dim3(grid), dim3(256), 0, st, d);
    });
    return hipGetLastError();
}

template <typename E, int KH, int NT, bool POOL>
static hipError_t stem_launch_act(const StemDev& d, int act, bool packed_in, hipStream_t st) {
    const int cap_grid = 32 * persist_slots(2);   // 1024 at the default: two generations of two workgroups per CU
    const int grid = d.ntiles < cap_grid ? d.ntiles : cap_grid;
    if constexpr (KH == 3 && !POOL) {      // LeakyReLU(0.1): the 3x3 stems only (YOLOv7)
        if (act == ACT_LEAKY) {
            if (packed_in) hipLaunchKernelGGL((conv_stem_kernel<E, KH, NT, ACT_LEAKY, POOL, false, true>), dim3(grid), dim3(256), 0, st, d);
            else hipLaunchKernelGGL((conv_stem_kernel<E, KH, NT, ACT_LEAKY, POOL, false, false>), dim3(grid), dim3(256), 0, st, d);
            return hipGetLastError();
        }
    }
    if (packed_in) {
        if (act == ACT_RELU) hipLaunchKernelGGL((conv_stem_kernel<E, KH, NT, ACT_RELU, POOL, false, true>), dim3(grid), dim3(256), 0, st, d);
        else hipLaunchKernelGGL((conv_stem_kernel<E, KH, NT, ACT_SILU, POOL, false, true>), dim3(grid), dim3(256), 0, st, d);
    } else {
        if (act == ACT_RELU) hipLaunchKernelGGL((conv_stem_kernel<E, KH, NT, ACT_RELU, POOL, false, false>), dim3(grid), dim3(256), 0, st, d);
        else hipLaunchKernelGGL((conv_stem_kernel<E, KH, NT, ACT_SILU, POOL, false, false>), dim3(grid), dim3(256), 0, st, d);
    }
    return hipGetLastError();
}

template <typename E>
static hipError_t stem_launch_shape(const StemDev& d, int kh, int nt, bool pool, int act, bool packed_in, hipStream_t st) {
    if (pool) return stem_launch_act<E, 7, 4, true>(d, act, packed_in, st);
#define STEM_CASE(KH_, NT_) \
    if (kh == KH_ && nt == NT_) return stem_launch_act<E, KH_, NT_, false>(d, act, packed_in, st);
    STEM_CASE(3, 1) STEM_CASE(3, 2) STEM_CASE(3, 3) STEM_CASE(3, 4) STEM_CASE(3, 5)
    STEM_CASE(6, 1) STEM_CASE(6, 2) STEM_CASE(6, 3) STEM_CASE(6, 4) STEM_CASE(6, 5)
    STEM_CASE(7, 4)
#undef STEM_CASE
    return hipErrorInvalidValue;
}

hipError_t launch_conv_stem(const float* nchw, int n, int c_true, int H, int W, int kh, int pad, int act, const void* wfrag,
                            const float* bias, const TView& conv_out, bool pool, const TView& pool_out, bool packed_in, int prec, hipStream_t st) {
    StemDev d;
    const TView& o = pool ? pool_out : conv_out;
    d.in = nchw; d.wfrag = (const uint16_t*)wfrag; d.bias = bias;
    d.out = (uint16_t*)o.p; d.out_cs = o.cs; d.out_coff = o.coff; d.cout = conv_out.c;
    d.N = n; d.C = c_true; d.H = H; d.W = W; d.Ho = conv_out.h; d.Wo = conv_out.w;
    d.Hp = pool ? pool_out.h : 0; d.Wp = pool ? pool_out.w : 0;
    d.pad = pad;
    d.wfrag2 = nullptr; d.bias2 = nullptr;
    if (pool) {
        d.tiles_x = (d.Wp + 15) / 16; d.tiles_y = (d.Hp + 3) / 4;
    } else {
        d.tiles_x = (d.Wo + 31) / 32; d.tiles_y = (d.Ho + 7) / 8;
    }
    d.ntiles = n * d.tiles_x * d.tiles_y;
    if ((size_t)c_true * H * W * 4 >= (1ull << 31)) return hipErrorInvalidValue;
    const int nt = (conv_out.c + 15) / 16;
    if (prec == PREC_FP16) return stem_launch_shape<Fp16>(d, kh, nt, pool, act, packed_in, st);
    return stem_launch_shape<Bf16>(d, kh, nt, pool, act, packed_in, st);
}

}  // namespace adas
