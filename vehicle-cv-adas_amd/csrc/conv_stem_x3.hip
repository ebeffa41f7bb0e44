// conv_stem_x3.hip -- the first layer of each network in the split precision (ADAS_PREC_FP16X3): engine-seam tensor (NCHW fp32,
// coreEngine.py:150-157) -> stride-2 conv (7x7 ResNet stem, backbone.py:50-52; 3x3 / 6x6 YOLO stems) + bias + ReLU / SiLU -> G8 NHWC.
//
// The generic split-precision path spends two launches here (NCHW -> 8-channel G8 conversion, then an implicit GEMM whose K is padded
// from 147 to 392 because Cin = 3 is stored as 8: 2.7 + 1.2 ms per 64 UFLD frames, measured).  This kernel is conv_stem.hip's scheme
// with both halves of every operand: the fp32 planes are read once, split, and kept as two zero-padded (c0, c1, c2, 0) windows in LDS
// (hi and lo); with 4-channel pixels one 16x16x32 B fragment is 8 consecutive window pixels of one tap row, so a KH-row kernel is KH
// K-steps of three MFMAs per tile pair (main += w_hi x_hi; cross += w_lo x_hi + w_hi x_lo).  Weights sit in LDS in fragment order
// (hi array, lo array) for the whole persistent workgroup; the next tile's window is fetched into registers under the current tile's
// MFMAs.  The ResNet stem takes its 3x3 s2 max-pool into the launch (conv_stem_pool_x3_kernel below).
#include "kernels.h"
#include "elem16.h"
#include <string.h>
#include <type_traits>

namespace adas {

typedef __attribute__((ext_vector_type(4))) float zf32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t zu32x2;

struct StemX3Dev {
    const float* in;        // [N][C][H][W] fp32
    const uint16_t* wfrag;  // [2 (hi | lo)][NT][KH][64 lanes][8] halves, fragment order
    const float* bias;
    x3s* out;               // NHWC G8 view
    int out_cs, out_coff, cout;
    int N, C, H, W;
    int Ho, Wo;
    int pad;
    int tiles_x, tiles_y, ntiles;
};

constexpr int SX3_WW = 72;  // window row pitch in pixels

__host__ __device__ constexpr int sx3_cth(int kh) { return kh == 7 ? 6 : 8; }   // conv tile rows (7x7: 6, so that two workgroups share a CU's LDS)
__host__ __device__ constexpr int sx3_lds_bytes(int kh, int nt) {
    return (2 * nt * kh * 512 + 2 * (2 * (sx3_cth(kh) - 1) + kh) * SX3_WW * 4) * 2;
}

template <int ACT>
__device__ __forceinline__ float sx3_act(float v) {
    if (ACT == ACT_SILU) return x3_silu(v);
    if (ACT == ACT_RELU) return fmaxf(v, 0.0f);
    if (ACT == ACT_LEAKY) return fmaxf(v, 0.1f * v);
    return v;
}

template <int KH, int NT, int ACT>
__global__ __launch_bounds__(256, 2) void conv_stem_x3_kernel(StemX3Dev a) {
    Fp16::enter();
    constexpr int CTH = sx3_cth(KH), CTW = 32;
    constexpr int NPIX = CTH * CTW;
    constexpr int NMT = NPIX / 16, MT = NMT / 4;
    static_assert(NMT % 4 == 0, "M tiles over 4 waves");
    constexpr int WW = SX3_WW, WH = 2 * (CTH - 1) + KH;
    constexpr int NQ = (WH * WW + 255) / 256;
    constexpr int WFR = NT * KH * 512;   // halves per weight array
    extern __shared__ __attribute__((aligned(16))) uint16_t sx3_lds[];
    uint16_t* wlh = sx3_lds;
    uint16_t* wll = wlh + WFR;
    uint16_t* winh = wll + WFR;
    uint16_t* winl = winh + WH * WW * 4;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 15, kg = lane >> 4;
    stage_lds16<256, 8>(wlh, a.wfrag, 2 * NT * KH * 64, tid);   // both arrays, contiguous

    int boff[MT];
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        const int p = (wave * MT + j) * 16 + lrow;
        const int cy = p / CTW, cx = p - cy * CTW;
        boff[j] = ((2 * cy) * WW + 2 * cx + 2 * kg) * 4;
    }
    const int per_img = a.tiles_x * a.tiles_y;
    const int plane = a.H * a.W;

    uint32_t px[NQ][3];
    auto fetch = [&](int tile) {
        const bool live = tile < a.ntiles;
        const int tl = live ? tile : 0;
        const int img = tl / per_img;
        const int t2 = tl - img * per_img;
        const int ty = t2 / a.tiles_x, tx = t2 - ty * a.tiles_x;
        const int iy0 = 2 * (ty * CTH) - a.pad, ix0 = 2 * (tx * CTW) - a.pad;
        const void* in_img = (const void*)(a.in + (size_t)img * a.C * plane);
        __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)in_img, 0, a.C * plane * 4, 0x00020000);
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int q = tid + 256 * i;
            const int wy = q / WW, wx = q - wy * WW;
            const int iy = iy0 + wy, ix = ix0 + wx;
            const bool ok = live && q < WH * WW && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            // branch-free: written with nested conditions the compiler puts every load under its own exec-masked branch, with waits between them
            // (conv_stem_pool_x3_kernel, profiles/r06/stem_pool_x3_phases.txt); outside the window / image / channel count: bit 31 -> past num_records -> 0
            const uint32_t off = ((uint32_t)((iy * a.W + ix) * 4) & 0x7FFFFFFFu) | (ok ? 0u : 0x80000000u);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const uint32_t cp = c < a.C ? (uint32_t)(c * plane * 4) : 0x80000000u;
                px[i][c] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc, (off + cp) | ((off | cp) & 0x80000000u), 0, 0);
            }
        }
    };

    const int gstride = gridDim.x;
    auto step = [&](const int tile) {
        const int img = tile / per_img;
        const int t2 = tile - img * per_img;
        const int ty = t2 / a.tiles_x, tx = t2 - ty * a.tiles_x;
        const int cy0 = ty * CTH, cx0 = tx * CTW;

        __syncthreads();  // previous tile's readers of the windows are done (first trip: the weights are in LDS)
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int q = tid + 256 * i;
            if (q < WH * WW) {
                _Float16 h[3], l[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) x3_split(__uint_as_float(px[i][c]), h[c], l[c]);
                zu32x2 vh, vl;
                vh.x = __builtin_bit_cast(uint32_t, e_f16x2{h[0], h[1]});
                vh.y = __builtin_bit_cast(uint32_t, e_f16x2{h[2], (_Float16)0.0f});
                vl.x = __builtin_bit_cast(uint32_t, e_f16x2{l[0], l[1]});
                vl.y = __builtin_bit_cast(uint32_t, e_f16x2{l[2], (_Float16)0.0f});
                *reinterpret_cast<zu32x2*>(winh + q * 4) = vh;
                *reinterpret_cast<zu32x2*>(winl + q * 4) = vl;
            }
        }
        __syncthreads();
        fetch(tile + gstride);  // in flight under this tile's MFMAs

        zf32x4 accm[MT][NT], accx[MT][NT];
#pragma unroll
        for (int j = 0; j < MT; ++j)
#pragma unroll
            for (int i = 0; i < NT; ++i) accm[j][i] = accx[j][i] = zf32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < KH; ++r) {
            e_u32x4 wh[NT], wl[NT], xh[MT], xl[MT];
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                wh[i] = *reinterpret_cast<const e_u32x4*>(wlh + ((i * KH + r) * 64 + lane) * 8);
                wl[i] = *reinterpret_cast<const e_u32x4*>(wll + ((i * KH + r) * 64 + lane) * 8);
            }
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                xh[j] = *reinterpret_cast<const e_u32x4*>(winh + boff[j] + r * WW * 4);
                xl[j] = *reinterpret_cast<const e_u32x4*>(winl + boff[j] + r * WW * 4);
            }
#pragma unroll
            for (int j = 0; j < MT; ++j)
#pragma unroll
                for (int i = 0; i < NT; ++i) accm[j][i] = Fp16::mfma(wh[i], xh[j], accm[j][i]);
#pragma unroll
            for (int j = 0; j < MT; ++j)
#pragma unroll
                for (int i = 0; i < NT; ++i) accx[j][i] = Fp16::mfma(wl[i], xh[j], accx[j][i]);
#pragma unroll
            for (int j = 0; j < MT; ++j)
#pragma unroll
                for (int i = 0; i < NT; ++i) accx[j][i] = Fp16::mfma(wh[i], xl[j], accx[j][i]);
        }

        // ---- epilogue: lane holds channels i*16 + kg*4 .. +3 of conv pixel (cy, cx) of the tile
        float4 bias4[NT];
#pragma unroll
        for (int i = 0; i < NT; ++i) bias4[i] = *reinterpret_cast<const float4*>(a.bias + i * 16 + kg * 4);
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            const int p = (wave * MT + j) * 16 + lrow;
            const int cy = p / CTW, cx = p - cy * CTW;
            const int oy = cy0 + cy, ox = cx0 + cx;
            if (oy >= a.Ho || ox >= a.Wo) continue;
            x3s* op = a.out + ((size_t)(img * a.Ho + oy) * a.Wo + ox) * a.out_cs + a.out_coff + kg * 4;
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                if (i * 16 + kg * 4 >= a.cout) continue;
                const float v[4] = {sx3_act<ACT>(accm[j][i][0] + accx[j][i][0] * kX3Down + bias4[i].x),
                                    sx3_act<ACT>(accm[j][i][1] + accx[j][i][1] * kX3Down + bias4[i].y),
                                    sx3_act<ACT>(accm[j][i][2] + accx[j][i][2] * kX3Down + bias4[i].z),
                                    sx3_act<ACT>(accm[j][i][3] + accx[j][i][3] * kX3Down + bias4[i].w)};
                x3_store4(op + i * 16, v);
            }
        }
    };
    int tile = blockIdx.x;
    if (tile >= a.ntiles) return;
    fetch(tile);
    for (; tile < a.ntiles; tile += gstride) step(tile);
}

// ---- the ResNet stem with its 3x3 s2 p1 max-pool (backbone.py:50-53) in one launch: 7x7 s2 conv + ReLU on the 9 x 33 conv pixels a
// 4 x 16 tile of pooled pixels needs (incl. the pool halo), kept in LDS as fp32 (conv pixels outside the image = -inf: the pool's
// padding), pooled, split and stored -- the 64-channel conv output (32.8 MB per 1600x320 frame in the split storage, written and read
// back by a separate pool launch: 0.57 ms per 64 frames at the HBM roofline, measured) never exists.  8 waves, one workgroup per CU:
// 19 M tiles dealt round-robin (tile t -> wave t % 8: every SIMD gets 4-5 tiles); weights 56 KB + max(two windows 26 KB, conv tile
// 79 KB) of LDS.
struct StemPoolX3Dev {
    const float* in;
    const uint16_t* wfrag;
    const float* bias;
    x3s* out;               // pooled NHWC G8 view
    int out_cs, out_coff;
    int N, C, H, W;
    int Ho, Wo, Hp, Wp;     // conv / pooled extents
    int pad;
    int tiles_x, tiles_y, ntiles;
};
constexpr int SP3_CTH = 9, SP3_CTW = 33, SP3_NPIX = SP3_CTH * SP3_CTW, SP3_NMT = (SP3_NPIX + 15) / 16;   // 297 conv pixels, 19 M tiles
constexpr int SP3_WH = 2 * (SP3_CTH - 1) + 7, SP3_CP = 68;                                               // window rows; conv-tile pitch (floats)
constexpr int SP3_WFR = 4 * 7 * 512;
constexpr int SP3_LDS = 2 * SP3_WFR * 2 + (SP3_NPIX * SP3_CP * 4 > 2 * SP3_WH * SX3_WW * 8 ? SP3_NPIX * SP3_CP * 4 : 2 * SP3_WH * SX3_WW * 8);

#ifdef ADAS_SP3_PROF   // scratch instrumentation (tools/experiments/stem_pool_prof.py): shader cycles of waves 0 and 3 per tile phase
__device__ unsigned long long g_sp3_prof[256][32];
#define SP3P(i)                                     \
    if (lane == 0 && (wave == 0 || wave == 3)) {    \
        const unsigned long long t__ = clock64();   \
        pacc__[i] += t__ - tprev__;                 \
        tprev__ = t__;                              \
    }
#else
#define SP3P(i)
#endif

__global__ __launch_bounds__(512, 1) void conv_stem_pool_x3_kernel(StemPoolX3Dev a) {
    Fp16::enter();
    constexpr int KH = 7, NT = 4, WW = SX3_WW, WH = SP3_WH, CTW = SP3_CTW, NPIX = SP3_NPIX, CP = SP3_CP;
    constexpr int NQ = (WH * WW + 511) / 512;
    extern __shared__ __attribute__((aligned(16))) uint16_t sx3_lds[];
    uint16_t* wlh = sx3_lds;
    uint16_t* wll = wlh + SP3_WFR;
    uint16_t* winh = wll + SP3_WFR;            // windows and conv tile share one region (never live together)
    uint16_t* winl = winh + WH * WW * 4;
    float* ctile = reinterpret_cast<float*>(winh);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 15, kg = lane >> 4;
    stage_lds16<512, 8>(wlh, a.wfrag, 2 * NT * KH * 64, tid);

    // this wave's M tiles: wave, wave + 8, wave + 16 (the third only for waves 0-2)
    int boff[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int p = (wave + 8 * j) * 16 + lrow;
        const int pc = p < NPIX ? p : NPIX - 1;
        const int cy = pc / CTW, cx = pc - cy * CTW;
        boff[j] = ((2 * cy) * WW + 2 * cx + 2 * kg) * 4;
    }
    const bool three = wave + 16 < SP3_NMT;
    const int per_img = a.tiles_x * a.tiles_y;
    const int plane = a.H * a.W;

    // window slot of this thread in pass i: (row, column) inside the window -- the same for every tile (round 6: computed once; the
    // per-tile part of an address is two adds, two range checks and a select, no divergent control flow -- the compiler had turned the
    // nested conditions into exec-masked branches with a vmcnt wait between them, 2,200 cycles per tile in the phase profile)
    int wyq[NQ], wxq[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const int q = tid + 512 * i;
        wyq[i] = q < WH * WW ? q / WW : -(1 << 20);    // (a slot past the window: a row that is never inside the image)
        wxq[i] = q - (q / WW) * WW;
    }
    uint32_t cplane[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) cplane[c] = c < a.C ? (uint32_t)(c * plane * 4) : 0x80000000u;
    uint32_t px[NQ][3];
    auto fetch = [&](int tile) {
        const bool live = tile < a.ntiles;
        const int tl = live ? tile : 0;
        const int img = tl / per_img;
        const int t2 = tl - img * per_img;
        const int ty = t2 / a.tiles_x, tx = t2 - ty * a.tiles_x;
        const int cy0 = 2 * (ty * 4) - 1, cx0 = 2 * (tx * 16) - 1;
        const int iy0 = live ? 2 * cy0 - a.pad : -(1 << 20), ix0 = 2 * cx0 - a.pad;
        const void* in_img = (const void*)(a.in + (size_t)img * a.C * plane);
        __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)in_img, 0, a.C * plane * 4, 0x00020000);
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int iy = iy0 + wyq[i], ix = ix0 + wxq[i];
            const uint32_t inside = (uint32_t)(((unsigned)iy < (unsigned)a.H) & ((unsigned)ix < (unsigned)a.W));
            const uint32_t off = ((uint32_t)((iy * a.W + ix) * 4) & 0x7FFFFFFFu) | ((inside ^ 1u) << 31);   // outside: past num_records -> reads 0
#pragma unroll
            for (int c = 0; c < 3; ++c) px[i][c] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc, (off + cplane[c]) | (cplane[c] & 0x80000000u) | (off & 0x80000000u), 0, 0);
        }
    };

    float4 bias4[NT];    // (once per workgroup: fetched per tile they sat, with their latency, between the MFMAs and the conv tile's stores)
#pragma unroll
    for (int i = 0; i < NT; ++i) bias4[i] = *reinterpret_cast<const float4*>(a.bias + i * 16 + kg * 4);
    const int gstride = gridDim.x;
#ifdef ADAS_SP3_PROF
    unsigned long long tprev__ = clock64();
    unsigned long long pacc__[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int nit__ = 0;
#endif
    auto step = [&](const int tile) {
        const int img = tile / per_img;
        const int t2 = tile - img * per_img;
        const int ty = t2 / a.tiles_x, tx = t2 - ty * a.tiles_x;
        const int cy0 = 2 * (ty * 4) - 1, cx0 = 2 * (tx * 16) - 1;

        __syncthreads();  // the previous tile's pool is done reading the shared region
        SP3P(0)
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int q = tid + 512 * i;
            if (q < WH * WW) {
                _Float16 h[3], l[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) x3_split(__uint_as_float(px[i][c]), h[c], l[c]);
                zu32x2 vh, vl;
                vh.x = __builtin_bit_cast(uint32_t, e_f16x2{h[0], h[1]});
                vh.y = __builtin_bit_cast(uint32_t, e_f16x2{h[2], (_Float16)0.0f});
                vl.x = __builtin_bit_cast(uint32_t, e_f16x2{l[0], l[1]});
                vl.y = __builtin_bit_cast(uint32_t, e_f16x2{l[2], (_Float16)0.0f});
                *reinterpret_cast<zu32x2*>(winh + q * 4) = vh;
                *reinterpret_cast<zu32x2*>(winl + q * 4) = vl;
            }
        }
        __syncthreads();
        SP3P(1)
        // (measured and dropped, round 6: the two waves of a SIMD issuing these loads at opposite ends of the MFMA phase -- the loads then
        // issue at half speed beside the partner's MFMAs, +7 % per tile: tools/experiments/stem_pool_prof.py)
        fetch(tile + gstride);
        SP3P(2)

        zf32x4 accm[3][NT], accx[3][NT];
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int i = 0; i < NT; ++i) accm[j][i] = accx[j][i] = zf32x4{0.f, 0.f, 0.f, 0.f};
        auto mma = [&](auto mt_c) {
            constexpr int MT = decltype(mt_c)::value;
#pragma unroll
            for (int r = 0; r < KH; ++r) {
                e_u32x4 wh[NT], wl[NT], xh[MT], xl[MT];
#pragma unroll
                for (int i = 0; i < NT; ++i) {
                    wh[i] = *reinterpret_cast<const e_u32x4*>(wlh + ((i * KH + r) * 64 + lane) * 8);
                    wl[i] = *reinterpret_cast<const e_u32x4*>(wll + ((i * KH + r) * 64 + lane) * 8);
                }
#pragma unroll
                for (int j = 0; j < MT; ++j) {
                    xh[j] = *reinterpret_cast<const e_u32x4*>(winh + boff[j] + r * WW * 4);
                    xl[j] = *reinterpret_cast<const e_u32x4*>(winl + boff[j] + r * WW * 4);
                }
#pragma unroll
                for (int j = 0; j < MT; ++j)
#pragma unroll
                    for (int i = 0; i < NT; ++i) accm[j][i] = Fp16::mfma(wh[i], xh[j], accm[j][i]);
#pragma unroll
                for (int j = 0; j < MT; ++j)
#pragma unroll
                    for (int i = 0; i < NT; ++i) accx[j][i] = Fp16::mfma(wl[i], xh[j], accx[j][i]);
#pragma unroll
                for (int j = 0; j < MT; ++j)
#pragma unroll
                    for (int i = 0; i < NT; ++i) accx[j][i] = Fp16::mfma(wh[i], xl[j], accx[j][i]);
            }
        };
        if (three) mma(std::integral_constant<int, 3>{});
        else mma(std::integral_constant<int, 2>{});
        SP3P(3)

        __syncthreads();  // every wave is done reading the windows: the conv tile may overwrite them
        SP3P(4)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int p = (wave + 8 * j) * 16 + lrow;
            if (p >= NPIX || (j == 2 && !three)) continue;
            const int cy = p / CTW, cx = p - cy * CTW;
            const int gy = cy0 + cy, gx = cx0 + cx;
            const bool valid = (unsigned)gy < (unsigned)a.Ho && (unsigned)gx < (unsigned)a.Wo;   // else: the pool's padding
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                float4 v;
                v.x = valid ? fmaxf(accm[j][i][0] + accx[j][i][0] * kX3Down + bias4[i].x, 0.0f) : -3.0e38f;
                v.y = valid ? fmaxf(accm[j][i][1] + accx[j][i][1] * kX3Down + bias4[i].y, 0.0f) : -3.0e38f;
                v.z = valid ? fmaxf(accm[j][i][2] + accx[j][i][2] * kX3Down + bias4[i].z, 0.0f) : -3.0e38f;
                v.w = valid ? fmaxf(accm[j][i][3] + accx[j][i][3] * kX3Down + bias4[i].w, 0.0f) : -3.0e38f;
                *reinterpret_cast<float4*>(ctile + p * CP + i * 16 + kg * 4) = v;
            }
        }
        __syncthreads();
        SP3P(5)
        {   // pool: thread = (pooled pixel, 8-channel group): 64 x 8 = 512
            const int pp = tid >> 3, cg = tid & 7;
            const int py = pp >> 4, pxx = pp & 15;
            const int gpy = ty * 4 + py, gpx = tx * 16 + pxx;
            if (gpy < a.Hp && gpx < a.Wp) {
                float m[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) m[e] = -3.0e38f;
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const float* cp = ctile + ((2 * py + dy) * CTW + 2 * pxx + dx) * CP + cg * 8;
                        const float4 v0 = *reinterpret_cast<const float4*>(cp), v1 = *reinterpret_cast<const float4*>(cp + 4);
                        m[0] = fmaxf(m[0], v0.x); m[1] = fmaxf(m[1], v0.y); m[2] = fmaxf(m[2], v0.z); m[3] = fmaxf(m[3], v0.w);
                        m[4] = fmaxf(m[4], v1.x); m[5] = fmaxf(m[5], v1.y); m[6] = fmaxf(m[6], v1.z); m[7] = fmaxf(m[7], v1.w);
                    }
                x3_store8(a.out + ((size_t)(img * a.Hp + gpy) * a.Wp + gpx) * a.out_cs + a.out_coff + cg * 8, m);
            }
        }
    };
    int tile = blockIdx.x;
    if (tile >= a.ntiles) return;
    fetch(tile);
#ifdef ADAS_SP3_PROF
    for (; tile < a.ntiles; tile += gstride) { step(tile); SP3P(6) ++nit__; }
    if (lane == 0 && (wave == 0 || wave == 3)) {
        unsigned long long* b__ = g_sp3_prof[blockIdx.x & 255] + (wave ? 16 : 0);
        for (int i__ = 0; i__ < 7; ++i__) atomicAdd(&b__[i__], pacc__[i__]);
        atomicAdd(&b__[7], (unsigned long long)nit__);
    }
#else
    for (; tile < a.ntiles; tile += gstride) step(tile);
#endif
}

#ifdef ADAS_SP3_PROF
extern "C" int adas_debug_sp3_prof(unsigned long long* out32, int reset) {
    static unsigned long long h[256][32];
    if (out32) {
        if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_sp3_prof), sizeof(h)) != hipSuccess) return -1;
        for (int i = 0; i < 32; ++i) {
            out32[i] = 0;
            for (int b = 0; b < 256; ++b) out32[i] += h[b][i];
        }
    }
    if (reset) {
        memset(h, 0, sizeof(h));
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_sp3_prof), h, sizeof(h)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

bool stem_pool_x3_applicable(int in_c_true, int kh, int kw, int stride, int pad, int act, int res_mode, const TView& conv_out, const TView& pool_out) {
    if (in_c_true > 3 || stride != 2 || res_mode != RES_NONE || kh != 7 || kw != 7 || pad > 3 || act != ACT_RELU) return false;
    if (conv_out.c != 64 || pool_out.c != 64 || pool_out.f32 || (pool_out.cs & 7) || (pool_out.coff & 7)) return false;
    return pool_out.h == (conv_out.h + 2 - 3) / 2 + 1 && pool_out.w == (conv_out.w + 2 - 3) / 2 + 1;
}

hipError_t launch_conv_stem_pool_x3(const float* nchw, int n, int c_true, int H, int W, int pad, const void* wfrag, const float* bias,
                                    const TView& conv_out, const TView& pool_out, hipStream_t st) {
    StemPoolX3Dev d;
    d.in = nchw; d.wfrag = (const uint16_t*)wfrag; d.bias = bias;
    d.out = (x3s*)pool_out.p; d.out_cs = pool_out.cs; d.out_coff = pool_out.coff;
    d.N = n; d.C = c_true; d.H = H; d.W = W; d.Ho = conv_out.h; d.Wo = conv_out.w; d.Hp = pool_out.h; d.Wp = pool_out.w;
    d.pad = pad;
    d.tiles_x = (d.Wp + 15) / 16; d.tiles_y = (d.Hp + 3) / 4;
    d.ntiles = n * d.tiles_x * d.tiles_y;
    if ((size_t)c_true * H * W * 4 >= (1ull << 31)) return hipErrorInvalidValue;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)conv_stem_pool_x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    static int wgs = -1;   // ADAS_STEMP_X3_WGS: workgroups of the launch (default 256 = one persistent workgroup per CU; more: the later ones are dealt as CUs fall free)
    if (wgs < 0) { const char* e = getenv("ADAS_STEMP_X3_WGS"); wgs = e ? atoi(e) : 256; if (wgs < 64 || wgs > 65536) wgs = 256; }
    const int grid = d.ntiles < wgs ? d.ntiles : wgs;
    hipLaunchKernelGGL(conv_stem_pool_x3_kernel, dim3(grid), dim3(512), SP3_LDS, st, d);
    return hipGetLastError();
}

// ---- the YOLO stems with the 3x3 s2 p1 conv behind them (model.1: 16 -> 32 channels) in one launch, split precision: conv_stem.hip's
// CONV2 scheme with both halves of every operand.  The 17 x 33 stem pixels an 8 x 16 tile of the second conv needs (incl. its padding
// ring: stem pixels outside the stem's output are ZERO) stay in LDS as a hi plane and a lo plane (split after the SiLU); the second
// conv runs from there -- K step = two taps x 16 channels, three MFMAs per step and 16-channel output tile -- and only its 32-channel
// output reaches HBM: the stem's 16-channel tensor (6.5 MB per 640^2 frame in the split storage, written by one launch and read back
// by the generic split kernel: 0.29 + 0.23 ms per 64 frames) never exists.  8 waves, one persistent workgroup per CU.
struct Stem2X3Dev {
    const float* in;         // [N][C][H][W] fp32
    const uint16_t* wfrag;   // stem: [hi | lo][KH][64 lanes][8]
    const float* bias;
    const uint16_t* wfrag2;  // second conv: [hi | lo][2 tiles][5 K steps][64 lanes][8]
    const float* bias2;
    x3s* out;                // the second conv's NHWC G8 view (32 channels)
    int out_cs, out_coff;
    int N, C, H, W;
    int Ho, Wo;              // stem output
    int Hp, Wp;              // second conv output
    int pad;
    int tiles_x, tiles_y, ntiles;
};
constexpr int S2X_CTH = 17, S2X_CTW = 33, S2X_NPIX = S2X_CTH * S2X_CTW, S2X_NMT = (S2X_NPIX + 15) / 16;   // 561 stem pixels, 36 M tiles
constexpr int S2X_CP = 24;                                                                            // stem-tile pixel pitch in halves (48 B: aligned 16-byte fragment reads)
__host__ __device__ constexpr int s2x_wh(int kh) { return 2 * (S2X_CTH - 1) + kh; }
__host__ __device__ constexpr int s2x_region(int kh) {   // the windows and the stem tile share one region (never live together)
    return 2 * s2x_wh(kh) * SX3_WW * 8 > 2 * S2X_NPIX * S2X_CP * 2 ? 2 * s2x_wh(kh) * SX3_WW * 8 : 2 * S2X_NPIX * S2X_CP * 2;
}
__host__ __device__ constexpr int s2x_lds_bytes(int kh) { return 2 * kh * 1024 + 2 * 10 * 1024 + s2x_region(kh); }

// Workgroups per CU the kernel is compiled for.  The stages of a tile (window conversion, stem MFMAs, SiLU + split of the stem tile, second
// conv, SiLU + stores) are serial inside a workgroup; 80.5 KB of LDS lets two of them share a CU, which costs 10-15 spilled VGPRs at
// the 128-register bound and measured +2.2 % end to end against one workgroup per CU (5,380 vs 5,265 frames/s, same box,
// profiles/r06/ab_stem2_wgs.txt).  -DADAS_STEM2_X3_WGS=1 builds the single-workgroup form.
#ifndef ADAS_STEM2_X3_WGS
#define ADAS_STEM2_X3_WGS 2
#endif
template <int KH>
__global__ __launch_bounds__(512, 2 * ADAS_STEM2_X3_WGS) void conv_stem2_x3_kernel(Stem2X3Dev a) {
    Fp16::enter();
    constexpr int CTW = S2X_CTW, NPIX = S2X_NPIX, WW = SX3_WW, WH = s2x_wh(KH), CP = S2X_CP;
    constexpr int NQ = (WH * WW + 511) / 512;
    constexpr int MT = (S2X_NMT + 7) / 8;   // 5 M tiles per wave (36 over 8 waves: the last ones guarded)
    extern __shared__ __attribute__((aligned(16))) uint16_t sx3_lds[];
    uint16_t* wlh = sx3_lds;                 // stem weights, hi then lo (KH fragments each)
    uint16_t* wll = wlh + KH * 512;
    uint16_t* w2h = wll + KH * 512;          // second conv's weights, hi then lo (10 fragments each)
    uint16_t* w2l = w2h + 10 * 512;
    uint16_t* winh = w2l + 10 * 512;
    uint16_t* winl = winh + WH * WW * 4;
    uint16_t* cth = winh;                    // the stem tile's planes take the windows' place
    uint16_t* ctl = cth + NPIX * CP;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 15, kg = lane >> 4;
    stage_lds16<512, 2>(wlh, a.wfrag, 2 * KH * 64, tid);
    stage_lds16<512, 3>(w2h, a.wfrag2, 2 * 10 * 64, tid);

    int boff[MT];
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        const int p = (wave + 8 * j) * 16 + lrow;
        const int pc = p < NPIX ? p : NPIX - 1;
        const int cy = pc / CTW, cx = pc - cy * CTW;
        boff[j] = ((2 * cy) * WW + 2 * cx + 2 * kg) * 4;
    }
    const int per_img = a.tiles_x * a.tiles_y;
    const int plane = a.H * a.W;
    const float4 bias1 = *reinterpret_cast<const float4*>(a.bias + kg * 4);
    const float4 bias2lo = *reinterpret_cast<const float4*>(a.bias2 + kg * 4), bias2hi = *reinterpret_cast<const float4*>(a.bias2 + 16 + kg * 4);

    uint32_t px[NQ][3];
    auto fetch = [&](int tile) {
        const bool live = tile < a.ntiles;
        const int tl = live ? tile : 0;
        const int img = tl / per_img;
        const int t2 = tl - img * per_img;
        const int ty = t2 / a.tiles_x, tx = t2 - ty * a.tiles_x;
        const int cy0 = 2 * (ty * 8) - 1, cx0 = 2 * (tx * 16) - 1;   // stem-output origin of the tile (the second conv's pad 1)
        const int iy0 = 2 * cy0 - a.pad, ix0 = 2 * cx0 - a.pad;
        const void* in_img = (const void*)(a.in + (size_t)img * a.C * plane);
        __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)in_img, 0, a.C * plane * 4, 0x00020000);
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int q = tid + 512 * i;
            const int wy = q / WW, wx = q - wy * WW;
            const int iy = iy0 + wy, ix = ix0 + wx;
            const bool ok = live && q < WH * WW && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            // branch-free: written with nested conditions the compiler puts every load under its own exec-masked branch, with waits between them
            // (conv_stem_pool_x3_kernel, profiles/r06/stem_pool_x3_phases.txt); outside the window / image / channel count: bit 31 -> past num_records -> 0
            const uint32_t off = ((uint32_t)((iy * a.W + ix) * 4) & 0x7FFFFFFFu) | (ok ? 0u : 0x80000000u);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const uint32_t cp = c < a.C ? (uint32_t)(c * plane * 4) : 0x80000000u;
                px[i][c] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc, (off + cp) | ((off | cp) & 0x80000000u), 0, 0);
            }
        }
    };

    const int gstride = gridDim.x;
    int tile = blockIdx.x;
    if (tile >= a.ntiles) return;
    fetch(tile);
    for (; tile < a.ntiles; tile += gstride) {
        const int img = tile / per_img;
        const int t2 = tile - img * per_img;
        const int ty = t2 / a.tiles_x, tx = t2 - ty * a.tiles_x;
        const int cy0 = 2 * (ty * 8) - 1, cx0 = 2 * (tx * 16) - 1;

        __syncthreads();  // the previous tile's second conv is done reading the shared region (first trip: the weights are in LDS)
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int q = tid + 512 * i;
            if (q < WH * WW) {
                _Float16 h[3], l[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) x3_split(__uint_as_float(px[i][c]), h[c], l[c]);
                zu32x2 vh, vl;
                vh.x = __builtin_bit_cast(uint32_t, e_f16x2{h[0], h[1]});
                vh.y = __builtin_bit_cast(uint32_t, e_f16x2{h[2], (_Float16)0.0f});
                vl.x = __builtin_bit_cast(uint32_t, e_f16x2{l[0], l[1]});
                vl.y = __builtin_bit_cast(uint32_t, e_f16x2{l[2], (_Float16)0.0f});
                *reinterpret_cast<zu32x2*>(winh + q * 4) = vh;
                *reinterpret_cast<zu32x2*>(winl + q * 4) = vl;
            }
        }
        __syncthreads();
        fetch(tile + gstride);  // in flight under this tile's MFMAs

        // ---- stem conv on the 36 M tiles (wave w: tiles w, w + 8, ...): KH K-steps of three MFMAs
        zf32x4 accm[MT], accx[MT];
#pragma unroll
        for (int j = 0; j < MT; ++j) accm[j] = accx[j] = zf32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < KH; ++r) {
            const e_u32x4 wh = *reinterpret_cast<const e_u32x4*>(wlh + (r * 64 + lane) * 8);
            const e_u32x4 wl = *reinterpret_cast<const e_u32x4*>(wll + (r * 64 + lane) * 8);
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                if (wave + 8 * j < S2X_NMT) {   // wave-uniform
                    const e_u32x4 xh = *reinterpret_cast<const e_u32x4*>(winh + boff[j] + r * WW * 4);
                    const e_u32x4 xl = *reinterpret_cast<const e_u32x4*>(winl + boff[j] + r * WW * 4);
                    accm[j] = Fp16::mfma(wh, xh, accm[j]);
                    accx[j] = Fp16::mfma(wl, xh, accx[j]);
                    accx[j] = Fp16::mfma(wh, xl, accx[j]);
                }
            }
        }
        __syncthreads();  // every wave is done reading the windows: the stem tile may overwrite them
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            const int p = (wave + 8 * j) * 16 + lrow;
            if (wave + 8 * j >= S2X_NMT || p >= NPIX) continue;
            const int cy = p / CTW, cx = p - cy * CTW;
            const int gy = cy0 + cy, gx = cx0 + cx;
            const bool valid = (unsigned)gy < (unsigned)a.Ho && (unsigned)gx < (unsigned)a.Wo;   // else: the second conv's zero padding
            _Float16 h[4], l[4];
            const float b[4] = {bias1.x, bias1.y, bias1.z, bias1.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) x3_split(valid ? sx3_act<ACT_SILU>(accm[j][e] + accx[j][e] * kX3Down + b[e]) : 0.0f, h[e], l[e]);
            zu32x2 vh, vl;
            vh.x = __builtin_bit_cast(uint32_t, e_f16x2{h[0], h[1]}); vh.y = __builtin_bit_cast(uint32_t, e_f16x2{h[2], h[3]});
            vl.x = __builtin_bit_cast(uint32_t, e_f16x2{l[0], l[1]}); vl.y = __builtin_bit_cast(uint32_t, e_f16x2{l[2], l[3]});
            *reinterpret_cast<zu32x2*>(cth + p * CP + kg * 4) = vh;
            *reinterpret_cast<zu32x2*>(ctl + p * CP + kg * 4) = vl;
        }
        __syncthreads();

        // ---- second conv from the stem tile: wave w = output row w of the 8 x 16 tile; K step s2 = taps 2 s2 and 2 s2 + 1 x 16 channels
        // (the tenth tap slot has zero weights)
        {
            const int py2 = wave, px2 = lrow;
            zf32x4 m0{bias2lo.x, bias2lo.y, bias2lo.z, bias2lo.w}, m1{bias2hi.x, bias2hi.y, bias2hi.z, bias2hi.w};
            zf32x4 c0{0.f, 0.f, 0.f, 0.f}, c1{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s2 = 0; s2 < 5; ++s2) {
                const int t = 2 * s2 + (kg >> 1) < 9 ? 2 * s2 + (kg >> 1) : 8;
                const int kh2 = t / 3, kw2 = t - kh2 * 3;
                const int o = ((2 * py2 + kh2) * CTW + 2 * px2 + kw2) * CP + (kg & 1) * 8;
                const e_u32x4 xh = *reinterpret_cast<const e_u32x4*>(cth + o), xl = *reinterpret_cast<const e_u32x4*>(ctl + o);
                const e_u32x4 wh0 = *reinterpret_cast<const e_u32x4*>(w2h + ((0 * 5 + s2) * 64 + lane) * 8), wl0 = *reinterpret_cast<const e_u32x4*>(w2l + ((0 * 5 + s2) * 64 + lane) * 8);
                const e_u32x4 wh1 = *reinterpret_cast<const e_u32x4*>(w2h + ((1 * 5 + s2) * 64 + lane) * 8), wl1 = *reinterpret_cast<const e_u32x4*>(w2l + ((1 * 5 + s2) * 64 + lane) * 8);
                m0 = Fp16::mfma(wh0, xh, m0); c0 = Fp16::mfma(wl0, xh, c0); c0 = Fp16::mfma(wh0, xl, c0);
                m1 = Fp16::mfma(wh1, xh, m1); c1 = Fp16::mfma(wl1, xh, c1); c1 = Fp16::mfma(wh1, xl, c1);
            }
            const int oy = ty * 8 + py2, ox = tx * 16 + px2;
            if (oy < a.Hp && ox < a.Wp) {
                x3s* op = a.out + ((size_t)(img * a.Hp + oy) * a.Wp + ox) * a.out_cs + a.out_coff + kg * 4;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = sx3_act<ACT_SILU>(m0[e] + c0[e] * kX3Down);
                x3_store4(op, v);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = sx3_act<ACT_SILU>(m1[e] + c1[e] * kX3Down);
                x3_store4(op + 16, v);
            }
        }
    }
}

static bool stem2_x3_enabled() {   // ADAS_NO_STEM2_X3=1: stem and second conv as their two launches again
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("ADAS_NO_STEM2_X3");
        v = (e && e[0] == '1') ? 0 : 1;
    }
    return v == 1;
}
bool stem2_x3_applicable(int in_c_true, int kh, int pad, int act, const TView& stem_out, int kh2, int kw2, int stride2, int pad2, int act2, int res_mode2,
                         const TView& out2) {
    if (!stem2_x3_enabled() || in_c_true > 3 || !(kh == 3 || kh == 6) || act != ACT_SILU || act2 != ACT_SILU || res_mode2 != RES_NONE) return false;
    if (stem_out.c != 16 || stem_out.f32 || out2.c != 32 || out2.f32 || (out2.cs & 7) || (out2.coff & 7)) return false;
    if (kh2 != 3 || kw2 != 3 || stride2 != 2 || pad2 != 1 || pad > kh / 2) return false;
    return out2.h == (stem_out.h + 2 - 3) / 2 + 1 && out2.w == (stem_out.w + 2 - 3) / 2 + 1;
}
size_t stem2_x3_weight_bytes() { return 2 * stem2_weight_bytes(); }
// second conv: w = [32][3][3][16] fp32 (OHWI) -> [hi | lo][2][5][64 lanes][8]; K index kk = 8 kgroup + e of step s: tap 2 s + (kk >> 4), channel kk & 15
void stem2_x3_pack_weights(const float* w, uint16_t* dst) {
    const size_t arr = (size_t)2 * 5 * 64 * 8;
    for (int n = 0; n < 2; ++n)
        for (int s = 0; s < 5; ++s)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 8; ++e) {
                    const int co = n * 16 + (lane & 15), kk = (lane >> 4) * 8 + e;
                    const int tap = 2 * s + (kk >> 4), ch = kk & 15;
                    _Float16 h, l;
                    x3_split(tap < 9 ? w[((size_t)co * 9 + tap) * 16 + ch] : 0.f, h, l);
                    const size_t at = ((size_t)(n * 5 + s) * 64 + lane) * 8 + e;
                    dst[at] = __builtin_bit_cast(uint16_t, h);
                    dst[arr + at] = __builtin_bit_cast(uint16_t, l);
                }
}
hipError_t launch_conv_stem2_x3(const float* nchw, int n, int c_true, int H, int W, int kh, int pad, const void* wfrag, const float* bias,
                                const TView& stem_out, const void* wfrag2, const float* bias2, const TView& out2, hipStream_t st) {
    Stem2X3Dev d;
    d.in = nchw; d.wfrag = (const uint16_t*)wfrag; d.bias = bias; d.wfrag2 = (const uint16_t*)wfrag2; d.bias2 = bias2;
    d.out = (x3s*)out2.p; d.out_cs = out2.cs; d.out_coff = out2.coff;
    d.N = n; d.C = c_true; d.H = H; d.W = W; d.Ho = stem_out.h; d.Wo = stem_out.w; d.Hp = out2.h; d.Wp = out2.w;
    d.pad = pad;
    d.tiles_x = (d.Wp + 15) / 16; d.tiles_y = (d.Hp + 7) / 8;
    d.ntiles = n * d.tiles_x * d.tiles_y;
    if ((size_t)c_true * H * W * 4 >= (1ull << 31) || (kh != 3 && kh != 6)) return hipErrorInvalidValue;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)conv_stem2_x3_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv_stem2_x3_kernel<6>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    static int wgs2 = -1;  // ADAS_STEM2_X3_GRID: workgroups of the launch (default: the persistent 256 x workgroups-per-CU)
    if (wgs2 < 0) { const char* e = getenv("ADAS_STEM2_X3_GRID"); wgs2 = e ? atoi(e) : 256 * ADAS_STEM2_X3_WGS; if (wgs2 < 64 || wgs2 > 65536) wgs2 = 256 * ADAS_STEM2_X3_WGS; }
    const int grid = d.ntiles < wgs2 ? d.ntiles : wgs2;
    if (kh == 3) hipLaunchKernelGGL(conv_stem2_x3_kernel<3>, dim3(grid), dim3(512), s2x_lds_bytes(3), st, d);
    else hipLaunchKernelGGL(conv_stem2_x3_kernel<6>, dim3(grid), dim3(512), s2x_lds_bytes(6), st, d);
    return hipGetLastError();
}

// -------------------------------------------------------------------------------------
bool stem_x3_applicable(int in_c_true, int kh, int kw, int stride, int pad, int act, int res_mode, const TView& out) {
    if (in_c_true > 3 || stride != 2 || res_mode != RES_NONE) return false;
    if (!(kh == 3 || kh == 6 || kh == 7) || kw != kh || pad > kh / 2) return false;
    if (out.f32 || (out.cs & 7) || (out.coff & 7)) return false;
    if (kh == 7) return out.c == 64 && act == ACT_RELU;
    return (out.c == 16 || out.c == 32 || out.c == 48 || out.c == 64 || out.c == 80) && (act == ACT_SILU || act == ACT_RELU || act == ACT_LEAKY);
}

size_t stem_x3_weight_bytes(int kh, int cout) { return 2 * stem_weight_bytes(kh, cout); }

// host-side packing: w = [cout][kh][kw][cs] fp32 (OHWI) -> fragment order, hi array then lo array (stem_pack_weights' order)
void stem_x3_pack_weights(const float* w, int cout, int kh, int kw, int cs, int c_true, uint16_t* dst) {
    const int NT = (cout + 15) / 16;
    const size_t arr = (size_t)NT * kh * 64 * 8;
    for (int nt = 0; nt < NT; ++nt)
        for (int r = 0; r < kh; ++r)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 8; ++e) {
                    const int co = nt * 16 + (lane & 15), kg = lane >> 4;
                    const int s = 2 * kg + e / 4, ch = e & 3;
                    float v = 0.f;
                    if (co < cout && s < kw && ch < c_true) v = w[(((size_t)co * kh + r) * kw + s) * cs + ch];
                    _Float16 h, l;
                    x3_split(v, h, l);
                    const size_t at = ((size_t)(nt * kh + r) * 64 + lane) * 8 + e;
                    dst[at] = __builtin_bit_cast(uint16_t, h);
                    dst[arr + at] = __builtin_bit_cast(uint16_t, l);
                }
}

template <int KH, int NT>
static hipError_t sx3_launch(const StemX3Dev& d, int act, hipStream_t st) {
    constexpr int lds = sx3_lds_bytes(KH, NT);
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)conv_stem_x3_kernel<KH, NT, ACT_RELU>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv_stem_x3_kernel<KH, NT, ACT_SILU>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv_stem_x3_kernel<KH, NT, ACT_LEAKY>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    const int grid = d.ntiles < 512 ? d.ntiles : 512;   // persistent: two workgroups per CU
    if (act == ACT_RELU) hipLaunchKernelGGL((conv_stem_x3_kernel<KH, NT, ACT_RELU>), dim3(grid), dim3(256), lds, st, d);
    else if (act == ACT_LEAKY) hipLaunchKernelGGL((conv_stem_x3_kernel<KH, NT, ACT_LEAKY>), dim3(grid), dim3(256), lds, st, d);
    else hipLaunchKernelGGL((conv_stem_x3_kernel<KH, NT, ACT_SILU>), dim3(grid), dim3(256), lds, st, d);
    return hipGetLastError();
}

hipError_t launch_conv_stem_x3(const float* nchw, int n, int c_true, int H, int W, int kh, int pad, int act, const void* wfrag, const float* bias,
                               const TView& out, hipStream_t st) {
    StemX3Dev d;
    d.in = nchw; d.wfrag = (const uint16_t*)wfrag; d.bias = bias;
    d.out = (x3s*)out.p; d.out_cs = out.cs; d.out_coff = out.coff; d.cout = out.c;
    d.N = n; d.C = c_true; d.H = H; d.W = W; d.Ho = out.h; d.Wo = out.w;
    d.pad = pad;
    d.tiles_x = (d.Wo + 31) / 32; d.tiles_y = (d.Ho + sx3_cth(kh) - 1) / sx3_cth(kh);
    d.ntiles = n * d.tiles_x * d.tiles_y;
    if ((size_t)c_true * H * W * 4 >= (1ull << 31)) return hipErrorInvalidValue;
    const int nt = (out.c + 15) / 16;
#define SX3_CASE(KH_, NT_) \
    if (kh == KH_ && nt == NT_) return sx3_launch<KH_, NT_>(d, act, st);
    SX3_CASE(3, 1) SX3_CASE(3, 2) SX3_CASE(3, 3) SX3_CASE(3, 4) SX3_CASE(3, 5)
    SX3_CASE(6, 1) SX3_CASE(6, 2) SX3_CASE(6, 3) SX3_CASE(6, 4) SX3_CASE(6, 5)
    SX3_CASE(7, 4)
#undef SX3_CASE
    return hipErrorInvalidValue;
}

}  // namespace adas
