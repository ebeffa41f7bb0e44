// conv_halo.hip -- stride-1 3x3 convolution with im2col-free LDS halo tiles (bf16 MFMA).
//
// The gather kernel (conv_kernels.hip) re-fetches every input pixel once per tap and synchronises
// once per 32-deep K step.  Here a workgroup stages, once per 32-channel chunk, the input WINDOW
// its 256 output pixels need (tile + halo, zero-filled outside the image) and the 9 weight slabs
// of its BN output channels into LDS; all 9 taps then read *shifted* windows straight out of LDS:
// 9 * 16 MFMAs per wave between barriers and ~1.3x instead of 9x the input bytes from L2/HBM.
//
// Tiling ("strip-linear"): the image is cut into vertical strips of width SW; a tile is 256
// consecutive pixels of one strip in row-major order (it may wrap rows).  SW is chosen per layer on
// the host to maximise useful pixels per tile under the LDS budget; 16-pixel MFMA rows need not be
// rectangles because every lane computes its own window offset.
//
// Pipeline: the global loads of chunk c+1 (window + weights, addresses precomputed per thread) are
// issued into registers before the MFMAs of chunk c and written to LDS after them (async-stage
// split), so HBM/L2 latency hides under 144 MFMAs per wave.
// LDS window pixel = 32 ch * 2 B = 64 B, unpadded, with the 16-byte channel chunk kg of window pixel p stored
// at position kg ^ (((p >> 2) & 1) << 1).  ds_read_b128 is serviced in the 16-lane groups {0-3,12-15,20-27},...
// (MI355X_MICROARCH.md LDS table), i.e. pixels {0-3,12-15} of k-group g with pixels {4-11} of k-group g^1:
// exhaustive search over XOR tables (period 4: none, period 8: this one) shows this swizzle is conflict-free
// for every alignment of 16 consecutive window pixels.  (A 96 B padded pitch is also conflict-free but costs
// 61 KB for the window; 40 KB + 36 KB of weights lets two workgroups share a CU, so one's staging, barriers
// and epilogue hide under the other's MFMAs.)
//
// Stride 2 (template S = 2: the down-sampling 3x3 convs of both networks): same structure on 128-pixel tiles
// (TM = 2); the window is ((rows-1)*2+3) x ((SW-1)*2+3) input pixels and a lane's 16 output pixels read every
// second window pixel.  At a 64 B pitch that is inherently a 2-way conflict on the activation reads (only two
// of the four 64-byte residues are touched; an 80 B pitch would be conflict-free but does not leave room for two
// workgroups per CU) -- LDS is ~20 % utilised in this kernel, so the smaller footprint wins.
#include "kernels.h"
#include "elem16.h"
#include <stdlib.h>
#include <string.h>
#include <map>
#include <mutex>
#include <tuple>

namespace adas {

typedef __attribute__((ext_vector_type(4))) float hf32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t hu32x4;

template <int ACT>
__device__ __forceinline__ float h_act(float v) {
    if (ACT == ACT_SILU) return v * __frcp_rn(1.0f + __expf(-v));
    if (ACT == ACT_RELU) return fmaxf(v, 0.0f);
    if (ACT == ACT_LEAKY) return fmaxf(v, 0.1f * v);
    return v;
}

struct HaloDev {
    const uint16_t* in;
    const uint16_t* wgt;
    const float* bias;
    void* out;
    const uint16_t* res;
    int in_cs, in_coff, cin, H, W;
    int out_cs, out_coff, cout;
    int res_cs, res_coff, res_mode;
    int pad, kpad, cin_pad;
    int SW, NS, TPS, WW, maxpix;  // strip width, strips per row, tiles per strip, window width, LDS pixels
    int Ho, Wo;                   // output extent (== H, W at stride 1)
    int out_f32;
    uint32_t mg_ww, mg_sw;        // n / WW == (n * mg_ww) >> 20 and n / SW == (n * mg_sw) >> 20 for every n the kernel divides
    int xmap;
    int ntiles, tiles8, ncb, cbg;  // workgroup id -> (tile, cout block) map: tiles, ceil(tiles/8), cout blocks, blocks kept adjacent
};

constexpr int HALO_CK = 32;
__host__ __device__ constexpr int halo_bm(int S) { return S == 1 ? 256 : 128; }
// window pixels: 2 workgroups per CU at the default tile; the 128-pixel stride-1 tiles of small layers (round 3) keep 384 (3 per CU)
__host__ __device__ constexpr int halo_maxpix(int S, int BM = 0) { return S == 1 ? ((BM == 128) ? 384 : 640) : 704; }
constexpr int HALO_PIX = HALO_CK;       // elements per LDS window pixel (64 B, chunk-swizzled)
constexpr int HALO_WPIX = HALO_CK;      // weight rows are unpadded (64 B) and XOR-swizzled instead: their
                                        // fragment reads always start at a 16-aligned row, so chunk kg of row r is
                                        // stored at position kg ^ g[(r>>2)&3], g = {0,2,3,1} -> all 4 lane groups
                                        // of ds_read_b128 hit 16 distinct 16-byte slots

#ifdef ADAS_HALO_PROF  // scratch instrumentation (tools/experiments/halo_prof.py): per-phase shader cycles of wave 0, accumulated in
// registers and flushed once per workgroup into one of 256 counter banks (so the atomics do not serialise the chip)
__device__ unsigned long long g_halo_prof[256][16];
#define HPROF(i)                                    \
    if (tid == 0) {                                 \
        const unsigned long long t__ = clock64();   \
        pacc__[i] += t__ - tprev__;                 \
        tprev__ = t__;                              \
    }
#define HPROF_INIT                               \
    unsigned long long tprev__ = clock64();      \
    unsigned long long pacc__[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define HPROF_FLUSH                                                                      \
    if (tid == 0) {                                                                      \
        unsigned long long* b__ = g_halo_prof[blockIdx.x & 255];      \
        for (int i__ = 0; i__ < 10; ++i__) atomicAdd(&b__[i__], pacc__[i__]);            \
        atomicAdd(&b__[15], 1ull);                                                       \
    }
#else
#define HPROF(i)
#define HPROF_INIT
#define HPROF_FLUSH
#endif

// BM: output pixels per tile.  halo_bm(S) by default; stride-1 layers whose 256-pixel tiling gives the chip fewer than 320 workgroups
// (20x20 maps at 64 frames, everything at batch 1) run 128-pixel tiles instead: half the work per workgroup, twice the workgroups --
// their duration is one workgroup's latency chain, and a second / third resident workgroup hides it.  Measured (round 3): YOLOv8l +
// UFLDv2 one frame at a time 364 -> 453 frames/s; at 64 frames the 20x20 layers gain 1-8 us each, the 40x40 ones (448 workgroups)
// would lose 2 us each and stay on 256-pixel tiles; end to end at 64 frames within the +-1 % run-to-run noise.
template <typename E, int BN, int ACT, int S, int BM = halo_bm(S)>   // E: element tag (elem16.h)
__global__ __launch_bounds__(256, (S == 1 && BM == 128) ? 3 : 2) void conv_halo_kernel(HaloDev a) {
    E::enter();
    typedef typename E::vec8 hvec8;
    constexpr int TAPS = 9;
    constexpr int HALO_BM = BM;
    constexpr int HALO_NA = halo_maxpix(S, BM) * 4 / 256;
    constexpr int TM = HALO_BM / 64, TN = BN / 16;
    constexpr int NW = (TAPS * BN * 4 + 255) / 256;  // weight chunk loads per thread per channel chunk
    constexpr int WROWS = TAPS * BN;
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    uint16_t* Aw = lds;                               // [maxpix][HALO_PIX]
    uint16_t* Ww = lds + (size_t)a.maxpix * HALO_PIX;  // [TAPS*BN][HALO_WPIX], swizzled

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    HPROF_INIT
    const int lrow = lane & 15, kg = lane >> 4;
    // Workgroup id -> (tile, cout block).  Block b is observed to run on XCD b % 8 (MI355X_MICROARCH.md, workgroup
    // dispatch); the cbg cout blocks of one tile take consecutive slots of ONE XCD, so the window they all read comes
    // from HBM once and is re-read from that XCD's L2 (x-fastest 2-D order streamed the whole input once per cout block).
    // xmap: each XCD walks a contiguous range of tiles, so the halo rows shared by consecutive tiles hit its L2 too
    // (0-1.6 % over the interleaved map; ADAS_HALO_XMAP=0 restores that one).
    const int xslot = blockIdx.x >> 3;
    const int xr = xslot / a.cbg;
    const int cb = (xr / a.tiles8) * a.cbg + (xslot - xr * a.cbg);
    int tile = a.xmap ? (int)(blockIdx.x & 7) * a.tiles8 + (xr % a.tiles8) : (xr % a.tiles8) * 8 + (blockIdx.x & 7);
    if (cb >= a.ncb || tile >= a.ntiles) return;
    const int n0 = cb * BN;
    const int per_img = a.NS * a.TPS;
    const int img = tile / per_img;
    tile -= img * per_img;
    const int strip = tile / a.TPS, t = tile - strip * a.TPS;
    const int sx0 = strip * a.SW, p0 = t * HALO_BM;
    const int y_first = (int)(((uint32_t)p0 * a.mg_sw) >> 20);
    const int y_lastp = (int)(((uint32_t)(p0 + HALO_BM - 1) * a.mg_sw) >> 20);
    const int WH = (y_lastp - y_first) * S + 3;
    const int wy0 = y_first * S - a.pad, wx0 = sx0 * S - a.pad;
    const int npix4 = WH * a.WW * 4;
    const int na = (npix4 + 255) >> 8;  // window load slots this tile uses (workgroup-uniform)

    // ---- per-thread staging addresses (identical for every channel chunk).  The window is read with
    // buffer loads: an out-of-range byte offset makes the hardware return zeros, so halo pixels outside
    // the image need neither a branch nor a select (a per-element "load or zero" branch makes hipcc wait
    // vmcnt(0) per load -- cdna_hip_programming.md, "Three .s-level traps" (c)).
    const uint16_t* in_img = a.in + (size_t)img * a.H * a.W * a.in_cs + a.in_coff;
    const int img_bytes = (a.H * a.W * a.in_cs - a.in_coff) * 2;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)in_img, 0, img_bytes, 0x00020000);
    uint32_t goff[HALO_NA];  // byte offset from in_img, or 0x80000000: zero fill
#pragma unroll
    for (int i = 0; i < HALO_NA; ++i) {
        int e = tid + 256 * i;
        int pix = e >> 2, c8 = e & 3;
        int wy = (int)(((uint32_t)pix * a.mg_ww) >> 20), wx = pix - wy * a.WW;
        int iy = wy0 + wy, ix = wx0 + wx;
        bool ok = e < npix4 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        goff[i] = ok ? (uint32_t)(((iy * a.W + ix) * a.in_cs + c8 * 8) * 2) : 0x80000000u;
    }
    // weights: slab (cout tile, chunk) = WROWS rows of 64 B, contiguous (kernels.h: CONV_HALO packing); thread e = tid + 256*i
    // fetches 16 B number e of the slab -> every wave-level load is 1 KB of consecutive bytes
    const int nchunk_w = a.cin_pad >> 5;
    const uint16_t* wbase = a.wgt + (size_t)cb * nchunk_w * WROWS * 32 + tid * 8;
    int woff[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) woff[i] = ((tid >> 2) + 64 * i < WROWS) ? 256 * 8 * i : -1;
    const int gsw[4] = {0, 2, 3, 1};
    const int wst = (((tid & 3) ^ gsw[(tid >> 4) & 3])) * 8;             // swizzled store position (row>>2 == tid>>4 mod 4)
    const int wrd = lrow * HALO_WPIX + ((kg ^ gsw[(lrow >> 2) & 3])) * 8;  // swizzled per-lane fragment read offset

    // per-lane window offsets of this wave's 4 x 16 output pixels
    int apix[TM], oy[TM], ox[TM];  // window pixel index of this lane's output pixel at tap (0,0)
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        int p = p0 + (wave * TM + j) * 16 + lrow;
        int y = (int)(((uint32_t)p * a.mg_sw) >> 20), xs = p - y * a.SW;
        oy[j] = y;
        ox[j] = sx0 + xs;
        apix[j] = (y - y_first) * S * a.WW + xs * S;
    }

    hf32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = hf32x4{0.f, 0.f, 0.f, 0.f};

    hu32x4 ra[HALO_NA], rw[NW];
    // No predicates: the channel tail of the last chunk multiplies zero-padded weights, window pixels
    // outside the image come back as zeros from the buffer bounds check.
    auto gload = [&](int c0) {
#pragma unroll
        for (int i = 0; i < HALO_NA; ++i)  // unconditional: a branch around a load makes hipcc wait vmcnt(0) at the join, serialising them
            ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, goff[i] + (uint32_t)c0 * 2u, 0, 0);
#pragma unroll
        for (int i = 0; i < NW; ++i) rw[i] = *reinterpret_cast<const hu32x4*>(wbase + (woff[i] < 0 ? 0 : woff[i]) + (size_t)(c0 >> 5) * WROWS * 32);
    };
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < HALO_NA; ++i) {
            int e = tid + 256 * i;
            if (i < na && e < npix4) *reinterpret_cast<hu32x4*>(Aw + (e >> 2) * HALO_PIX + (((e & 3) ^ ((e >> 3) & 2)) << 3)) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < NW; ++i)
            if (woff[i] >= 0) *reinterpret_cast<hu32x4*>(Ww + ((tid >> 2) + 64 * i) * HALO_WPIX + wst) = rw[i];
    };

    const int nchunk = (a.cin + HALO_CK - 1) / HALO_CK;
    HPROF(0)  // setup
    gload(0);
    HPROF(1)  // first loads issued
    lstore();
    HPROF(2)  // first loads landed + LDS stores
    __syncthreads();
    HPROF(3)
    for (int cc = 0; cc < nchunk; ++cc) {
        if (cc + 1 < nchunk) gload((cc + 1) * HALO_CK);
        HPROF(4)  // prefetch issue
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int r = tap / 3, s = tap - r * 3;
            hvec8 wf[TN], xf[TM];
#pragma unroll
            for (int i = 0; i < TN; ++i)
                wf[i] = *reinterpret_cast<const hvec8*>(Ww + (tap * BN + i * 16) * HALO_WPIX + wrd);
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                const int pw = apix[j] + r * a.WW + s;
                xf[j] = *reinterpret_cast<const hvec8*>(Aw + pw * HALO_PIX + ((kg ^ ((pw >> 1) & 2)) << 3));
            }
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j)
                    acc[i][j] = E::mfma(wf[i], xf[j], acc[i][j]);
        }
        HPROF(5)  // tap loop (LDS reads + MFMA issue)
        if (cc + 1 < nchunk) {
            __syncthreads();
            HPROF(6)  // barrier: everyone done reading
            lstore();
            HPROF(7)  // LDS stores (incl. waiting for the prefetched loads)
            __syncthreads();
            HPROF(8)
        }
    }

    // ---- epilogue: lane holds channels c..c+3 of pixel (oy, ox).  All residual loads are issued up front from clamped
    // (always valid) addresses -- a load inside the bounds-check branches costs one exposed memory round trip per
    // (pixel, channel group), 16 in a row.
    float4 bias4[TN];
#pragma unroll
    for (int i = 0; i < TN; ++i) bias4[i] = *reinterpret_cast<const float4*>(a.bias + n0 + i * 16 + kg * 4);  // bias is padded to 128
    const bool full_n = n0 + BN <= a.cout;
    bool pok[TM];
    size_t mpix[TM];
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        pok[j] = oy[j] < a.Ho && ox[j] < a.Wo;
        mpix[j] = pok[j] ? ((size_t)img * a.Ho + oy[j]) * a.Wo + ox[j] : 0;
    }
    uint2 rq[TM][TN];
    if (a.res_mode != RES_NONE) {
        // 16-byte residual loads in the layout of the wide stores below (channel tile i + (kg&1), channels (kg>>1)*8..+7);
        // v_permlane16_swap is its own inverse, so the same exchange hands every lane its two 4-channel groups back
        const bool wide_res = TN >= 2 && (((a.res_cs | a.res_coff) & 7) == 0);
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            if (wide_res) {
#pragma unroll
                for (int i = 0; i + 1 < TN; i += 2) {
                    const hu32x4 w = *reinterpret_cast<const hu32x4*>(a.res + mpix[j] * a.res_cs + a.res_coff + n0 + (i + (kg & 1)) * 16 + (kg >> 1) * 8);
                    const auto s0 = __builtin_amdgcn_permlane16_swap(w[0], w[2], false, false);
                    const auto s1 = __builtin_amdgcn_permlane16_swap(w[1], w[3], false, false);
                    rq[j][i] = make_uint2(s0[0], s1[0]);
                    rq[j][i + 1] = make_uint2(s0[1], s1[1]);
                }
                if (TN & 1)   // the unpaired last channel tile (BN = 48): 8-byte load
                    rq[j][TN - 1] = *reinterpret_cast<const uint2*>(a.res + mpix[j] * a.res_cs + a.res_coff + n0 + kg * 4 + (TN - 1) * 16);
            } else {
#pragma unroll
                for (int i = 0; i < TN; ++i)
                    rq[j][i] = *reinterpret_cast<const uint2*>(a.res + mpix[j] * a.res_cs + a.res_coff + n0 + kg * 4 + i * 16);
            }
        }
    }
    // value of (pixel j, channel group i) after bias / residual / activation
    auto finish = [&](int i, int j, float v[4]) {
        v[0] = acc[i][j][0] + bias4[i].x; v[1] = acc[i][j][1] + bias4[i].y; v[2] = acc[i][j][2] + bias4[i].z; v[3] = acc[i][j][3] + bias4[i].w;
        if (a.res_mode != RES_NONE) {
            const uint2 q = rq[j][i];
            const float rv[4] = {E::lo(q.x), E::hi(q.x), E::lo(q.y), E::hi(q.y)};
            if (a.res_mode == RES_BEFORE_ACT) {
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = h_act<ACT>(v[k] + rv[k]);
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = h_act<ACT>(v[k]) + rv[k];
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = h_act<ACT>(v[k]);
        }
    };
    const bool wide = TN >= 2 && !a.out_f32 && (((a.out_cs | a.out_coff) & 7) == 0);
    if (wide) {
        // 16-byte stores: v_permlane16_swap exchanges the odd 16-lane rows of X (channel tile i) with the even rows of Y
        // (tile i+1), after which a lane owns 8 consecutive channels: tile i + (kg&1), channels (kg>>1)*8 .. +7.
#pragma unroll
        for (int j = 0; j < TM; ++j) {
#pragma unroll
            for (int i = 0; i + 1 < TN + 0; i += 2) {
                float vx[4], vy[4];
                finish(i, j, vx);
                finish(i + 1, j, vy);
                const uint32_t x0 = E::pack2(vx[0], vx[1]), x1 = E::pack2(vx[2], vx[3]);
                const uint32_t y0 = E::pack2(vy[0], vy[1]), y1 = E::pack2(vy[2], vy[3]);
                const auto s0 = __builtin_amdgcn_permlane16_swap(x0, y0, false, false);
                const auto s1 = __builtin_amdgcn_permlane16_swap(x1, y1, false, false);
                const int c = n0 + (i + (kg & 1)) * 16 + (kg >> 1) * 8;
                uint16_t* op = (uint16_t*)a.out + mpix[j] * a.out_cs + a.out_coff + c;
                if (pok[j]) {
                    if (full_n || c + 8 <= a.cout) *reinterpret_cast<hu32x4*>(op) = hu32x4{s0[0], s1[0], s0[1], s1[1]};
                    else if (c + 4 <= a.cout) *reinterpret_cast<uint2*>(op) = make_uint2(s0[0], s1[0]);
                }
            }
            if (TN & 1) {   // the unpaired last channel tile: 8-byte store
                float v[4];
                finish(TN - 1, j, v);
                uint2 q;
                q.x = E::pack2(v[0], v[1]);
                q.y = E::pack2(v[2], v[3]);
                if (pok[j] && (full_n || n0 + (TN - 1) * 16 + kg * 4 < a.cout))
                    *reinterpret_cast<uint2*>((uint16_t*)a.out + mpix[j] * a.out_cs + a.out_coff + n0 + kg * 4 + (TN - 1) * 16) = q;
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const size_t ob = mpix[j] * a.out_cs + a.out_coff + n0 + kg * 4;
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                float v[4];
                finish(i, j, v);
                const bool st_ok = pok[j] && (full_n || n0 + i * 16 + kg * 4 < a.cout);
                if (a.out_f32) {
                    if (st_ok) *reinterpret_cast<float4*>((float*)a.out + ob + i * 16) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
                    uint2 q;
                    q.x = E::pack2(v[0], v[1]);
                    q.y = E::pack2(v[2], v[3]);
                    if (st_ok) *reinterpret_cast<uint2*>((uint16_t*)a.out + ob + i * 16) = q;
                }
            }
        }
    }
    HPROF(9)  // epilogue
    HPROF_FLUSH
}

#ifdef ADAS_HALO_PROF
extern "C" int adas_debug_halo_prof(unsigned long long* out16, int reset) {
    static unsigned long long h[256][16];
    if (out16) {
        if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_halo_prof), sizeof(h)) != hipSuccess) return -1;
        for (int i = 0; i < 16; ++i) {
            out16[i] = 0;
            for (int b = 0; b < 256; ++b) out16[i] += h[b][i];
        }
    }
    if (reset) {
        memset(h, 0, sizeof(h));
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_halo_prof), h, sizeof(h)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

// -------------------------------------------------------------------------------------
// n / d == (n * magic) >> 20 for all n < nmax ?  (verified exhaustively; the kernel divides only such n)
static bool magic_ok(int d, int nmax, uint32_t* magic) {
    uint32_t m = ((1u << 20) + d - 1) / d;
    if ((uint64_t)nmax * m >= (1ull << 32)) return false;
    for (int n = 0; n < nmax; ++n)
        if ((int)(((uint32_t)n * m) >> 20) != n / d) return false;
    *magic = m;
    return true;
}

// Ho x Wo: OUTPUT extent; S: stride (1 or 2); pad = 1, 3x3.
static bool plan_halo_uncached(int Ho, int Wo, int S, int maxpix_cap, int bm, HaloPlan* best) {
    const int BM = bm > 0 ? bm : halo_bm(S), MAXPIX = maxpix_cap > 0 ? maxpix_cap : halo_maxpix(S, BM);
    // strip widths: powers of two, the whole row, and the row cut into 2 / 3 / 4 equal strips (40x200 maps: 100-wide strips fill
    // 97.7 % of their tiles' pixels, 32-wide ones 89.3 %)
    int cand[9] = {16, 32, 64, 128, 256, Wo, (Wo + 1) / 2, (Wo + 2) / 3, (Wo + 3) / 4};
    bool found = false;
    for (int k = 0; k < 9; ++k) {
        int SW = cand[k];
        if (SW < 8 || (SW > Wo && k != 5)) continue;
        if (k >= 5 && (SW == 16 || SW == 32 || SW == 64 || SW == 128 || SW == 256)) continue;
        int rows = (BM + SW - 1) / SW + ((BM % SW) ? 1 : 0);
        int WW = (SW - 1) * S + 3;
        int maxpix = ((rows - 1) * S + 3) * WW;
        if (maxpix > MAXPIX) continue;
        int NS = (Wo + SW - 1) / SW;
        int TPS = (Ho * SW + BM - 1) / BM;
        double eff = (double)Ho * Wo / ((double)NS * TPS * BM);
        uint32_t mw, ms;
        if (!magic_ok(WW, MAXPIX + 64, &mw) || !magic_ok(SW, TPS * BM + BM, &ms)) continue;
        if (!found || eff > best->eff + 1e-9 || (eff > best->eff - 1e-9 && SW > best->SW)) {
            *best = HaloPlan{SW, NS, TPS, WW, maxpix, eff, mw, ms};
            found = true;
        }
    }
    return found;
}

// one strip width, as given (experiments: ADAS_H8_SW): the same bookkeeping as a candidate of plan_halo_uncached
bool plan_halo_sw(int Ho, int Wo, int S, int SW, int maxpix_cap, HaloPlan* out) {
    const int BM = halo_bm(S), MAXPIX = maxpix_cap > 0 ? maxpix_cap : halo_maxpix(S, BM);
    if (SW < 8) return false;
    int rows = (BM + SW - 1) / SW + ((BM % SW) ? 1 : 0);
    int WW = (SW - 1) * S + 3;
    int maxpix = ((rows - 1) * S + 3) * WW;
    if (maxpix > MAXPIX) return false;
    int NS = (Wo + SW - 1) / SW;
    int TPS = (Ho * SW + BM - 1) / BM;
    double eff = (double)Ho * Wo / ((double)NS * TPS * BM);
    uint32_t mw, ms;
    if (!magic_ok(WW, MAXPIX + 64, &mw) || !magic_ok(SW, TPS * BM + BM, &ms)) return false;
    *out = HaloPlan{SW, NS, TPS, WW, maxpix, eff, mw, ms};
    return true;
}

// plans are pure functions of (Ho, Wo, S): memoised so eager launches do not redo the exhaustive checks
bool plan_halo(int Ho, int Wo, int S, HaloPlan* out, int maxpix_cap, int bm) {
    static std::mutex mu;
    static std::map<std::tuple<int, int, int, int, int>, std::pair<bool, HaloPlan>> cache;
    std::lock_guard<std::mutex> lk(mu);
    auto key = std::make_tuple(Ho, Wo, S, maxpix_cap, bm);
    auto it = cache.find(key);
    if (it == cache.end()) {
        HaloPlan p{};
        bool ok = plan_halo_uncached(Ho, Wo, S, maxpix_cap, bm, &p);
        it = cache.emplace(key, std::make_pair(ok, p)).first;
    }
    *out = it->second.second;
    return it->second.first;
}

template <typename E, int BN, int S, int BM = halo_bm(S)>
static hipError_t launch_bn(const HaloDev& d, int act, dim3 grid, size_t lds, hipStream_t st) {
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)conv_halo_kernel<E, BN, ACT_NONE, S, BM>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv_halo_kernel<E, BN, ACT_SILU, S, BM>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv_halo_kernel<E, BN, ACT_RELU, S, BM>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv_halo_kernel<E, BN, ACT_LEAKY, S, BM>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    if (act == ACT_SILU) hipLaunchKernelGGL((conv_halo_kernel<E, BN, ACT_SILU, S, BM>), grid, dim3(256), lds, st, d);
    else if (act == ACT_RELU) hipLaunchKernelGGL((conv_halo_kernel<E, BN, ACT_RELU, S, BM>), grid, dim3(256), lds, st, d);
    else if (act == ACT_LEAKY) hipLaunchKernelGGL((conv_halo_kernel<E, BN, ACT_LEAKY, S, BM>), grid, dim3(256), lds, st, d);
    else hipLaunchKernelGGL((conv_halo_kernel<E, BN, ACT_NONE, S, BM>), grid, dim3(256), lds, st, d);
    return hipGetLastError();
}
template <typename E>
static hipError_t launch_e(const HaloDev& d, int act, int stride, int bn, int bm, dim3 grid, size_t lds, hipStream_t st) {
    if (stride == 1 && bm == 128) {
        if (bn == 48) return launch_bn<E, 48, 1, 128>(d, act, grid, lds, st);
        if (bn == 64) return launch_bn<E, 64, 1, 128>(d, act, grid, lds, st);
        if (bn == 32) return launch_bn<E, 32, 1, 128>(d, act, grid, lds, st);
        return launch_bn<E, 16, 1, 128>(d, act, grid, lds, st);
    }
    if (stride == 2) {
        if (bn == 48) return launch_bn<E, 48, 2>(d, act, grid, lds, st);
        if (bn == 64) return launch_bn<E, 64, 2>(d, act, grid, lds, st);
        if (bn == 32) return launch_bn<E, 32, 2>(d, act, grid, lds, st);
        return launch_bn<E, 16, 2>(d, act, grid, lds, st);
    }
    if (bn == 48) return launch_bn<E, 48, 1>(d, act, grid, lds, st);
    if (bn == 64) return launch_bn<E, 64, 1>(d, act, grid, lds, st);
    if (bn == 32) return launch_bn<E, 32, 1>(d, act, grid, lds, st);
    return launch_bn<E, 16, 1>(d, act, grid, lds, st);
}

static int halo_min_cin() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("ADAS_HALO_MIN_CIN");
        v = e ? atoi(e) : 16;
    }
    return v;
}

static bool halo_s2_enabled() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("ADAS_NO_HALO_S2");
        v = (e && e[0] == '1') ? 0 : 1;
    }
    return v == 1;
}

// cout blocks of a tile that run back to back on one XCD: the largest divisor of ncb whose weight slabs together stay
// within half of the XCD's 4 MiB L2 (every tile re-reads them).  Measured at 64 frames (tools/experiments/cbg_sweep.sh):
// 256->256 20x100 188 -> 168 us, its stride-2 sibling 330 -> 300 us, 128->128 194 -> 190 us, 512->512 neutral at 2 or 8
// and 11 % slower with a non-divisor (the padded last group skews the dispatch order).  ADAS_HALO_CBG overrides.
static int halo_cb_group(int ncb, size_t slab_bytes) {
    static int forced = -1;
    if (forced < 0) {
        const char* e = getenv("ADAS_HALO_CBG");
        forced = e ? atoi(e) : 0;
    }
    int g = forced > 0 ? forced : (int)((2u << 20) / (slab_bytes ? slab_bytes : 1));
    if (g < 1) g = 1;
    if (g > ncb) g = ncb;
    while (ncb % g) --g;
    return g;
}

// Returns false when this kernel does not apply (caller falls back to the gather kernel).
bool halo_applicable(int kh, int kw, int stride, int pad, const TView& in, const TView& out) {
    if ((stride != 1 && stride != 2) || kh != 3 || kw != 3 || pad != 1) return false;
    if (stride == 2 && !halo_s2_enabled()) return false;
    if (in.f32 || out.h != (in.h + 2 - 3) / stride + 1 || out.w != (in.w + 2 - 3) / stride + 1) return false;
    if ((in.c & 7) || (in.cs & 7) || (in.coff & 7) || (out.c & 3) || (out.cs & 3) || (out.coff & 3)) return false;
    if (in.c < halo_min_cin()) return false;  // 16-channel layers waste half of every MFMA K step but are HBM-bound anyway
    if ((long)in.h * in.w * in.cs >= (1L << 30)) return false;  // 31-bit per-image byte offsets
    HaloPlan pl;
    return plan_halo(out.h, out.w, stride, &pl) && pl.eff >= 0.6;
}

// 128-pixel tiles for stride-1 layers whose 256-pixel tiling would leave the chip under three workgroups per CU
// ADAS_HALO_BM128 = workgroup-count threshold below which a stride-1 layer takes 128-pixel tiles (0: never)
static int halo_small_tiles() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("ADAS_HALO_BM128");
        v = e ? atoi(e) : 320;
        if (v < 0) v = 0;
    }
    return v;
}
int halo_tile_pixels(const ConvArgs& a) {
    if (a.stride != 1 || !halo_small_tiles()) return halo_bm(a.stride);
    HaloPlan p256, p128;
    if (!plan_halo(a.out.h, a.out.w, 1, &p256) || !plan_halo(a.out.h, a.out.w, 1, &p128, 0, 128) || p128.eff < 0.6) return 256;
    const int bn = a.halo_bn > 0 ? a.halo_bn : halo_bn(a.out.c);
    const long wgs = (long)a.n * p256.NS * p256.TPS * ((a.out.c + bn - 1) / bn);
    return wgs < halo_small_tiles() ? 128 : 256;
}

int plan_halo_bn(int max_n, int stride, const TView& in, const TView& out) {
    static int on = -1;
    if (on < 0) { const char* e = getenv("ADAS_NO_HALO_NARROW"); on = (e && e[0] == '1') ? 0 : 1; }
    if (!on || !halo_applicable(3, 3, stride, 1, in, out)) return 0;
    const int bn0 = halo_bn(out.c);
    HaloPlan p;
    if (!(stride == 1 ? plan_halo(out.h, out.w, 1, &p, 0, 128) : plan_halo(out.h, out.w, 2, &p))) return 0;
    const long tiles = (long)max_n * p.NS * p.TPS;
    int bn = bn0;
    while (bn > 16 && tiles * ((out.c + bn - 1) / bn) < 256) bn = bn == 48 ? 32 : bn / 2;   // a workgroup per CU at least, if the layer has them
    return bn == bn0 ? 0 : bn;
}

hipError_t launch_conv_halo(const ConvArgs& a, hipStream_t st) {
    HaloPlan pl;
    if (!halo_applicable(a.kh, a.kw, a.stride, a.pad, a.in, a.out)) return hipErrorNotSupported;
    const int bm = halo_tile_pixels(a);
    if (!plan_halo(a.out.h, a.out.w, a.stride, &pl, 0, bm == halo_bm(a.stride) ? 0 : bm)) return hipErrorNotSupported;
    HaloDev d;
    d.in = (const uint16_t*)a.in.p; d.wgt = (const uint16_t*)a.wgt; d.bias = a.bias; d.out = a.out.p;
    d.res = (const uint16_t*)a.res.p;
    d.in_cs = a.in.cs; d.in_coff = a.in.coff; d.cin = a.in.c; d.H = a.in.h; d.W = a.in.w;
    d.out_cs = a.out.cs; d.out_coff = a.out.coff; d.cout = a.out.c;
    d.res_cs = a.res.cs; d.res_coff = a.res.coff; d.res_mode = a.res_mode;
    d.pad = a.pad; d.kpad = a.kpad; d.cin_pad = (a.in.c + 31) / 32 * 32;
    d.SW = pl.SW; d.NS = pl.NS; d.TPS = pl.TPS; d.WW = pl.WW; d.maxpix = pl.maxpix;
    d.Ho = a.out.h; d.Wo = a.out.w;
    d.out_f32 = a.out.f32;
    d.mg_ww = pl.mg_ww;
    d.mg_sw = pl.mg_sw;
    const int bn = a.halo_bn > 0 ? a.halo_bn : halo_bn(a.out.c);
    d.ntiles = a.n * pl.NS * pl.TPS;
    { static int xm = -1; if (xm < 0) { const char* e = getenv("ADAS_HALO_XMAP"); xm = e ? atoi(e) : 1; } d.xmap = xm; }
    d.tiles8 = (d.ntiles + 7) / 8;
    d.ncb = (a.out.c + bn - 1) / bn;
    d.cbg = halo_cb_group(d.ncb, (size_t)d.cin_pad * 9 * bn * 2);
    dim3 grid(8 * d.tiles8 * d.cbg * ((d.ncb + d.cbg - 1) / d.cbg));
    size_t lds = ((size_t)pl.maxpix * HALO_PIX + (size_t)9 * bn * HALO_WPIX) * 2;
    if (a.prec == PREC_FP16) return launch_e<Fp16>(d, a.act, a.stride, bn, bm, grid, lds, st);
    return launch_e<Bf16>(d, a.act, a.stride, bn, bm, grid, lds, st);
}

}  // namespace adas
