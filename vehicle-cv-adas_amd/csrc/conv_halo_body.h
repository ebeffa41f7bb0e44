// conv_halo_body.h -- the tile body of the im2col-free 3x3 halo convolution (conv_halo.hip has the description of the tiling, the LDS
// layout and the staging pipeline), shared by the per-layer kernel conv_halo_kernel and by the multi-layer persistent kernel
// conv_ml_kernel (conv_ml.hip).
#pragma once
#include "kernels.h"
#include "elem16.h"
#include <type_traits>

namespace adas {

typedef __attribute__((ext_vector_type(4))) float hf32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t hu32x4;

template <int ACT>
__device__ __forceinline__ float h_act(float v) {
    if (ACT == ACT_SILU) return v * fast_rcp(1.0f + __expf(-v));
    if (ACT == ACT_RELU) return fmaxf(v, 0.0f);
    if (ACT == ACT_LEAKY) return fmaxf(v, 0.1f * v);
    return v;
}

// act(v) + r (RES_AFTER_ACT).  SiLU's multiply and the residual add are ONE fma -- what -ffp-contract=fast makes of `v * rcp(..) + r` when
// the activation is a template parameter; spelled out so that the run-time-activation variant below (where the add sits behind a
// select the compiler cannot contract through) rounds identically.
template <int ACT>
__device__ __forceinline__ float h_act_res(float v, float r) {
    if (ACT == ACT_SILU) return __builtin_fmaf(v, fast_rcp(1.0f + __expf(-v)), r);
    return h_act<ACT>(v) + r;
}
__device__ __forceinline__ float h_act_res_rt(int act, float v, float r) {
    if (act == ACT_SILU) return __builtin_fmaf(v, fast_rcp(1.0f + __expf(-v)), r);
    if (act == ACT_RELU) return fmaxf(v, 0.0f) + r;
    if (act == ACT_LEAKY) return fmaxf(v, 0.1f * v) + r;
    return v + r;
}
// run-time activation (halo_tile<..., ACT = -1, ...>): the same expressions as h_act<ACT>
__device__ __forceinline__ float h_act_rt(int act, float v) {
    if (act == ACT_SILU) return v * fast_rcp(1.0f + __expf(-v));
    if (act == ACT_RELU) return fmaxf(v, 0.0f);
    if (act == ACT_LEAKY) return fmaxf(v, 0.1f * v);
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
    int act;                       // ACT_*: read by the multi-layer kernel only (conv_halo_kernel has it as a template parameter)
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

bool halo_fill_dev(const ConvArgs& a, HaloDev* out, int* bn_out, int* bm_out, size_t* lds_out);   // conv_halo.hip

// Epilogue memory accesses of halo_tile.  ML = false: plain global loads / stores (the per-layer kernel: unchanged code).  ML = true: raw buffer
// instructions with aux = sc1 on a resource that covers the whole tensor -- 32-bit BYTE offsets, which the multi-layer launch checks on the host.
#ifdef ADAS_ML_PLAIN   // timing experiment only (tools/ml_debug.py): the multi-layer variant with cached loads / write-back stores -- NOT coherent
#define ADAS_ML_SC1(ML) false
#else
#define ADAS_ML_SC1(ML) (ML)
#endif
struct HaloIo {
    __amdgpu_buffer_rsrc_t res, out;
};
template <bool ML>
__device__ __forceinline__ hu32x4 halo_ld16(const uint16_t* base, const __amdgpu_buffer_rsrc_t& r, size_t elem) {
    if constexpr (ADAS_ML_SC1(ML)) return __builtin_amdgcn_raw_buffer_load_b128(r, (uint32_t)(elem * 2), 0, 16);
    else return *reinterpret_cast<const hu32x4*>(base + elem);
}
template <bool ML>
__device__ __forceinline__ uint2 halo_ld8(const uint16_t* base, const __amdgpu_buffer_rsrc_t& r, size_t elem) {
    if constexpr (ADAS_ML_SC1(ML)) {
        const auto v = __builtin_amdgcn_raw_buffer_load_b64(r, (uint32_t)(elem * 2), 0, 16);
        return make_uint2(v[0], v[1]);
    } else
        return *reinterpret_cast<const uint2*>(base + elem);
}
template <bool ML>
__device__ __forceinline__ void halo_st16(uint16_t* base, const __amdgpu_buffer_rsrc_t& r, size_t elem, hu32x4 v) {
    if constexpr (ADAS_ML_SC1(ML)) __builtin_amdgcn_raw_buffer_store_b128(v, r, (uint32_t)(elem * 2), 0, 16);
    else *reinterpret_cast<hu32x4*>(base + elem) = v;
}
template <bool ML>
__device__ __forceinline__ void halo_st8(uint16_t* base, const __amdgpu_buffer_rsrc_t& r, size_t elem, uint2 v) {
    if constexpr (ADAS_ML_SC1(ML)) {
        typedef __attribute__((ext_vector_type(2))) uint32_t hu32x2;
        __builtin_amdgcn_raw_buffer_store_b64(hu32x2{v.x, v.y}, r, (uint32_t)(elem * 2), 0, 16);
    } else
        *reinterpret_cast<uint2*>(base + elem) = v;
}

// halo_tile: ONE tile (HALO_BM output pixels of one strip x BN output channels) of the conv described by `a`, computed by the calling
// 256-thread workgroup with `lds` as its scratch.  conv_halo_kernel (conv_halo.hip) maps blockIdx -> (tile, cb) and calls it once;
// conv_ml_kernel (conv_ml.hip, round 5) walks a table of (layer, tile, cb) items and calls it per item with ML = true:
//   * activations written by OTHER workgroups of the same launch are read with sc1 loads (L1-bypassing; the producers store sc1 =
//     write-through): the per-CU L1 is never refreshed by another CU's stores and the XCDs' L2s are not coherent with each other
//     (MI355X_MICROARCH.md, inter-workgroup visibility) -- window, residual and output all go through buffer instructions with aux = sc1;
//   * the activation is a run-time field (a.act) so that one instantiation serves every layer of a launch (ACT = -1).
// Everything else -- tiling, staging, swizzles, MFMA order, epilogue arithmetic -- is the same code, so an ML launch produces the bits
// the per-layer launches produce.  The caller synchronises the workgroup before the next use of `lds`.
template <typename E, int BN, int ACT, int S, int BM, bool ML>
__device__ __forceinline__ void halo_tile(const HaloDev& a, int tile, const int cb, uint16_t* lds, const int tid) {
    typedef typename E::vec8 hvec8;
    constexpr int AUX = ADAS_ML_SC1(ML) ? 16 : 0;   // buffer-instruction cache policy: 16 = sc1
    constexpr int TAPS = 9;
    constexpr int HALO_BM = BM;
    constexpr int HALO_NA = halo_maxpix(S, BM) * 4 / 256;
    constexpr int TM = HALO_BM / 64, TN = BN / 16;
    constexpr int NW = (TAPS * BN * 4 + 255) / 256;  // weight chunk loads per thread per channel chunk
    constexpr int WROWS = TAPS * BN;
    uint16_t* Aw = lds;                               // [maxpix][HALO_PIX]
    uint16_t* Ww = lds + (size_t)a.maxpix * HALO_PIX;  // [TAPS*BN][HALO_WPIX], swizzled

    const int lane = tid & 63, wave = tid >> 6;   // tid: threadIdx.x (the multi-layer kernel passes a per-item opaque copy: see conv_ml.hip)
    HPROF_INIT
    const int lrow = lane & 15, kg = lane >> 4;
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
            ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, goff[i] + (uint32_t)c0 * 2u, 0, AUX);
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
    HaloIo io;
    if constexpr (ML) {
        io.res = __builtin_amdgcn_make_buffer_rsrc((void*)(a.res ? a.res : (const uint16_t*)a.out), 0, 0x7fffffff, 0x00020000);
        io.out = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, 0x7fffffff, 0x00020000);
    }
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
                    const hu32x4 w = halo_ld16<ML>(a.res, io.res, mpix[j] * a.res_cs + a.res_coff + n0 + (i + (kg & 1)) * 16 + (kg >> 1) * 8);
                    const auto s0 = __builtin_amdgcn_permlane16_swap(w[0], w[2], false, false);
                    const auto s1 = __builtin_amdgcn_permlane16_swap(w[1], w[3], false, false);
                    rq[j][i] = make_uint2(s0[0], s1[0]);
                    rq[j][i + 1] = make_uint2(s0[1], s1[1]);
                }
                if (TN & 1)   // the unpaired last channel tile (BN = 48): 8-byte load
                    rq[j][TN - 1] = halo_ld8<ML>(a.res, io.res, mpix[j] * a.res_cs + a.res_coff + n0 + kg * 4 + (TN - 1) * 16);
            } else {
#pragma unroll
                for (int i = 0; i < TN; ++i)
                    rq[j][i] = halo_ld8<ML>(a.res, io.res, mpix[j] * a.res_cs + a.res_coff + n0 + kg * 4 + i * 16);
            }
        }
    }
    // Bias / residual / activation / stores, instantiated per activation EA.  The run-time-activation variant (ACT = -1: the multi-layer kernel)
    // must NOT decide per element: a select of the activation around every SiLU puts each v_exp / v_rcp chain in its own basic block, the 256
    // values of a lane serialise on the transcendental latency and the epilogue takes longer than the MFMA loop (measured, round 5: tiles
    // 25 us instead of 13).  It branches ONCE here: SiLU (the YOLO graphs) gets the compile-time code, any other activation the per-element form.
    auto finish_and_store = [&](auto ea_tag) {
        constexpr int EA = decltype(ea_tag)::value;
        auto actf = [&](float v) {
            if constexpr (EA >= 0) return h_act<EA>(v);
            else return h_act_rt(a.act, v);
        };
        auto actresf = [&](float v, float r) {
            if constexpr (EA >= 0) return h_act_res<EA>(v, r);
            else return h_act_res_rt(a.act, v, r);
        };
        // value of (pixel j, channel group i) after bias / residual / activation
        auto finish = [&](int i, int j, float v[4]) {
            v[0] = acc[i][j][0] + bias4[i].x; v[1] = acc[i][j][1] + bias4[i].y; v[2] = acc[i][j][2] + bias4[i].z; v[3] = acc[i][j][3] + bias4[i].w;
            if (a.res_mode != RES_NONE) {
                const uint2 q = rq[j][i];
                const float rv[4] = {E::lo(q.x), E::hi(q.x), E::lo(q.y), E::hi(q.y)};
                if (a.res_mode == RES_BEFORE_ACT) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = actf(v[k] + rv[k]);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = actresf(v[k], rv[k]);
                }
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = actf(v[k]);
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
                    const size_t oe = mpix[j] * a.out_cs + a.out_coff + c;
                    if (pok[j]) {
                        if (full_n || c + 8 <= a.cout) halo_st16<ML>((uint16_t*)a.out, io.out, oe, hu32x4{s0[0], s1[0], s0[1], s1[1]});
                        else if (c + 4 <= a.cout) halo_st8<ML>((uint16_t*)a.out, io.out, oe, make_uint2(s0[0], s1[0]));
                    }
                }
                if (TN & 1) {   // the unpaired last channel tile: 8-byte store
                    float v[4];
                    finish(TN - 1, j, v);
                    uint2 q;
                    q.x = E::pack2(v[0], v[1]);
                    q.y = E::pack2(v[2], v[3]);
                    if (pok[j] && (full_n || n0 + (TN - 1) * 16 + kg * 4 < a.cout))
                        halo_st8<ML>((uint16_t*)a.out, io.out, mpix[j] * a.out_cs + a.out_coff + n0 + kg * 4 + (TN - 1) * 16, q);
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
                        if (st_ok) halo_st8<ML>((uint16_t*)a.out, io.out, ob + i * 16, q);
                    }
                }
            }
        }
    };
    if constexpr (ACT >= 0) finish_and_store(std::integral_constant<int, ACT>{});
    else if (a.act == ACT_SILU) finish_and_store(std::integral_constant<int, ACT_SILU>{});
    else finish_and_store(std::integral_constant<int, -1>{});
    HPROF(9)  // epilogue
    HPROF_FLUSH
}


}  // namespace adas
