// conv_halo_s2.hip -- stride-2 3x3 convolution (the down-sampling convs of both networks) for Cout % 128 == 0.
//
// conv_halo.hip's stride-2 instantiation runs at ~450 TFLOP/s, half of its stride-1 sibling, for two structural reasons
// (profiles/r02/layers_ufldv2_res18_b64_fp16.txt: the three s2 layers take as long as the s1 layers with twice their FLOPs):
//   * a stride-2 window is four times the output tile, so only 128 output pixels fit the LDS budget of two workgroups per CU:
//     TM = 2, i.e. 8 MFMAs per 6 fragment reads instead of 16 per 8 -- the wave is bound by ds_read issue, not by the MFMA pipe;
//   * a lane's 16 output pixels read every SECOND window pixel: at a 64-byte pixel pitch that is a 2-way bank conflict.
// Here the window is stored de-interleaved into its four PARITY PLANES P[a][b](i, j) = in(2i + a, 2j + b) (window coordinates):
// tap (r, s) of output pixel (y, x) reads plane (r & 1, s & 1) at (y + (r >> 1), x + (s >> 1)) -- nine taps, each a UNIT-stride
// access into one plane, so the fragment reads are the conflict-free ones of the stride-1 kernel (same XOR swizzle), and the
// de-interleave costs nothing: it is only a different LDS address in the staging store.  One 8-wave workgroup per CU owns 256
// output pixels x 128 output channels (waves 0-3 / 4-7 take the two 64-channel halves, each wave 64 pixels x 64 channels =
// 4 x 4 MFMA tiles: 16 MFMAs per 8 fragment reads); the window (4 planes, <= 1408 pixels) is staged once for both halves.
// A persistent form that fetches the next tile's first chunk under the current tile's last MFMAs was built and dropped: carrying
// the staging registers across the epilogue costs 30-50 spilled VGPRs at 4 x 4 accumulators per wave.
// Weight packing is conv_halo's (CONV_HALO, 64-channel slabs): the choice between the two kernels is made at launch time.
#include "kernels.h"
#include "elem16.h"
#include <stdlib.h>
#include <map>
#include <mutex>
#include <tuple>
#include <type_traits>

namespace adas {

typedef __attribute__((ext_vector_type(4))) float sf32x4_;
typedef __attribute__((ext_vector_type(4))) uint32_t su32x4_;

template <int ACT>
__device__ __forceinline__ float s2_act(float v) {
    if (ACT == ACT_SILU) return v * fast_rcp(1.0f + __expf(-v));
    if (ACT == ACT_RELU) return fmaxf(v, 0.0f);
    if (ACT == ACT_LEAKY) return fmaxf(v, 0.1f * v);
    return v;
}

struct S2Dev {
    const uint16_t* in;
    const uint16_t* wgt;
    const float* bias;
    uint16_t* out;
    int in_cs, in_coff, cin, H, W;
    int out_cs, out_coff, cout;
    int cin_pad;
    int SW, NS, TPS, PW, plane;  // strip width, strips per row, tiles per strip, plane width (SW + 1), plane stride in pixels (multiple of 8)
    int npix4;                   // 16-byte pieces of the four planes
    int Ho, Wo;
    uint32_t mg_pw, mg_sw, mg_plane;  // n / PW, n / SW, n / plane as (n * m) >> 20
    int ntiles, tiles8, ncb, xmap;
};

constexpr int S2_THR = 512;
constexpr int S2_BM = 256;
constexpr int S2_MAXPIX = 1408;                           // (160 KiB - 2 x 36 KiB of weights) / 64 B
constexpr int S2_NA = (S2_MAXPIX * 4 + S2_THR - 1) / S2_THR;   // window slots per thread (11)
constexpr int S2_WROWS = 9 * 64;                          // rows of one 64-channel weight slab
constexpr int S2_NW = (2 * S2_WROWS * 4) / S2_THR;        // weight slots per thread (9)

template <typename E, int ACT>
__global__ __launch_bounds__(S2_THR, 1) void conv_s2p_kernel(S2Dev a) {
    E::enter();
    typedef typename E::vec8 vec8;
    constexpr int TAPS = 9, TM = 4, TN = 4;
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    uint16_t* Aw = lds;                                    // [4 * plane][32]
    uint16_t* Ww = lds + (size_t)4 * a.plane * 32;         // [2][S2_WROWS][32], row-swizzled

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 15, kg = lane >> 4;
    const int half = wave >> 2, grp = wave & 3;
    // workgroup -> (tile, 128-channel block): the blocks of one tile sit in consecutive slots of one XCD (conv_halo.hip)
    const int xslot = blockIdx.x >> 3;
    const int xr = xslot / a.ncb;
    const int cb = xslot - xr * a.ncb;
    int tile = a.xmap ? (int)(blockIdx.x & 7) * a.tiles8 + xr : xr * 8 + (blockIdx.x & 7);
    if (tile >= a.ntiles) return;
    const int n0 = cb * 128 + half * 64;
    const int per_img = a.NS * a.TPS;
    const int img = tile / per_img;
    tile -= img * per_img;
    const int strip = tile / a.TPS, t = tile - strip * a.TPS;
    const int sx0 = strip * a.SW, p0 = t * S2_BM;
    const int y_first = (int)(((uint32_t)p0 * a.mg_sw) >> 20);
    const int wy0 = 2 * y_first - 1, wx0 = 2 * sx0 - 1;   // window origin in the input (pad 1)

    // ---- staging addresses (identical for every channel chunk); out-of-image pixels: out-of-range buffer offset -> zeros
    const uint16_t* in_img = a.in + (size_t)img * a.H * a.W * a.in_cs + a.in_coff;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)in_img, 0, (a.H * a.W * a.in_cs - a.in_coff) * 2, 0x00020000);
    uint32_t goff[S2_NA];
#pragma unroll
    for (int i = 0; i < S2_NA; ++i) {
        const int e = tid + S2_THR * i;
        const int pix = e >> 2, c8 = e & 3;
        const int q = (int)(((uint32_t)pix * a.mg_plane) >> 20);       // plane (a, b) = (q >> 1, q & 1)
        const int pp = pix - q * a.plane;
        const int py = (int)(((uint32_t)pp * a.mg_pw) >> 20), px = pp - py * a.PW;
        const int iy = wy0 + 2 * py + (q >> 1), ix = wx0 + 2 * px + (q & 1);
        const bool ok = e < a.npix4 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        goff[i] = ok ? (uint32_t)(((iy * a.W + ix) * a.in_cs + c8 * 8) * 2) : 0x80000000u;
    }
    const int nchunk_w = a.cin_pad >> 5;
    // weights: slabs (2 cb, chunk) and (2 cb + 1, chunk), each S2_WROWS contiguous 64-byte rows = 2304 16-byte pieces.  Piece
    // e = tid + 512 i of the pair goes to LDS piece e with its position within the row swizzled by the row's key; because
    // 512 is a multiple of 64 pieces the key and the position are the same for all of a thread's pieces: one base register.
    const uint16_t* wb0 = a.wgt + (size_t)(2 * cb) * nchunk_w * S2_WROWS * 32;      // workgroup-uniform
    const uint16_t* wb1 = wb0 + (size_t)nchunk_w * S2_WROWS * 32;
    const int gsw[4] = {0, 2, 3, 1};
    const int wdst0 = ((tid & ~3) + ((tid & 3) ^ gsw[(tid >> 4) & 3])) * 8;
    const int wrd = (half * S2_WROWS + lrow) * 32 + ((kg ^ gsw[(lrow >> 2) & 3]) << 3);

    // per-lane plane offsets of this wave's 4 x 16 output pixels (tap (0,0) of plane (0,0))
    int apl[TM], oy[TM], ox[TM];
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int p = p0 + (grp * TM + j) * 16 + lrow;
        const int y = (int)(((uint32_t)p * a.mg_sw) >> 20), xs = p - y * a.SW;
        oy[j] = y;
        ox[j] = sx0 + xs;
        apl[j] = (y - y_first) * a.PW + xs;
    }

    sf32x4_ acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = sf32x4_{0.f, 0.f, 0.f, 0.f};

    su32x4_ ra[S2_NA], rw[S2_NW];
    auto gload = [&](int c0) {
#pragma unroll
        for (int i = 0; i < S2_NA; ++i) ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, goff[i] + (uint32_t)c0 * 2u, 0, 0);
#pragma unroll
        for (int i = 0; i < S2_NW; ++i) {
            const int e = tid + S2_THR * i;   // only i = 4 straddles the two slabs
            const uint16_t* src = (e < S2_WROWS * 4 ? wb0 + (size_t)e * 8 : wb1 + (size_t)(e - S2_WROWS * 4) * 8) + (size_t)(c0 >> 5) * S2_WROWS * 32;
            rw[i] = *reinterpret_cast<const su32x4_*>(src);
        }
    };
    const int na = (a.npix4 + S2_THR - 1) / S2_THR;   // workgroup-uniform
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < S2_NA; ++i) {
            const int e = tid + S2_THR * i;
            // the plane stride is a multiple of 8 pixels, so the swizzle bit of the in-plane index is that of the flat index
            if (i < na && e < a.npix4) *reinterpret_cast<su32x4_*>(Aw + (e >> 2) * 32 + (((e & 3) ^ ((e >> 3) & 2)) << 3)) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < S2_NW; ++i) *reinterpret_cast<su32x4_*>(Ww + wdst0 + i * S2_THR * 8) = rw[i];
    };

    const int nchunk = (a.cin + 31) / 32;
    gload(0);
    lstore();
    __syncthreads();
    for (int cc = 0; cc < nchunk; ++cc) {
        if (cc + 1 < nchunk) gload((cc + 1) * 32);
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int r = tap / 3, s = tap - r * 3;
            const int tofs = ((r & 1) * 2 + (s & 1)) * a.plane + (r >> 1) * a.PW + (s >> 1);
            vec8 wf[TN], xf[TM];
#pragma unroll
            for (int i = 0; i < TN; ++i) wf[i] = *reinterpret_cast<const vec8*>(Ww + (tap * 64 + i * 16) * 32 + wrd);
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                const int pw = apl[j] + tofs;
                xf[j] = *reinterpret_cast<const vec8*>(Aw + pw * 32 + ((kg ^ ((pw >> 1) & 2)) << 3));
            }
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j) acc[i][j] = E::mfma(wf[i], xf[j], acc[i][j]);
        }
        if (cc + 1 < nchunk) {
            __syncthreads();
            lstore();
            __syncthreads();
        }
    }

    // ---- epilogue (conv_halo.hip): bias + activation, 16-byte stores after a v_permlane16_swap between channel tiles i, i+1
    float4 bias4[TN];
#pragma unroll
    for (int i = 0; i < TN; ++i) bias4[i] = *reinterpret_cast<const float4*>(a.bias + n0 + i * 16 + kg * 4);
    const bool wide = ((a.out_cs | a.out_coff) & 7) == 0;
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const bool pok = oy[j] < a.Ho && ox[j] < a.Wo;
        const size_t mpix = pok ? ((size_t)img * a.Ho + oy[j]) * a.Wo + ox[j] : 0;
#pragma unroll
        for (int i = 0; i < TN; i += 2) {
            float vx[4], vy[4];
            vx[0] = s2_act<ACT>(acc[i][j][0] + bias4[i].x); vx[1] = s2_act<ACT>(acc[i][j][1] + bias4[i].y);
            vx[2] = s2_act<ACT>(acc[i][j][2] + bias4[i].z); vx[3] = s2_act<ACT>(acc[i][j][3] + bias4[i].w);
            vy[0] = s2_act<ACT>(acc[i + 1][j][0] + bias4[i + 1].x); vy[1] = s2_act<ACT>(acc[i + 1][j][1] + bias4[i + 1].y);
            vy[2] = s2_act<ACT>(acc[i + 1][j][2] + bias4[i + 1].z); vy[3] = s2_act<ACT>(acc[i + 1][j][3] + bias4[i + 1].w);
            const uint32_t x0 = E::pack2(vx[0], vx[1]), x1 = E::pack2(vx[2], vx[3]);
            const uint32_t y0 = E::pack2(vy[0], vy[1]), y1 = E::pack2(vy[2], vy[3]);
            if (wide) {
                const auto s0 = __builtin_amdgcn_permlane16_swap(x0, y0, false, false);
                const auto s1 = __builtin_amdgcn_permlane16_swap(x1, y1, false, false);
                const int c = n0 + (i + (kg & 1)) * 16 + (kg >> 1) * 8;
                if (pok) *reinterpret_cast<su32x4_*>(a.out + mpix * a.out_cs + a.out_coff + c) = su32x4_{s0[0], s1[0], s0[1], s1[1]};
            } else {
                uint16_t* op = a.out + mpix * a.out_cs + a.out_coff + n0 + kg * 4;
                if (pok) {
                    *reinterpret_cast<uint2*>(op + i * 16) = make_uint2(x0, x1);
                    *reinterpret_cast<uint2*>(op + (i + 1) * 16) = make_uint2(y0, y1);
                }
            }
        }
    }
}

// -------------------------------------------------------------------------------------
struct S2Plan {
    int SW, NS, TPS, PW, plane;
    double eff;
    uint32_t mg_pw, mg_sw, mg_plane;
};

static bool s2_magic(int d, int nmax, uint32_t* magic) {
    uint32_t m = ((1u << 20) + d - 1) / d;
    if ((uint64_t)nmax * m >= (1ull << 32)) return false;
    for (int n = 0; n < nmax; ++n)
        if ((int)(((uint32_t)n * m) >> 20) != n / d) return false;
    *magic = m;
    return true;
}

static bool plan_s2_uncached(int Ho, int Wo, S2Plan* best) {
    int cand[7] = {16, 32, 64, 128, 256, Wo, (Wo + 1) / 2};
    bool found = false;
    for (int k = 0; k < 7; ++k) {
        const int SW = cand[k];
        if (SW < 8 || (SW > Wo && k != 5)) continue;
        const int rows = (S2_BM + SW - 1) / SW + ((S2_BM % SW) ? 1 : 0);
        const int PW = SW + 1, PH = rows + 1;
        const int plane = (PH * PW + 7) / 8 * 8;
        if (4 * plane > S2_MAXPIX) continue;
        const int NS = (Wo + SW - 1) / SW, TPS = (Ho * SW + S2_BM - 1) / S2_BM;
        const double eff = (double)Ho * Wo / ((double)NS * TPS * S2_BM);
        uint32_t mp, ms, ml;
        if (!s2_magic(PW, plane + 8, &mp) || !s2_magic(SW, TPS * S2_BM + S2_BM, &ms) || !s2_magic(plane, S2_NA * S2_THR / 4 + 8, &ml)) continue;
        if (!found || eff > best->eff + 1e-9 || (eff > best->eff - 1e-9 && SW > best->SW)) {
            *best = S2Plan{SW, NS, TPS, PW, plane, eff, mp, ms, ml};
            found = true;
        }
    }
    return found;
}

static bool plan_s2(int Ho, int Wo, S2Plan* out) {
    static std::mutex mu;
    static std::map<std::pair<int, int>, std::pair<bool, S2Plan>> cache;
    std::lock_guard<std::mutex> lk(mu);
    auto key = std::make_pair(Ho, Wo);
    auto it = cache.find(key);
    if (it == cache.end()) {
        S2Plan p{};
        const bool ok = plan_s2_uncached(Ho, Wo, &p);
        it = cache.emplace(key, std::make_pair(ok, p)).first;
    }
    *out = it->second.second;
    return it->second.first;
}

static bool s2p_enabled() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("ADAS_NO_HALO_S2P");
        v = (e && e[0] == '1') ? 0 : 1;
    }
    return v == 1;
}

// Launch-time choice on static shapes (conv_halo's weight packing): stride 2, 3x3, pad 1, Cout a multiple of 128, no residual.
bool halo_s2p_applicable(int kh, int kw, int stride, int pad, int res_mode, int n, const TView& in, const TView& out) {
    if (!s2p_enabled() || stride != 2 || kh != 3 || kw != 3 || pad != 1 || res_mode != RES_NONE) return false;
    if (in.f32 || out.f32 || out.h != (in.h + 2 - 3) / 2 + 1 || out.w != (in.w + 2 - 3) / 2 + 1) return false;
    if ((in.c & 7) || (in.cs & 7) || (in.coff & 7) || (out.c & 127) || (out.cs & 3) || (out.coff & 3)) return false;
    if (in.c < 16 || (long)in.h * in.w * in.cs >= (1L << 30)) return false;
    // One workgroup per CU and no second one to hide a tile's first global round trip and its epilogue: the kernel wins where a
    // tile runs many channel chunks (measured at 64 frames: Cin 256 -> 168 vs 99 us, Cin 128 -> 168 vs 138 us, Cin 64 -> no gain:
    // those stay on conv_halo's two-workgroups-per-CU instantiation) and where the launch fills the chip.
    if (in.c < 128) return false;
    S2Plan pl;
    if (!plan_s2(out.h, out.w, &pl) || pl.eff < 0.45) return false;
    return (long)n * pl.NS * pl.TPS * (out.c / 128) >= 512;
}

hipError_t launch_conv_halo_s2p(const ConvArgs& a, hipStream_t st) {
    S2Plan pl;
    if (!halo_s2p_applicable(a.kh, a.kw, a.stride, a.pad, a.res_mode, a.n, a.in, a.out) || !plan_s2(a.out.h, a.out.w, &pl)) return hipErrorNotSupported;
    S2Dev d;
    d.in = (const uint16_t*)a.in.p; d.wgt = (const uint16_t*)a.wgt; d.bias = a.bias; d.out = (uint16_t*)a.out.p;
    d.in_cs = a.in.cs; d.in_coff = a.in.coff; d.cin = a.in.c; d.H = a.in.h; d.W = a.in.w;
    d.out_cs = a.out.cs; d.out_coff = a.out.coff; d.cout = a.out.c;
    d.cin_pad = (a.in.c + 31) / 32 * 32;
    d.SW = pl.SW; d.NS = pl.NS; d.TPS = pl.TPS; d.PW = pl.PW; d.plane = pl.plane;
    d.npix4 = 4 * pl.plane * 4;
    d.Ho = a.out.h; d.Wo = a.out.w;
    d.mg_pw = pl.mg_pw; d.mg_sw = pl.mg_sw; d.mg_plane = pl.mg_plane;
    d.ntiles = a.n * pl.NS * pl.TPS;
    d.tiles8 = (d.ntiles + 7) / 8;
    d.ncb = a.out.c / 128;
    { static int xm = -1; if (xm < 0) { const char* e = getenv("ADAS_HALO_XMAP"); xm = e ? atoi(e) : 1; } d.xmap = xm; }
    const dim3 grid(8 * d.tiles8 * d.ncb);
    const size_t lds = ((size_t)4 * pl.plane * 32 + (size_t)2 * S2_WROWS * 32) * 2;
    static bool attr_done = false;
    if (!attr_done) {
#define S2_ATTR(E_, A_) (void)hipFuncSetAttribute((const void*)conv_s2p_kernel<E_, A_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
        S2_ATTR(Bf16, ACT_NONE); S2_ATTR(Bf16, ACT_SILU); S2_ATTR(Bf16, ACT_RELU); S2_ATTR(Bf16, ACT_LEAKY);
        S2_ATTR(Fp16, ACT_NONE); S2_ATTR(Fp16, ACT_SILU); S2_ATTR(Fp16, ACT_RELU); S2_ATTR(Fp16, ACT_LEAKY);
#undef S2_ATTR
        attr_done = true;
    }
    ADAS_DISPATCH_E16(a.prec == PREC_FP16, E, {
        if (a.act == ACT_SILU) hipLaunchKernelGGL((conv_s2p_kernel<E, ACT_SILU>), grid, dim3(S2_THR), lds, st, d);
        else if (a.act == ACT_RELU) hipLaunchKernelGGL((conv_s2p_kernel<E, ACT_RELU>), grid, dim3(S2_THR), lds, st, d);
        else if (a.act == ACT_LEAKY) hipLaunchKernelGGL((conv_s2p_kernel<E, ACT_LEAKY>), grid, dim3(S2_THR), lds, st, d);
        else hipLaunchKernelGGL((conv_s2p_kernel<E, ACT_NONE>), grid, dim3(S2_THR), lds, st, d);
    });
    return hipGetLastError();
}

// =====================================================================================
// Split precision (ADAS_PREC_FP16X3): the same parity-plane kernel fed with HALF-CHUNKS, as conv_halo8_x3.hip feeds its stride-1 stream.
//
// Until round 5 the exact mode sent every stride-2 3x3 conv to the generic gather kernel (conv_x3_igemm: 141-193 TFLOP/s of conv work on
// the three UFLD layers, 0.39-0.53 ms each at 64 frames).  Here a 32-channel chunk of the G8 activation tensor (128 B per pixel: four
// groups of [16 B hi | 16 B lo]) goes through the window as an H chunk (the four hi pieces) and then an L chunk (the four lo pieces);
// the weights are conv_halo8_x3's slabs ([32-channel block][half-chunk][tap][64 rows][32], rows 0-31 MAIN = hi(w), rows 32-63 CROSS =
// lo(w) 2^11 in an H chunk | hi(w) in an L chunk; launch_pack_weights_h8x3 -- the packing does not know the stride), so one workgroup
// (8 waves) owns 256 output pixels x 64 output channels: waves 0-3 / 4-7 take the two 32-channel blocks, each wave 64 pixels x
// (2 main + 2 cross) 16-row tiles.  H chunk: main += w_hi a_hi, cross += w_lo a_hi (16 MFMAs per tap); L chunk: cross += w_hi a_lo (8).
// Epilogue: act(main + 2^-11 cross) -> split -> G8 store.  The bias starts the main accumulators.
struct S2XDev {
    const unsigned char* in;    // G8: 4 bytes per channel slot
    const uint16_t* wgt;        // halo8_x3 slabs
    const float* bias;
    x3s* out;
    int in_cs, in_coff, cin, H, W;
    int out_cs, out_coff, cout;
    int nck;                     // half-chunks: 2 * cin / 32
    int SW, NS, TPS, PW, plane;
    int npix4;
    int Ho, Wo;
    uint32_t mg_pw, mg_sw, mg_plane;
    int ntiles, tiles8, ncb, xmap;   // ncb: 64-channel blocks
};

template <int ACT>
__device__ __forceinline__ float s2x_act(float v) {
    if (ACT == ACT_SILU) return x3_silu(v);   // (elem16.h: fp32-class, 12 instructions)
    if (ACT == ACT_RELU) return fmaxf(v, 0.0f);
    if (ACT == ACT_LEAKY) return fmaxf(v, 0.1f * v);
    return v;
}

#ifdef ADAS_S2X_PROF   // scratch instrumentation (tools/experiments/s2x_prof.py): shader cycles of thread 0 per workgroup phase
__device__ unsigned long long g_s2x_prof[16];
#define S2XP(i)                                     \
    if (tid == 0) {                                 \
        const unsigned long long t__ = clock64();   \
        pacc__[i] += t__ - tprev__;                 \
        tprev__ = t__;                              \
    }
#else
#define S2XP(i)
#endif

template <int ACT>
__global__ __launch_bounds__(S2_THR, 1) void conv_s2p_x3_kernel(S2XDev a) {
    Fp16::enter();
    typedef Fp16::vec8 vec8;
    constexpr int TAPS = 9, TM = 4;
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    uint16_t* Aw = lds;                                    // [4 * plane][32]: one half-chunk of the window
    uint16_t* Ww = lds + (size_t)4 * a.plane * 32;         // [2][S2_WROWS][32], row-swizzled: the two 32-channel blocks' slabs

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 15, kg = lane >> 4;
    const int half = wave >> 2, grp = wave & 3;
#ifdef ADAS_S2X_PROF
    unsigned long long tprev__ = clock64();
    unsigned long long pacc__[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    const int xslot = blockIdx.x >> 3;
    const int xr = xslot / a.ncb;
    const int cb = xslot - xr * a.ncb;
    int tile = a.xmap ? (int)(blockIdx.x & 7) * a.tiles8 + xr : xr * 8 + (blockIdx.x & 7);
    if (tile >= a.ntiles) return;
    const int n0 = cb * 64 + half * 32;
    const int per_img = a.NS * a.TPS;
    const int img = tile / per_img;
    tile -= img * per_img;
    const int strip = tile / a.TPS, t = tile - strip * a.TPS;
    const int sx0 = strip * a.SW, p0 = t * S2_BM;
    const int y_first = (int)(((uint32_t)p0 * a.mg_sw) >> 20);
    const int wy0 = 2 * y_first - 1, wx0 = 2 * sx0 - 1;   // window origin in the input (pad 1)

    // ---- staging addresses: byte offsets of K group c8's hi piece in half-chunk 0; out-of-image pixels: out-of-range offset -> zeros
    const unsigned char* in_img = a.in + ((size_t)img * a.H * a.W * a.in_cs + a.in_coff) * 4;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)in_img, 0, (a.H * a.W * a.in_cs - a.in_coff) * 4, 0x00020000);
    uint32_t goff[S2_NA];
#pragma unroll
    for (int i = 0; i < S2_NA; ++i) {
        const int e = tid + S2_THR * i;
        const int pix = e >> 2, c8 = e & 3;
        const int q = (int)(((uint32_t)pix * a.mg_plane) >> 20);       // plane (a, b) = (q >> 1, q & 1)
        const int pp = pix - q * a.plane;
        const int py = (int)(((uint32_t)pp * a.mg_pw) >> 20), px = pp - py * a.PW;
        const int iy = wy0 + 2 * py + (q >> 1), ix = wx0 + 2 * px + (q & 1);
        const bool ok = e < a.npix4 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        goff[i] = ok ? (uint32_t)((iy * a.W + ix) * a.in_cs * 4 + c8 * 32) : 0x80000000u;
    }
    // weights: slabs (block 2 cb, half-chunk ck) and (block 2 cb + 1, ck), each S2_WROWS contiguous 64-byte rows (conv_s2p_kernel's staging)
    const uint16_t* wb0 = a.wgt + (size_t)(2 * cb) * a.nck * S2_WROWS * 32;      // workgroup-uniform
    const uint16_t* wb1 = wb0 + (size_t)a.nck * S2_WROWS * 32;
    const int gsw[4] = {0, 2, 3, 1};
    const int wdst0 = ((tid & ~3) + ((tid & 3) ^ gsw[(tid >> 4) & 3])) * 8;
    const int wrd = (half * S2_WROWS + lrow) * 32 + ((kg ^ gsw[(lrow >> 2) & 3]) << 3);

    int apl[TM], oy[TM], ox[TM];
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int p = p0 + (grp * TM + j) * 16 + lrow;
        const int y = (int)(((uint32_t)p * a.mg_sw) >> 20), xs = p - y * a.SW;
        oy[j] = y;
        ox[j] = sx0 + xs;
        apl[j] = (y - y_first) * a.PW + xs;
    }

    sf32x4_ acc[4][TM];   // [0..1]: main, starts at the bias; [2..3]: cross, starts at zero
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float4 b = *reinterpret_cast<const float4*>(a.bias + n0 + i * 16 + kg * 4);
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            acc[i][j] = sf32x4_{b.x, b.y, b.z, b.w};
            acc[i + 2][j] = sf32x4_{0.f, 0.f, 0.f, 0.f};
        }
    }

    su32x4_ ra[S2_NA], rw[S2_NW];
    // Weights travel with the H half-chunks only (round 6, as conv_halo8_x3's SH form): an L half-chunk multiplies a_lo by w_hi, the MAIN
    // rows of the H slab already in LDS -- the L slabs of the packing (the same w_hi again) are not read, 73.7 of every 313 KB a
    // 32-channel chunk used to pull through L2 and the LDS write port.
    auto gload = [&](int ck, auto with_w) {
        // half-chunk ck: the hi pieces of channels 32 (ck >> 1) .., or (odd) their lo pieces 16 bytes on; a chunk is 128 bytes of a pixel
        const uint32_t cofs = (uint32_t)((ck >> 1) * 128 + (ck & 1) * 16);
#pragma unroll
        for (int i = 0; i < S2_NA; ++i) ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, goff[i] + cofs, 0, 0);
        if constexpr (!decltype(with_w)::value) return;
#pragma unroll
        for (int i = 0; i < S2_NW; ++i) {
            const int e = tid + S2_THR * i;   // only i = 4 straddles the two slabs
            const uint16_t* src = (e < S2_WROWS * 4 ? wb0 + (size_t)e * 8 : wb1 + (size_t)(e - S2_WROWS * 4) * 8) + (size_t)ck * S2_WROWS * 32;
            rw[i] = *reinterpret_cast<const su32x4_*>(src);
        }
    };
    const int na = (a.npix4 + S2_THR - 1) / S2_THR;   // workgroup-uniform
    auto lstore = [&](auto with_w) {
#pragma unroll
        for (int i = 0; i < S2_NA; ++i) {
            const int e = tid + S2_THR * i;
            if (i < na && e < a.npix4) *reinterpret_cast<su32x4_*>(Aw + (e >> 2) * 32 + (((e & 3) ^ ((e >> 3) & 2)) << 3)) = ra[i];
        }
        if constexpr (!decltype(with_w)::value) return;
#pragma unroll
        for (int i = 0; i < S2_NW; ++i) *reinterpret_cast<su32x4_*>(Ww + wdst0 + i * S2_THR * 8) = rw[i];
    };
    auto taps = [&](auto lo_c) {
        constexpr int I0 = decltype(lo_c)::value ? 2 : 0;   // L half-chunk: only the cross tiles accumulate
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int r = tap / 3, s = tap - r * 3;
            const int tofs = ((r & 1) * 2 + (s & 1)) * a.plane + (r >> 1) * a.PW + (s >> 1);
            vec8 wf[4], xf[TM];
#pragma unroll
            for (int i = I0; i < 4; ++i) wf[i] = *reinterpret_cast<const vec8*>(Ww + (tap * 64 + (i - I0) * 16) * 32 + wrd);   // (L: the H slab's MAIN rows)
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                const int pw = apl[j] + tofs;
                xf[j] = *reinterpret_cast<const vec8*>(Aw + pw * 32 + ((kg ^ ((pw >> 1) & 2)) << 3));
            }
#pragma unroll
            for (int i = I0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j) acc[i][j] = Fp16::mfma(wf[i], xf[j], acc[i][j]);
        }
    };

    // (measured and dropped, round 6: a persistent form -- 256 workgroups walking the block list, the next item's first half-chunk fetched
    // under the last nine taps -- hides the prologue the phase profile shows exposed (23 % of a 64 -> 128 channel layer), but the item loop
    // pushes the kernel from 256 VGPRs with 8 spills to 20-140 spilled registers whose scratch traffic shares vmcnt with the prefetch:
    // -1.8 % end to end at best.  tools/experiments/s2x_prof.py, profiles/r06/s2p_x3_phases.txt.)
    S2XP(0)
    gload(0, std::true_type{});
    lstore(std::true_type{});
    __syncthreads();
    S2XP(1)
    for (int ck = 0; ck < a.nck; ck += 2) {      // one 32-channel chunk per trip: its H half-chunk, then its L half-chunk
        gload(ck + 1, std::false_type{});
        taps(std::false_type{});
        S2XP(2)
        __syncthreads();
        S2XP(3)
        lstore(std::false_type{});
        __syncthreads();
        S2XP(4)
        if (ck + 2 < a.nck) gload(ck + 2, std::true_type{});
        taps(std::true_type{});
        S2XP(5)
        if (ck + 2 < a.nck) {
            __syncthreads();
            lstore(std::true_type{});
            __syncthreads();
            S2XP(6)
        }
    }

    // ---- epilogue: lane holds channels n0 + i*16 + kg*4 .. +3 of pixel (oy, ox)[j]
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const bool pok = oy[j] < a.Ho && ox[j] < a.Wo;
        const size_t mpix = pok ? ((size_t)img * a.Ho + oy[j]) * a.Wo + ox[j] : 0;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = s2x_act<ACT>(acc[i][j][e] + acc[i + 2][j][e] * kX3Down);
            const int c = n0 + i * 16 + kg * 4;
            if (pok && c < a.cout) x3_store4(a.out + mpix * a.out_cs + a.out_coff + c, v);   // (channels past cout: zero weight rows, not stored)
        }
    }
#ifdef ADAS_S2X_PROF
    S2XP(7)
    if (tid == 0) {
        for (int i__ = 0; i__ < 8; ++i__) atomicAdd(&g_s2x_prof[i__], pacc__[i__]);
        atomicAdd(&g_s2x_prof[8], 1ull);
    }
#endif
}

#ifdef ADAS_S2X_PROF
extern "C" int adas_debug_s2x_prof(unsigned long long* out16, int reset) {
    static unsigned long long h[16];
    if (out16 && hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_s2x_prof), sizeof(h)) != hipSuccess) return -1;
    if (reset) {
        for (int i = 0; i < 16; ++i) h[i] = 0;
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_s2x_prof), h, sizeof(h)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

// =====================================================================================
// conv_s2d_x3_kernel (round 6): conv_s2p_x3's successor with the window and the weights brought in by LDS-DMA.
//
// conv_s2p_x3 stages 157 KB per half-chunk through 80 of its 256 VGPRs (global -> registers -> ds_write), one workgroup per CU, four
// barriers per chunk: its phase profile (profiles/r06/s2p_x3_phases.txt) has 40 % of the deepest layer and 14 % of the shallowest at the
// MFMA bound, and a persistent form spills.  A second window buffer does not fit (the stride-2 window is 4 input pixels per output pixel:
// 83 KB per half-chunk), so the single window is recycled ONE PARITY PLANE AT A TIME: the nine taps run grouped by the plane they read,
//     group 0: tap (1,1) -> plane (1,1) | group 1: (0,1) (2,1) -> plane (0,1) | group 2: (1,0) (1,2) -> plane (1,0) | group 3: the four
//     even-even taps -> plane (0,0),
// a barrier closes each group, and right behind it every wave issues its share of THE NEXT HALF-CHUNK's pieces of the plane just
// released (buffer_load ... lds, source-side swizzle, out-of-range offsets = zero padding) -- a full half-chunk of MFMAs before they are
// read.  Weights as conv_h8x3's shared tiles: the nine tap tiles of a chunk's H slab stay through its L half-chunk (which multiplies by
// their MAIN rows), and the tiles of the next chunk -- or the next item -- are issued behind the L half-chunk's group barriers.  Every
// wave issues NP pieces per plane (a wave whose share is short repeats a piece: same bytes to the same place) and one per tap tile, so
// every s_waitcnt vmcnt is an immediate: before the barrier of group g, everything but the two most recent issue slots has landed.
// Persistent: the workgroups walk the (tile, 64-channel block) list; the next item's first half-chunk arrives under the last nine taps.
struct S2DDev {
    const void* in;
    const void* wgt;
    const float* bias;
    void* out;
    uint32_t in_bytes, wgt_bytes, out_bytes;
    int in_cs, in_coff, cin, H, W;
    int out_cs, out_coff, cout, Ho, Wo;
    int nck;                        // half-chunks: 2 * cin / 32
    int SW, NS, TPS, PW, PH, plane16;   // strip width, strips, tiles per strip; plane extent; LDS pixels per plane (a multiple of 16)
    uint32_t mg_pw, mg_sw;
    int ntiles, tiles8, ncb, xmap;
};
typedef __attribute__((address_space(3))) void* slds_vp;
typedef __attribute__((ext_vector_type(2))) float sf32x2_;
constexpr uint32_t S2D_OOB = 0xF0000000u;
constexpr int S2D_TAP = 2 * 64 * 64;          // bytes of one tap's weights in LDS (two 32-channel blocks of 64 rows)
constexpr int S2D_SLAB = 9 * 64 * 64;         // bytes of one (32-channel block, half-chunk) slab of the packing


template <int N>
__device__ __forceinline__ void s2d_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
template <int ACT>
__device__ __forceinline__ sf32x2_ s2d_act2(sf32x2_ v) {
    if (ACT == ACT_RELU) return __builtin_elementwise_max(v, sf32x2_{0.0f, 0.0f});
    return sf32x2_{s2x_act<ACT>(v[0]), s2x_act<ACT>(v[1])};
}

#ifdef ADAS_S2D_PROF   // scratch instrumentation (tools/experiments/s2x_prof.py --dma): shader cycles of wave 0 per phase
__device__ unsigned long long g_s2d_prof[16];
#define S2DP(i)                                     \
    if (tid == 0) {                                 \
        const unsigned long long t__ = clock64();   \
        pacc__[i] += t__ - tprev__;                 \
        tprev__ = t__;                              \
    }
#else
#define S2DP(i)
#endif

template <int ACT, int NP>
__global__ __launch_bounds__(S2_THR, 1) void conv_s2d_x3_kernel(S2DDev a) {
    Fp16::enter();
    typedef Fp16::vec8 vec8;
    extern __shared__ __attribute__((aligned(16))) uint8_t s2d_lds[];
    const int WOFF = 4 * a.plane16 * 64;       // window: [4 planes][plane16 pixels][64 B]; weights: [9 taps][2 blocks][64 rows][64 B]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane & 15, kg = lane >> 4;
    const int hb = wave >> 2, grp = wave & 3;
#ifdef ADAS_S2D_PROF
    unsigned long long tprev__ = clock64();
    unsigned long long pacc__[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long nitem__ = 0;
#endif
    __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, a.in_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rwg = __builtin_amdgcn_make_buffer_rsrc((void*)a.wgt, 0, a.wgt_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc((void*)a.out, 0, a.out_bytes, 0x00020000);
    const int gsw[4] = {0, 2, 3, 1};
    const uint32_t wlane = (uint32_t)((lane >> 2) * 64 + (((lane & 3) ^ gsw[(lane >> 4) & 3]) << 4));   // this lane's 16 bytes of a wave's 16 weight rows
    const uint32_t wrd = (uint32_t)(WOFF + hb * 4096 + lrow * 64 + ((kg ^ gsw[(lrow >> 2) & 3]) << 4));
    const uint32_t ch_lane = (uint32_t)((hb * 4 + (kg >> 1)) * 32 + (kg & 1) * 16);   // epilogue: 16-byte piece of the lane pair's 8-channel group

    // ---- work items: virtual block v = blockIdx.x + k gridDim.x of the (tile, 64-channel block) list (gridDim.x a multiple of 8: the XCD slot stays)
    struct Item {
        int img, cb, sx0, p0, y_first;
    };
    const int nvb = 8 * a.tiles8 * a.ncb, per_img = a.NS * a.TPS;
    auto decode = [&](int v, Item& it) -> bool {
        if (v >= nvb) return false;
        const int xslot = v >> 3, xr = xslot / a.ncb;
        int tile = a.xmap ? (v & 7) * a.tiles8 + xr : xr * 8 + (v & 7);
        if (tile >= a.ntiles) return false;
        it.cb = xslot - xr * a.ncb;
        it.img = tile / per_img;
        tile -= it.img * per_img;
        const int strip = tile / a.TPS, t = tile - strip * a.TPS;
        it.sx0 = strip * a.SW;
        it.p0 = t * S2_BM;
        it.y_first = (int)(((uint32_t)it.p0 * a.mg_sw) >> 20);
        return true;
    };
    int vb = blockIdx.x;
    const int vstep = gridDim.x;
    Item cur, nxt;
    if (!decode(vb, cur)) return;
    bool has_next = decode(vb + vstep, nxt);

    // ---- window pieces of this wave: piece n of a plane covers plane pixels 16 j .. 16 j + 15, j = wave + 8 n (a short share repeats a piece)
    const int npieces = a.plane16 >> 4;
    int pj[3];   // (the first NP entries are used: a template-sized local array makes hipcc drop the kernel's host stub)
#pragma unroll
    for (int n = 0; n < NP; ++n) pj[n] = wave + 8 * n < npieces ? wave + 8 * n : wave % npieces;
    // byte offset of this lane's 16 bytes (hi piece, channels 0-31) of piece n of plane q of an item's window; out of the image: S2D_OOB -> zeros
    auto src_offset = [&](const Item& it, int q, int n) {
        // (the lane index through an opaque move: otherwise the item-independent half of this arithmetic is hoisted out of the item loop for
        // all twelve pieces and lives in registers across it -- the kernel has exactly 256)
        int lz;
        asm volatile("v_mov_b32 %0, %1" : "=v"(lz) : "v"(lane));
        const int px = pj[n] * 16 + (lz >> 2);
        const int grpk = (lz & 3) ^ ((px >> 1) & 2);                            // LDS position (lane & 3) of pixel px holds this K group
        const int py = (int)(((uint32_t)px * a.mg_pw) >> 20), pxx = px - py * a.PW;
        const int iy = 2 * it.y_first - 1 + 2 * py + (q >> 1), ix = 2 * it.sx0 - 1 + 2 * pxx + (q & 1);
        const bool ok = py < a.PH && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        const uint32_t off = ((uint32_t)((it.img * a.H + iy) * a.W + ix) * (uint32_t)a.in_cs + (uint32_t)a.in_coff) * 4u + (uint32_t)(grpk << 5);
        const uint32_t m = 0u - (uint32_t)ok;
        return (off & m) | (S2D_OOB & ~m);
    };
    uint32_t gsrc[4][3];     // source of the NEXT issue of each plane's pieces (the first NP entries are used)
    // scalar byte offset of this wave's 16 rows of (block 2 cb + hb, half-chunk 0, tap 0)
    auto wgt_base = [&](int cb) { return (uint32_t)(((2 * cb + hb) * a.nck) * S2D_SLAB + grp * 1024); };
    auto issue_tile = [&](int tap, uint32_t wsrc, uint32_t woob) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rwg, (slds_vp)(s2d_lds + WOFF + tap * S2D_TAP + wave * 1024), 16, wlane | woob, wsrc + (uint32_t)tap * 4096u, 0, 0);
    };

    // ---- this wave's MFMA operands: LDS byte address of pixel tile j's fragment for tap t; output pixel (or OOB)
    // (the fragment address of tap t is formed at the read: plane pixel = apl[j] + the tap's step, a few VALU instructions beside the other
    // wave's MFMAs -- a table of the 36 addresses is 36 registers)
    uint32_t apl[4], po[4];
    auto set_output = [&](const Item& it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int p = it.p0 + (grp * 4 + j) * 16 + lrow;
            const int y = (int)(((uint32_t)p * a.mg_sw) >> 20), xs = p - y * a.SW;
            apl[j] = (uint32_t)((y - it.y_first) * a.PW + xs);
            const int ox = it.sx0 + xs;
            po[j] = (y < a.Ho && ox < a.Wo) ? (uint32_t)((it.img * a.Ho + y) * a.Wo + ox) : S2D_OOB;
        }
    };
    const uint32_t kg16 = (uint32_t)(kg << 4);
    auto frag_addr = [&](int j, int t) {
        const int r = t / 3, sx = t % 3;
        const uint32_t pw = apl[j] + (uint32_t)((r >> 1) * a.PW + (sx >> 1));
        return ((uint32_t)(((r & 1) * 2 + (sx & 1)) * a.plane16) + pw) * 64u + (kg16 ^ ((pw << 3) & 0x20u));
    };
    sf32x4_ acc[4][4];   // [0..1]: main, starts at the bias; [2..3]: cross, starts at zero
    // (gridDim.x is a multiple of 8 ncb: a workgroup keeps its 64-channel block, so its bias and its weight slabs are loaded / addressed
    // once -- a bias load per item is a vector-memory load whose wait drains the DMA pieces already in flight for that item)
    float4 bias4[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) bias4[i] = *reinterpret_cast<const float4*>(a.bias + cur.cb * 64 + hb * 32 + i * 16 + kg * 4);
    auto reset_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[i][j] = sf32x4_{bias4[i].x, bias4[i].y, bias4[i].z, bias4[i].w};
                acc[i + 2][j] = sf32x4_{0.f, 0.f, 0.f, 0.f};
            }
        }
    };

    // ---- issue slots.  Slot (h, g) = what the barrier of group g in half-chunk h releases: the next half-chunk's pieces of plane Q[g] and,
    // behind an L half-chunk, the next H slab's tiles of group g's taps.  Its DMA instructions are not issued in one burst behind the
    // barrier (8 waves x 4-7 KB at once: the first form of this kernel spent 11 k of an item's 54 k cycles there, and the epilogue's
    // stores queued behind the last burst) but one or two at a time between the MFMA blocks of the FOLLOWING group.
    struct Pend {
        uint32_t wsrc, woob;   // tile source (scalar part), its out-of-range flag
        bool last;             // the planes move on to the next item behind these pieces
    };
    auto issue_piece = [&](auto q_c, auto g_c, auto lo_c, auto k_c, const Pend& pd) {
        constexpr int q = decltype(q_c)::value, g = decltype(g_c)::value, k = decltype(k_c)::value;
        constexpr bool pislo = decltype(lo_c)::value;   // the slot belongs to an L half-chunk: tiles follow the window pieces
        if constexpr (k < NP) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (slds_vp)(s2d_lds + (q * a.plane16 + pj[k] * 16) * 64), 16, gsrc[q][k], 0, 0, 0);
            // behind hi pieces: the lo pieces; then the next 32 channels -- or the next item's half-chunk 0 (computed here: an array of the
            // twelve offsets across the item would be twelve more registers in a kernel that spills at 256)
            if (pd.last) gsrc[q][k] = has_next ? src_offset(nxt, q, k) : S2D_OOB;
            else gsrc[q][k] += pislo ? 16u : 112u;
        } else {
            constexpr int tl[4][4] = {{4, 4, 4, 4}, {1, 7, 7, 7}, {3, 5, 5, 5}, {0, 2, 6, 8}};
            issue_tile(tl[g][k - NP], pd.wsrc, pd.woob);
        }
    };

    // ---- prologue: slots (-1, 0..2) of the first item: planes (1,1), (0,1), (1,0) of its half-chunk 0 and the tiles of their taps; slot (-1, 3)
    // (plane (0,0), tiles 0 2 6 8) is issued under the first tap like every item's
    const uint32_t wcur = wgt_base(cur.cb);
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int n = 0; n < NP; ++n) gsrc[q][n] = src_offset(cur, q, n);
    {
        const Pend p0{wcur, 0u, false};
        auto slot = [&](auto q_c, auto g_c) {
            constexpr int g = decltype(g_c)::value;
            constexpr int nt = g == 0 ? 1 : 2;
            issue_piece(q_c, g_c, std::true_type{}, std::integral_constant<int, 0>{}, p0);
            issue_piece(q_c, g_c, std::true_type{}, std::integral_constant<int, 1>{}, p0);
            if constexpr (NP > 2) issue_piece(q_c, g_c, std::true_type{}, std::integral_constant<int, 2>{}, p0);
            issue_piece(q_c, g_c, std::true_type{}, std::integral_constant<int, NP>{}, p0);
            if constexpr (nt > 1) issue_piece(q_c, g_c, std::true_type{}, std::integral_constant<int, NP + 1>{}, p0);
        };
        slot(std::integral_constant<int, 3>{}, std::integral_constant<int, 0>{});
        slot(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
        slot(std::integral_constant<int, 2>{}, std::integral_constant<int, 2>{});
    }
    set_output(cur);
    reset_acc();
    s2d_wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    S2DP(0)

    for (;;) {
        const uint32_t wnext = wcur;                              // (the next item is of the same 64-channel block)
        const uint32_t woob_next = has_next ? 0u : S2D_OOB;

        // one half-chunk: four tap groups, each closed by (counted wait, barrier)
        auto half_chunk = [&](auto lo_c, const int ck) {
            constexpr bool islo = decltype(lo_c)::value;   // L half-chunk: only the cross tiles (i = 2, 3) accumulate, on the H slab's MAIN rows
            constexpr int I0 = islo ? 2 : 0, NI = islo ? 2 : 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(apl[j]));   // (keeps the 36 fragment addresses from being hoisted across half-chunks)
            const bool to_next_item = ck + 1 >= a.nck;     // this half-chunk's slots fetch the next item's half-chunk 0
            // slot (ck - 1, 3), issued under group 0: the tiles of THIS half-chunk's slab if the one before was an L; slots (ck, 0..2) under groups 1..3
            const Pend pprev{wcur + (uint32_t)ck * S2D_SLAB, 0u, ck - 1 == a.nck - 2};
            const Pend pthis{to_next_item ? wnext : wcur + (uint32_t)(ck + 1) * S2D_SLAB, to_next_item ? woob_next : 0u, ck == a.nck - 2};
            // tap t; between its MFMA blocks the pieces [K0, K1) of pending slot (plane PQ, group PG, L flag PL), spread over the NI blocks
            auto tap = [&](auto t_c, auto pq_c, auto pg_c, auto pl_c, auto k0_c, auto k1_c, const Pend& pd) {
                constexpr int t = decltype(t_c)::value, K0 = decltype(k0_c)::value, K1 = decltype(k1_c)::value;
                vec8 wf[4], xf[4];
#pragma unroll
                for (int i = I0; i < 4; ++i) wf[i] = *reinterpret_cast<const vec8*>(s2d_lds + wrd + t * S2D_TAP + (i - I0) * 1024);
#pragma unroll
                for (int j = 0; j < 4; ++j) xf[j] = *reinterpret_cast<const vec8*>(s2d_lds + frag_addr(j, t));
                auto block = [&](auto b_c) {
                    constexpr int b = decltype(b_c)::value;
                    if constexpr (b < NI) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[I0 + b][j] = Fp16::mfma(wf[I0 + b], xf[j], acc[I0 + b][j]);
                        constexpr int ka = K0 + (K1 - K0) * b / NI, kb = K0 + (K1 - K0) * (b + 1) / NI;
                        if constexpr (kb > ka) issue_piece(pq_c, pg_c, pl_c, std::integral_constant<int, ka>{}, pd);
                        if constexpr (kb > ka + 1) issue_piece(pq_c, pg_c, pl_c, std::integral_constant<int, ka + 1>{}, pd);
                        if constexpr (kb > ka + 2) issue_piece(pq_c, pg_c, pl_c, std::integral_constant<int, ka + 2>{}, pd);
                        static_assert(kb <= ka + 3, "at most three pieces between two MFMA blocks");
                    }
                };
                block(std::integral_constant<int, 0>{}); block(std::integral_constant<int, 1>{});
                block(std::integral_constant<int, 2>{}); block(std::integral_constant<int, 3>{});
            };
            auto close = [&](auto g_c) {
                constexpr int g = decltype(g_c)::value;
                // in flight past this point: the two most recent slots (NP window pieces each, + the tap tiles of L slots)
                constexpr int allow = 2 * NP + (islo ? (g == 0 ? 0 : (g == 1 ? 1 : (g == 2 ? 3 : 4))) : (g == 0 ? 6 : (g == 1 ? 4 : 0)));
                __builtin_amdgcn_sched_barrier(0);
                S2DP(islo ? 2 : 1)
                s2d_wait_vm<allow>();
                S2DP(3)
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                S2DP(4)
            };
            using std::integral_constant;
            typedef integral_constant<bool, !islo> PrevL;     // the half-chunk before this one was an L (or, at ck = 0, the previous item's last L)
            typedef integral_constant<bool, islo> ThisL;
            // pieces per slot: NP window pieces + (L slots) 1, 2, 2, 4 tiles
            constexpr int P3 = NP + (!islo ? 4 : 0), P0 = NP + (islo ? 1 : 0), P1 = NP + (islo ? 2 : 0), P2 = NP + (islo ? 2 : 0);
            // group 0: tap (1,1), plane 3; under it slot (ck - 1, 3): plane 0
            tap(integral_constant<int, 4>{}, integral_constant<int, 0>{}, integral_constant<int, 3>{}, PrevL{}, integral_constant<int, 0>{}, integral_constant<int, P3>{}, pprev);
            close(integral_constant<int, 0>{});
            // group 1: taps (0,1) (2,1), plane 1; under them slot (ck, 0): plane 3
            tap(integral_constant<int, 1>{}, integral_constant<int, 3>{}, integral_constant<int, 0>{}, ThisL{}, integral_constant<int, 0>{}, integral_constant<int, P0 / 2>{}, pthis);
            tap(integral_constant<int, 7>{}, integral_constant<int, 3>{}, integral_constant<int, 0>{}, ThisL{}, integral_constant<int, P0 / 2>{}, integral_constant<int, P0>{}, pthis);
            close(integral_constant<int, 1>{});
            // group 2: taps (1,0) (1,2), plane 2; slot (ck, 1): plane 1
            tap(integral_constant<int, 3>{}, integral_constant<int, 1>{}, integral_constant<int, 1>{}, ThisL{}, integral_constant<int, 0>{}, integral_constant<int, P1 / 2>{}, pthis);
            tap(integral_constant<int, 5>{}, integral_constant<int, 1>{}, integral_constant<int, 1>{}, ThisL{}, integral_constant<int, P1 / 2>{}, integral_constant<int, P1>{}, pthis);
            close(integral_constant<int, 2>{});
            // group 3: the four even-even taps, plane 0; slot (ck, 2): plane 2
            tap(integral_constant<int, 0>{}, integral_constant<int, 2>{}, integral_constant<int, 2>{}, ThisL{}, integral_constant<int, 0>{}, integral_constant<int, P2 / 4>{}, pthis);
            tap(integral_constant<int, 2>{}, integral_constant<int, 2>{}, integral_constant<int, 2>{}, ThisL{}, integral_constant<int, P2 / 4>{}, integral_constant<int, P2 / 2>{}, pthis);
            tap(integral_constant<int, 6>{}, integral_constant<int, 2>{}, integral_constant<int, 2>{}, ThisL{}, integral_constant<int, P2 / 2>{}, integral_constant<int, 3 * P2 / 4>{}, pthis);
            tap(integral_constant<int, 8>{}, integral_constant<int, 2>{}, integral_constant<int, 2>{}, ThisL{}, integral_constant<int, 3 * P2 / 4>{}, integral_constant<int, P2>{}, pthis);
            close(integral_constant<int, 3>{});
        };
        for (int ck = 0; ck < a.nck; ck += 2) {
            half_chunk(std::false_type{}, ck);
            half_chunk(std::true_type{}, ck + 1);
        }
        S2DP(5)

        // ---- epilogue: lane holds channels kg*4 .. +3 of pixel lrow of every (i, j) tile; act(main + 2^-11 cross) -> split -> 16-byte G8 stores
        {
            const uint32_t ch0 = (uint32_t)(cur.cb * 256) + ch_lane;
            // one pixel tile at a time (eight values per lane live: with all 32 the SiLU instantiations spill).  The hi halves are converted under
            // MODE.FP_DENORM[7:6] = 0: a hi below the half normal range is flushed and the value moves into lo (x3_split's rule); volatile asm
            // keeps the conversions between the two s_setreg (conv_halo8_x3.hip's epilogue)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                sf32x2_ val[2][2];
                uint32_t hw[2][2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    val[i][0] = s2d_act2<ACT>(sf32x2_{acc[i + 2][j][0], acc[i + 2][j][1]} * kX3Down + sf32x2_{acc[i][j][0], acc[i][j][1]});
                    val[i][1] = s2d_act2<ACT>(sf32x2_{acc[i + 2][j][2], acc[i + 2][j][3]} * kX3Down + sf32x2_{acc[i][j][2], acc[i][j][3]});
                }
                __builtin_amdgcn_s_setreg((unsigned short)(1 | (6 << 6) | (1 << 11)), 0u);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int h = 0; h < 2; ++h) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hw[i][h]) : "v"(val[i][h][0]), "v"(val[i][h][1]));
                __builtin_amdgcn_s_setreg((unsigned short)(1 | (6 << 6) | (1 << 11)), 3u);
                const uint32_t oo = po[j] == S2D_OOB ? S2D_OOB : (po[j] * (uint32_t)a.out_cs + (uint32_t)a.out_coff) * 4u + ch0;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const uint32_t oi = (cur.cb * 64 + hb * 32 + i * 16 + (kg >> 1) * 8 < a.cout) ? oo + i * 64 : S2D_OOB;   // (cout is a multiple of 8)
                    uint32_t lw[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const sf32x2_ d = (val[i][h] - __builtin_convertvector(__builtin_bit_cast(e_f16x2, hw[i][h]), sf32x2_)) * kX3Up;
                        lw[h] = __builtin_bit_cast(uint32_t, __builtin_convertvector(d, e_f16x2));
                    }
                    // even 16-lane rows end up with the group's 16 hi bytes, odd rows with its 16 lo bytes
                    const auto s0 = __builtin_amdgcn_permlane16_swap(hw[i][0], lw[0], false, false);
                    const auto s1 = __builtin_amdgcn_permlane16_swap(hw[i][1], lw[1], false, false);
                    __builtin_amdgcn_raw_buffer_store_b128(su32x4_{s0[0], s1[0], s0[1], s1[1]}, rout, oi, 0, 0);
                }
            }
        }
        S2DP(6)
#ifdef ADAS_S2D_PROF
        ++nitem__;
#endif
        if (!has_next) break;
        cur = nxt;
        vb += vstep;
        has_next = decode(vb + vstep, nxt);
        set_output(cur);
        reset_acc();
        S2DP(7)
    }
#ifdef ADAS_S2D_PROF
    if (tid == 0) {
        for (int i__ = 0; i__ < 8; ++i__) atomicAdd(&g_s2d_prof[i__], pacc__[i__]);
        atomicAdd(&g_s2d_prof[8], nitem__);
    }
#endif
    s2d_wait_vm<0>();   // (the tail's out-of-range pieces: nothing may still be landing in LDS when the workgroup ends)
}

#ifdef ADAS_S2D_PROF
extern "C" int adas_debug_s2d_prof(unsigned long long* out16, int reset) {
    static unsigned long long h[16];
    if (out16 && hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_s2d_prof), sizeof(h)) != hipSuccess) return -1;
    if (reset) {
        for (int i = 0; i < 16; ++i) h[i] = 0;
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_s2d_prof), h, sizeof(h)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

static bool s2x_enabled() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("ADAS_NO_HALO_S2P_X3");
        v = (e && e[0] == '1') ? 0 : 1;
    }
    return v == 1;
}

// static part (shapes): the conv also gets conv_halo8_x3's weight packing (engine.cpp)
bool halo_s2p_x3_shape_ok(int kh, int kw, int stride, int pad, int res_mode, const TView& in, const TView& out) {
    if (!s2x_enabled() || stride != 2 || kh != 3 || kw != 3 || pad != 1 || res_mode != RES_NONE) return false;
    if (in.f32 || out.f32 || out.h != (in.h + 2 - 3) / 2 + 1 || out.w != (in.w + 2 - 3) / 2 + 1) return false;
    if ((in.c & 31) || (in.cs & 7) || (in.coff & 7) || (out.c & 7) || (out.cs & 7) || (out.coff & 7)) return false;
    if ((long)in.h * in.w * in.cs * 4 >= (1L << 31)) return false;
    if (2 * out.c < (out.c + 63) / 64 * 64) return false;        // more than half of the MFMA work on padding rows
    S2Plan pl;
    return plan_s2(out.h, out.w, &pl) && pl.eff >= 0.45;
}

bool halo_s2p_x3_applicable(int kh, int kw, int stride, int pad, int res_mode, int n, const TView& in, const TView& out) {
    if (!halo_s2p_x3_shape_ok(kh, kw, stride, pad, res_mode, in, out)) return false;
    S2Plan pl;
    if (!plan_s2(out.h, out.w, &pl)) return false;
    // one 8-wave workgroup per CU: the launch has to fill a good part of the chip (ADAS_S2X_MIN_ITEMS, default 96; 256 until the end of
    // round 6: YOLOv8l's 40x40 -> 20x20 layers at 8 frames are 128 items and ran on the generic kernel at 134 TFLOP/s)
    static long min_items = -1;
    if (min_items < 0) { const char* e = getenv("ADAS_S2X_MIN_ITEMS"); min_items = e ? atol(e) : 96; if (min_items < 1) min_items = 96; }
    return (long)n * pl.NS * pl.TPS * ((out.c + 63) / 64) >= min_items;
}

// the LDS-DMA form (conv_s2d_x3_kernel); hipErrorNotSupported where it does not apply (ADAS_NO_S2D_X3=1, tensors past the 32-bit offsets)
static bool s2d_x3_fits(int n, const TView& in, const TView& out) {
    static int on = -1;
    if (on < 0) { const char* e = getenv("ADAS_NO_S2D_X3"); on = (e && e[0] == '1') ? 0 : 1; }
    const size_t in_bytes = (size_t)n * in.h * in.w * in.cs * 4, out_bytes = (size_t)n * out.h * out.w * out.cs * 4;
    const int ncb = (out.c + 63) / 64, nck = 2 * (in.c / 32);
    const size_t wgt_bytes = (size_t)2 * ncb * nck * S2D_SLAB;
    return on && ncb <= 32 && in_bytes < 0x70000000ull && out_bytes < 0x70000000ull && wgt_bytes < 0x70000000ull;
}
// which of the two stride-2 kernels of the split precision a layer runs on (adas_engine_layer_kernel's label)
bool halo_s2d_x3_applicable(int kh, int kw, int stride, int pad, int res_mode, int n, const TView& in, const TView& out) {
    return halo_s2p_x3_applicable(kh, kw, stride, pad, res_mode, n, in, out) && s2d_x3_fits(n, in, out);
}
static hipError_t launch_conv_s2d_x3(const ConvArgs& a, const S2Plan& pl, hipStream_t st) {
    if (!s2d_x3_fits(a.n, a.in, a.out)) return hipErrorNotSupported;
    const size_t in_bytes = (size_t)a.n * a.in.h * a.in.w * a.in.cs * 4, out_bytes = (size_t)a.n * a.out.h * a.out.w * a.out.cs * 4;
    const int ncb = (a.out.c + 63) / 64, nck = 2 * (a.in.c / 32);
    const size_t wgt_bytes = (size_t)2 * ncb * nck * S2D_SLAB;
    S2DDev d;
    d.in = a.in.p; d.wgt = a.wgt_h8x3; d.bias = a.bias; d.out = a.out.p;
    d.in_bytes = (uint32_t)in_bytes; d.wgt_bytes = (uint32_t)wgt_bytes; d.out_bytes = (uint32_t)out_bytes;
    d.in_cs = a.in.cs; d.in_coff = a.in.coff; d.cin = a.in.c; d.H = a.in.h; d.W = a.in.w;
    d.out_cs = a.out.cs; d.out_coff = a.out.coff; d.cout = a.out.c; d.Ho = a.out.h; d.Wo = a.out.w;
    d.nck = nck;
    d.SW = pl.SW; d.NS = pl.NS; d.TPS = pl.TPS; d.PW = pl.PW;
    d.PH = (S2_BM + pl.SW - 1) / pl.SW + ((S2_BM % pl.SW) ? 1 : 0) + 1;
    d.plane16 = (d.PH * d.PW + 15) / 16 * 16;
    d.mg_pw = pl.mg_pw; d.mg_sw = pl.mg_sw;
    d.ntiles = a.n * pl.NS * pl.TPS;
    d.tiles8 = (d.ntiles + 7) / 8;
    d.ncb = ncb;
    { static int xm = -1; if (xm < 0) { const char* e = getenv("ADAS_HALO_XMAP"); xm = e ? atoi(e) : 1; } d.xmap = xm; }
    const size_t lds = (size_t)4 * d.plane16 * 64 + 9 * S2D_TAP;
    if (lds > 160 * 1024) return hipErrorNotSupported;   // (plan_s2 keeps 4 planes <= S2_MAXPIX pixels: plane16 <= 352; its mg_pw covers plane + 8 >= plane16 pixels)
    const int nvb = 8 * d.tiles8 * d.ncb;
    // workgroups per CU over the launch (ADAS_S2D_ROUNDS, default 16: up to 4,096 workgroups, the later ones dealt as CUs fall free).  One
    // persistent workgroup per CU (= 1) is the faster launch on its own (profiles/r06/s2d_x3_phases.txt) but in the step, where the other
    // network's launches run beside it, it measured 1.3-1.6 % slower than the fine-grained launch: a 256-workgroup wall takes and releases all
    // CUs at once (profiles/r06/ab_s2d_x3.txt)
    static int rounds = -1;
    if (rounds < 0) { const char* e = getenv("ADAS_S2D_ROUNDS"); rounds = e ? atoi(e) : 16; if (rounds < 1 || rounds > 16) rounds = 16; }
    const int gmax = 8 * d.ncb * (32 / d.ncb) * rounds;      // a multiple of 8 ncb: a workgroup's 64-channel block never changes
    const dim3 grid(nvb > gmax ? gmax : nvb);
    static bool attr_done = false;
    if (!attr_done) {
#define S2D_ATTR(A_) (void)hipFuncSetAttribute((const void*)conv_s2d_x3_kernel<A_, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
                     (void)hipFuncSetAttribute((const void*)conv_s2d_x3_kernel<A_, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
        S2D_ATTR(ACT_NONE); S2D_ATTR(ACT_SILU); S2D_ATTR(ACT_RELU); S2D_ATTR(ACT_LEAKY);
#undef S2D_ATTR
        attr_done = true;
    }
    const bool np3 = d.plane16 > 256;
#define S2D_GO(A_) do { if (np3) hipLaunchKernelGGL((conv_s2d_x3_kernel<A_, 3>), grid, dim3(S2_THR), lds, st, d); \
                        else hipLaunchKernelGGL((conv_s2d_x3_kernel<A_, 2>), grid, dim3(S2_THR), lds, st, d); } while (0)
    if (a.act == ACT_SILU) S2D_GO(ACT_SILU);
    else if (a.act == ACT_RELU) S2D_GO(ACT_RELU);
    else if (a.act == ACT_LEAKY) S2D_GO(ACT_LEAKY);
    else S2D_GO(ACT_NONE);
#undef S2D_GO
    return hipGetLastError();
}

hipError_t launch_conv_s2p_x3(const ConvArgs& a, hipStream_t st) {
    S2Plan pl;
    if (!a.wgt_h8x3 || !halo_s2p_x3_applicable(a.kh, a.kw, a.stride, a.pad, a.res_mode, a.n, a.in, a.out) || !plan_s2(a.out.h, a.out.w, &pl))
        return hipErrorNotSupported;
    {
        const hipError_t e = launch_conv_s2d_x3(a, pl, st);
        if (e != hipErrorNotSupported) return e;
    }
    S2XDev d;
    d.in = (const unsigned char*)a.in.p; d.wgt = (const uint16_t*)a.wgt_h8x3; d.bias = a.bias; d.out = (x3s*)a.out.p;
    d.in_cs = a.in.cs; d.in_coff = a.in.coff; d.cin = a.in.c; d.H = a.in.h; d.W = a.in.w;
    d.out_cs = a.out.cs; d.out_coff = a.out.coff; d.cout = a.out.c;
    d.nck = 2 * (a.in.c / 32);
    d.SW = pl.SW; d.NS = pl.NS; d.TPS = pl.TPS; d.PW = pl.PW; d.plane = pl.plane;
    d.npix4 = 4 * pl.plane * 4;
    d.Ho = a.out.h; d.Wo = a.out.w;
    d.mg_pw = pl.mg_pw; d.mg_sw = pl.mg_sw; d.mg_plane = pl.mg_plane;
    d.ntiles = a.n * pl.NS * pl.TPS;
    d.tiles8 = (d.ntiles + 7) / 8;
    d.ncb = (a.out.c + 63) / 64;
    { static int xm = -1; if (xm < 0) { const char* e = getenv("ADAS_HALO_XMAP"); xm = e ? atoi(e) : 1; } d.xmap = xm; }
    const dim3 grid(8 * d.tiles8 * d.ncb);
    const size_t lds = ((size_t)4 * pl.plane * 32 + (size_t)2 * S2_WROWS * 32) * 2;
    static bool attr_done = false;
    if (!attr_done) {
#define S2X_ATTR(A_) (void)hipFuncSetAttribute((const void*)conv_s2p_x3_kernel<A_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
        S2X_ATTR(ACT_NONE); S2X_ATTR(ACT_SILU); S2X_ATTR(ACT_RELU); S2X_ATTR(ACT_LEAKY);
#undef S2X_ATTR
        attr_done = true;
    }
    if (a.act == ACT_SILU) hipLaunchKernelGGL((conv_s2p_x3_kernel<ACT_SILU>), grid, dim3(S2_THR), lds, st, d);
    else if (a.act == ACT_RELU) hipLaunchKernelGGL((conv_s2p_x3_kernel<ACT_RELU>), grid, dim3(S2_THR), lds, st, d);
    else if (a.act == ACT_LEAKY) hipLaunchKernelGGL((conv_s2p_x3_kernel<ACT_LEAKY>), grid, dim3(S2_THR), lds, st, d);
    else hipLaunchKernelGGL((conv_s2p_x3_kernel<ACT_NONE>), grid, dim3(S2_THR), lds, st, d);
    return hipGetLastError();
}

}  // namespace adas
