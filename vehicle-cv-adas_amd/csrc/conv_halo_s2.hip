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
    return (long)n * pl.NS * pl.TPS * ((out.c + 63) / 64) >= 256;   // one 8-wave workgroup per CU: the launch has to fill the chip
}

hipError_t launch_conv_s2p_x3(const ConvArgs& a, hipStream_t st) {
    S2Plan pl;
    if (!a.wgt_h8x3 || !halo_s2p_x3_applicable(a.kh, a.kw, a.stride, a.pad, a.res_mode, a.n, a.in, a.out) || !plan_s2(a.out.h, a.out.w, &pl))
        return hipErrorNotSupported;
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
