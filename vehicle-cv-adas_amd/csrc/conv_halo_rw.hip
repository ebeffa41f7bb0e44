// conv_halo_rw.hip -- stride-1 3x3 convolution for SHORT K (Cin <= 64): persistent workgroups with the weights
// resident in LDS.
//
// Phase profile of conv_halo at Cin = 64 (tools/experiments/halo_prof.py, 80x400x64->64, 64 frames): only ~1/4 of a
// workgroup's cycles are the tap loop; the rest is per-tile fixed cost -- 64 % of the bytes it stages are the 73 KB
// of weights, identical for every one of the 8000 tiles.  Here a workgroup
//   * loads the 9 x 64 x Cin weight slab of its output-channel tile ONCE and keeps it in LDS (<= 73.7 KB),
//   * walks spatial tiles with a grid-stride loop; a tile's window holds ALL input channels (two 32-channel planes in
//     the conv_halo layout: 64 B pixels, chunk-swizzled, conflict-free) and is double-buffered, so
//   * the only per-tile synchronisation is ONE barrier: window(t+1) is fetched into registers before the MFMAs of
//     tile t, written to the other LDS buffer after its epilogue; residual values are fetched before the MFMAs too.
// One workgroup (4 waves) per CU: 73.7 KB weights + 2 x 41.5 KB windows at Cin = 64.  The tap loop, fragment layouts,
// strip-linear tiling and the 16-byte permlane epilogue are those of conv_halo.hip; weights use the same packing
// (per-tap channel runs padded to 32), so the choice between the two kernels is a launch-time decision.
#include "kernels.h"
#include "elem16.h"
#include <stdlib.h>
#include <map>
#include <mutex>
#include <tuple>

namespace adas {

typedef __attribute__((ext_vector_type(4))) float rf32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t ru32x4;

template <int ACT>
__device__ __forceinline__ float r_act(float v) {
    if (ACT == ACT_SILU) return v * fast_rcp(1.0f + __expf(-v));
    if (ACT == ACT_RELU) return fmaxf(v, 0.0f);
    if (ACT == ACT_LEAKY) return fmaxf(v, 0.1f * v);
    return v;
}

struct RwDev {
    const uint16_t* in;
    const uint16_t* wgt;
    const float* bias;
    uint16_t* out;
    const uint16_t* res;
    int in_cs, in_coff, cin, H, W;
    int out_cs, out_coff, cout;
    int res_cs, res_coff, res_mode;
    int kpad, cin_pad;
    int SW, NS, TPS, WW, maxpix;
    int n_spatial, NT;            // spatial tiles (all frames), output-channel tiles of 64
    uint32_t mg_ww, mg_sw;
};

constexpr int RW_BM = 256;
constexpr int RW_BN = 64;
constexpr int RW_ELEMS = 11 * 256;  // 16-byte window pieces per tile that fit the register staging: nchunk * maxpix * 4 <= 2816
constexpr int RW_NW = 8;            // waves per workgroup: two per SIMD (one workgroup per CU) so LDS/global latencies of one
                                    // wave hide under the other's MFMAs; measured faster than 4 waves with TM = 4
constexpr int RW_THR = 64 * RW_NW;
constexpr int RW_NA = RW_ELEMS / RW_THR + 1;

template <typename E, int NCH, int ACT, bool HAS_RES, int BN = 64>  // NCH = 32-channel planes (1 | 2); BN = output channels per workgroup (64 | 32,
                                                       // the CONV_HALO packing of the layer: halo_bn(cout)); HAS_RES is a template flag because a runtime branch
                                          // around the residual loads makes hipcc drain vmcnt(0) at the join -- and with it the window prefetch
__global__ __launch_bounds__(RW_THR, RW_NW / 4) void conv_halo_rw_kernel(RwDev a) {
    E::enter();
    typedef typename E::vec8 rvec8;
    constexpr int TAPS = 9, TM = RW_BM / 16 / RW_NW, TN = BN / 16;
    constexpr int WROWS = TAPS * BN;                  // weight rows of 64 B per plane (576 at BN = 64)
    constexpr int NWL = (NCH * WROWS * 4 + RW_THR - 1) / RW_THR;  // one-time weight loads per thread
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    uint16_t* Ww = lds;                                  // [NCH][WROWS][32], row-swizzled
    const int plane = a.maxpix * 32;                     // elements per window plane
    uint16_t* Win0 = lds + NCH * WROWS * 32;             // [NCH][maxpix][32]
    uint16_t* Win1 = Win0 + NCH * plane;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 15, kg = lane >> 4;
    const int nt = blockIdx.x % a.NT;
    const int n0 = nt * BN;
    const int first = blockIdx.x / a.NT, step = gridDim.x / a.NT;
    const int per_img = a.NS * a.TPS;
    const int NCH_PACK = a.cin_pad >> 5;  // chunks in the packed weights (== NCH)
    if (first >= a.n_spatial) return;

    // ---- weights: once.  row = tap*64 + n; chunk kg of row r lives at position kg ^ g[(r>>2)&3], g = {0,2,3,1}
    const int gsw[4] = {0, 2, 3, 1};
    {
#pragma unroll
        for (int i = 0; i < NWL; ++i) {
            const int e = tid + RW_THR * i;  // (plane, row, piece)
            const int pl = e / (WROWS * 4), rem = e - pl * (WROWS * 4);
            const int row = rem >> 2, pc = rem & 3;
            if (pl < NCH) {
                // CONV_HALO packing (kernels.h): slab (cout tile nt, chunk pl) is WROWS contiguous 64-byte rows
                const ru32x4 v = *reinterpret_cast<const ru32x4*>(a.wgt + ((size_t)(nt * NCH_PACK + pl) * WROWS + row) * 32 + pc * 8);
                *reinterpret_cast<ru32x4*>(Ww + (pl * WROWS + row) * 32 + ((pc ^ gsw[(row >> 2) & 3]) << 3)) = v;
            }
        }
    }
    const int wrd = lrow * 32 + ((kg ^ gsw[(lrow >> 2) & 3]) << 3);
    float4 bias4[TN];
#pragma unroll
    for (int i = 0; i < TN; ++i) bias4[i] = *reinterpret_cast<const float4*>(a.bias + n0 + i * 16 + kg * 4);  // bias is padded to 128
    const bool full_n = n0 + BN <= a.cout;
    const bool wide = ((a.out_cs | a.out_coff) & 7) == 0;

    struct Geo {
        int img, sx0, p0, y_first, npix4;
    };
    auto geo_of = [&](int tile, Geo& g) {
        g.img = tile / per_img;
        tile -= g.img * per_img;
        const int strip = tile / a.TPS, t = tile - strip * a.TPS;
        g.sx0 = strip * a.SW;
        g.p0 = t * RW_BM;
        g.y_first = (int)(((uint32_t)g.p0 * a.mg_sw) >> 20);
        const int y_lastp = (int)(((uint32_t)(g.p0 + RW_BM - 1) * a.mg_sw) >> 20);
        g.npix4 = (y_lastp - g.y_first + 3) * a.WW * 4;
    };
    // window fetch of one tile into registers: slot i covers element e = tid + 256*i of [NCH][npix][4 pieces]
    ru32x4 ra[RW_NA];
    auto wload = [&](const Geo& g) {
        const uint16_t* in_img = a.in + (size_t)g.img * a.H * a.W * a.in_cs + a.in_coff;
        __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)in_img, 0, (a.H * a.W * a.in_cs - a.in_coff) * 2, 0x00020000);
        const int wy0 = g.y_first - 1, wx0 = g.sx0 - 1;
#pragma unroll
        for (int i = 0; i < RW_NA; ++i) {  // unconditional loads (unused slots are out of range -> zeros, no memory access):
            const int e = tid + RW_THR * i;  // a branch around a load makes hipcc wait vmcnt(0) at the join
            const int pl = e >= g.npix4 ? 1 : 0;
            const int e2 = e - pl * g.npix4;
            const int pix = e2 >> 2, c8 = e2 & 3;
            const int wy = (int)(((uint32_t)pix * a.mg_ww) >> 20), wx = pix - wy * a.WW;
            const int iy = wy0 + wy, ix = wx0 + wx;
            const bool ok = e < NCH * g.npix4 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            const uint32_t off = ok ? (uint32_t)(((iy * a.W + ix) * a.in_cs + pl * 32 + c8 * 8) * 2) : 0x80000000u;
            ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0);
        }
    };
    auto wstore = [&](uint16_t* Win, const Geo& g) {
        const int na = (NCH * g.npix4 + RW_THR - 1) / RW_THR;
#pragma unroll
        for (int i = 0; i < RW_NA; ++i) {
            const int e = tid + RW_THR * i;
            if (i < na && e < NCH * g.npix4) {
                const int pl = e >= g.npix4 ? 1 : 0;
                const int e2 = e - pl * g.npix4;
                *reinterpret_cast<ru32x4*>(Win + pl * plane + (e2 >> 2) * 32 + (((e2 & 3) ^ ((e2 >> 3) & 2)) << 3)) = ra[i];
            }
        }
    };

    Geo cur, nxt;
    int tile = first;
    geo_of(tile, cur);
    wload(cur);
    wstore(Win0, cur);
    __syncthreads();
    int par = 0;
    for (; tile < a.n_spatial; tile += step) {
        const int next_tile = tile + step;
        const bool has_next = next_tile < a.n_spatial;
        if (has_next) {
            geo_of(next_tile, nxt);
            wload(nxt);  // in flight under this tile's MFMAs
        }
        const uint16_t* Win = par ? Win1 : Win0;
        // ---- this tile's pixels
        int apix[TM], oy[TM], ox[TM];
        bool pok[TM];
        size_t mpix[TM];
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int p = cur.p0 + (wave * TM + j) * 16 + lrow;
            const int y = (int)(((uint32_t)p * a.mg_sw) >> 20), xs = p - y * a.SW;
            oy[j] = y;
            ox[j] = cur.sx0 + xs;
            apix[j] = (y - cur.y_first) * a.WW + xs;
            pok[j] = oy[j] < a.H && ox[j] < a.W;
            mpix[j] = pok[j] ? ((size_t)cur.img * a.H + oy[j]) * a.W + ox[j] : 0;
        }
        // residual: 16-byte loads in the wide-store layout (channel tile i + (kg&1), channels (kg>>1)*8..+7), issued here so they
        // fly under the MFMAs; the (self-inverse) permlane16 swap that hands each lane its own channel groups waits for the epilogue
        ru32x4 rwide[TM][TN / 2];
        if (HAS_RES) {
#pragma unroll
            for (int j = 0; j < TM; ++j)
#pragma unroll
                for (int i = 0; i < TN; i += 2)
                    rwide[j][i / 2] = *reinterpret_cast<const ru32x4*>(a.res + mpix[j] * a.res_cs + a.res_coff + n0 + (i + (kg & 1)) * 16 + (kg >> 1) * 8);
        }
        rf32x4 acc[TN][TM];
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j) acc[i][j] = rf32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int pl = 0; pl < NCH; ++pl) {
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) {
                const int r = tap / 3, s = tap - r * 3;
                rvec8 wf[TN], xf[TM];
#pragma unroll
                for (int i = 0; i < TN; ++i) wf[i] = *reinterpret_cast<const rvec8*>(Ww + (pl * WROWS + tap * BN + i * 16) * 32 + wrd);
#pragma unroll
                for (int j = 0; j < TM; ++j) {
                    const int pw = apix[j] + r * a.WW + s;
                    xf[j] = *reinterpret_cast<const rvec8*>(Win + pl * plane + pw * 32 + ((kg ^ ((pw >> 1) & 2)) << 3));
                }
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int j = 0; j < TM; ++j)
                        acc[i][j] = E::mfma(wf[i], xf[j], acc[i][j]);
            }
        }
        uint2 rq[TM][TN];
        if (HAS_RES) {
#pragma unroll
            for (int j = 0; j < TM; ++j)
#pragma unroll
                for (int i = 0; i < TN; i += 2) {
                    const ru32x4 w = rwide[j][i / 2];
                    const auto s0 = __builtin_amdgcn_permlane16_swap(w[0], w[2], false, false);
                    const auto s1 = __builtin_amdgcn_permlane16_swap(w[1], w[3], false, false);
                    rq[j][i] = make_uint2(s0[0], s1[0]);
                    rq[j][i + 1] = make_uint2(s0[1], s1[1]);
                }
        }
        // ---- epilogue (conv_halo.hip): 16-byte stores after a v_permlane16_swap between channel tiles i and i+1
        auto finish = [&](int i, int j, float v[4]) {
            v[0] = acc[i][j][0] + bias4[i].x; v[1] = acc[i][j][1] + bias4[i].y; v[2] = acc[i][j][2] + bias4[i].z; v[3] = acc[i][j][3] + bias4[i].w;
            if (HAS_RES) {
                const uint2 q = rq[j][i];
                const float rv[4] = {E::lo(q.x), E::hi(q.x), E::lo(q.y), E::hi(q.y)};
                if (a.res_mode == RES_BEFORE_ACT) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = r_act<ACT>(v[k] + rv[k]);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = r_act<ACT>(v[k]) + rv[k];
                }
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = r_act<ACT>(v[k]);
            }
        };
#pragma unroll
        for (int j = 0; j < TM; ++j) {
#pragma unroll
            for (int i = 0; i < TN; i += 2) {
                float vx[4], vy[4];
                finish(i, j, vx);
                finish(i + 1, j, vy);
                const uint32_t x0 = E::pack2(vx[0], vx[1]), x1 = E::pack2(vx[2], vx[3]);
                const uint32_t y0 = E::pack2(vy[0], vy[1]), y1 = E::pack2(vy[2], vy[3]);
                if (wide) {
                    const auto s0 = __builtin_amdgcn_permlane16_swap(x0, y0, false, false);
                    const auto s1 = __builtin_amdgcn_permlane16_swap(x1, y1, false, false);
                    const int c = n0 + (i + (kg & 1)) * 16 + (kg >> 1) * 8;
                    uint16_t* op = a.out + mpix[j] * a.out_cs + a.out_coff + c;
                    if (pok[j]) {
                        if (full_n || c + 8 <= a.cout) *reinterpret_cast<ru32x4*>(op) = ru32x4{s0[0], s1[0], s0[1], s1[1]};
                        else if (c + 4 <= a.cout) *reinterpret_cast<uint2*>(op) = make_uint2(s0[0], s1[0]);
                    }
                } else {
                    uint16_t* op = a.out + mpix[j] * a.out_cs + a.out_coff + n0 + kg * 4;
                    if (pok[j] && (full_n || n0 + i * 16 + kg * 4 < a.cout)) *reinterpret_cast<uint2*>(op + i * 16) = make_uint2(x0, x1);
                    if (pok[j] && (full_n || n0 + (i + 1) * 16 + kg * 4 < a.cout)) *reinterpret_cast<uint2*>(op + (i + 1) * 16) = make_uint2(y0, y1);
                }
            }
        }
        if (has_next) {
            wstore(par ? Win0 : Win1, nxt);  // the other buffer: last read one barrier ago
            cur = nxt;
        }
        __syncthreads();
        par ^= 1;
    }
}

// -------------------------------------------------------------------------------------
struct RwPlan {
    int SW, NS, TPS, WW, maxpix;
    double eff;
    uint32_t mg_ww, mg_sw;
};

static bool rw_magic_ok(int d, int nmax, uint32_t* magic) {
    uint32_t m = ((1u << 20) + d - 1) / d;
    if ((uint64_t)nmax * m >= (1ull << 32)) return false;
    for (int n = 0; n < nmax; ++n)
        if ((int)(((uint32_t)n * m) >> 20) != n / d) return false;
    *magic = m;
    return true;
}

static int rw_pix_cap(int nch, int bn, int per_cu) {  // window pixels per buffer that leave room for the resident weights
    const int lds = 160 * 1024 / per_cu - nch * 9 * bn * 64;
    int cap = lds / (2 * nch * 64);
    const int reg = RW_ELEMS / (4 * nch);  // register staging slots
    return cap < reg ? cap : reg;
}

static bool plan_rw_uncached(int H, int W, int nch, int bn, int per_cu, RwPlan* best) {
    const int cap = rw_pix_cap(nch, bn, per_cu);
    int cand[6] = {16, 32, 64, 128, 256, W};
    bool found = false;
    for (int k = 0; k < 6; ++k) {
        int SW = cand[k];
        if (SW > W && k != 5) continue;
        if (k == 5 && (W == 16 || W == 32 || W == 64 || W == 128 || W == 256)) continue;
        int rows = (RW_BM + SW - 1) / SW + ((RW_BM % SW) ? 1 : 0);
        int WW = SW + 2;
        int maxpix = (rows + 2) * WW;
        if (maxpix > cap) continue;
        int NS = (W + SW - 1) / SW;
        int TPS = (H * SW + RW_BM - 1) / RW_BM;
        double eff = (double)H * W / ((double)NS * TPS * RW_BM);
        uint32_t mw, ms;
        if (!rw_magic_ok(WW, maxpix + 64, &mw) || !rw_magic_ok(SW, TPS * RW_BM + RW_BM, &ms)) continue;
        if (!found || eff > best->eff + 1e-9 || (eff > best->eff - 1e-9 && SW > best->SW)) {
            *best = RwPlan{SW, NS, TPS, WW, maxpix, eff, mw, ms};
            found = true;
        }
    }
    return found;
}

// per_cu (out): workgroups per CU the plan leaves LDS for.  The 32-output-channel variant needs <= 128 VGPRs, so two of its
// 8-wave workgroups fit a CU when each stays within 80 KB: these layers (Cin, Cout <= 32) are bound by bytes in flight --
// one 22 KB window per CU at a time -- and a second resident workgroup doubles them.
static bool plan_rw(int H, int W, int nch, int bn, RwPlan* out, int* per_cu) {
    static std::mutex mu;
    static std::map<std::tuple<int, int, int, int>, std::tuple<bool, RwPlan, int>> cache;
    std::lock_guard<std::mutex> lk(mu);
    auto key = std::make_tuple(H, W, nch, bn);
    auto it = cache.find(key);
    if (it == cache.end()) {
        RwPlan p1{}, p2{};
        const bool ok1 = plan_rw_uncached(H, W, nch, bn, 1, &p1);
        const bool ok2 = bn == 32 && plan_rw_uncached(H, W, nch, bn, 2, &p2);
        const bool two = ok2 && (!ok1 || p2.eff >= p1.eff - 0.05);
        it = cache.emplace(key, std::make_tuple(two || ok1, two ? p2 : p1, two ? 2 : 1)).first;
    }
    *out = std::get<1>(it->second);
    if (per_cu) *per_cu = std::get<2>(it->second);
    return std::get<0>(it->second);
}

static bool rw_enabled() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("ADAS_NO_HALO_RW");
        v = (e && e[0] == '1') ? 0 : 1;
    }
    return v == 1;
}

// Launch-time choice on static shapes (same weight packing as conv_halo).  n = frames in this launch.
bool halo_rw_applicable(int kh, int kw, int stride, int pad, int n, const TView& in, const TView& out) {
    if (!rw_enabled()) return false;
    if (stride != 1 || kh != 3 || kw != 3 || pad != 1) return false;
    if (in.f32 || out.f32 || out.h != in.h || out.w != in.w) return false;
    if ((in.c & 7) || (in.cs & 7) || (in.coff & 7) || (out.c & 3) || (out.cs & 3) || (out.coff & 3)) return false;
    if (in.c < 16 || in.c > 64 || out.c <= 16 || halo_bn(out.c) == 48) return false;  // BN = 16 / 48 packings stay on conv_halo
    if ((long)in.h * in.w * in.cs >= (1L << 30)) return false;
    RwPlan pl;
    const int nch = (in.c + 31) / 32;
    const int bn = out.c <= 32 ? 32 : RW_BN;
    if (!plan_rw(in.h, in.w, nch, bn, &pl, nullptr) || pl.eff < 0.6) return false;
    // persistence pays only when every workgroup sees several tiles
    const long tiles = (long)n * pl.NS * pl.TPS * ((out.c + bn - 1) / bn);
    return tiles >= 4 * 256;
}

template <typename E, int NCH, bool HAS_RES, int BN>
static hipError_t rw_launch(const RwDev& d, int act, int grid, size_t lds, hipStream_t st) {
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)conv_halo_rw_kernel<E, NCH, ACT_NONE, HAS_RES, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv_halo_rw_kernel<E, NCH, ACT_SILU, HAS_RES, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv_halo_rw_kernel<E, NCH, ACT_RELU, HAS_RES, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)conv_halo_rw_kernel<E, NCH, ACT_LEAKY, HAS_RES, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    if (act == ACT_SILU) hipLaunchKernelGGL((conv_halo_rw_kernel<E, NCH, ACT_SILU, HAS_RES, BN>), dim3(grid), dim3(RW_THR), lds, st, d);
    else if (act == ACT_RELU) hipLaunchKernelGGL((conv_halo_rw_kernel<E, NCH, ACT_RELU, HAS_RES, BN>), dim3(grid), dim3(RW_THR), lds, st, d);
    else if (act == ACT_LEAKY) hipLaunchKernelGGL((conv_halo_rw_kernel<E, NCH, ACT_LEAKY, HAS_RES, BN>), dim3(grid), dim3(RW_THR), lds, st, d);
    else hipLaunchKernelGGL((conv_halo_rw_kernel<E, NCH, ACT_NONE, HAS_RES, BN>), dim3(grid), dim3(RW_THR), lds, st, d);
    return hipGetLastError();
}

hipError_t launch_conv_halo_rw(const ConvArgs& a, hipStream_t st) {
    RwPlan pl;
    const int nch = (a.in.c + 31) / 32;
    int per_cu = 1;
    if (!halo_rw_applicable(a.kh, a.kw, a.stride, a.pad, a.n, a.in, a.out) ||
        !plan_rw(a.in.h, a.in.w, nch, a.out.c <= 32 ? 32 : RW_BN, &pl, &per_cu))
        return hipErrorNotSupported;
    RwDev d;
    d.in = (const uint16_t*)a.in.p; d.wgt = (const uint16_t*)a.wgt; d.bias = a.bias; d.out = (uint16_t*)a.out.p;
    d.res = (const uint16_t*)a.res.p;
    d.in_cs = a.in.cs; d.in_coff = a.in.coff; d.cin = a.in.c; d.H = a.in.h; d.W = a.in.w;
    d.out_cs = a.out.cs; d.out_coff = a.out.coff; d.cout = a.out.c;
    d.res_cs = a.res.cs; d.res_coff = a.res.coff; d.res_mode = a.res_mode;
    d.kpad = a.kpad; d.cin_pad = nch * 32;
    d.SW = pl.SW; d.NS = pl.NS; d.TPS = pl.TPS; d.WW = pl.WW; d.maxpix = pl.maxpix;
    d.n_spatial = a.n * pl.NS * pl.TPS;
    const int bn = a.out.c <= 32 ? 32 : RW_BN;  // == halo_bn(cout): the packing the weights were given
    d.NT = (a.out.c + bn - 1) / bn;
    d.mg_ww = pl.mg_ww; d.mg_sw = pl.mg_sw;
    int grid = 8 * persist_slots(1) * per_cu / d.NT * d.NT;  // one (or two, see plan_rw) workgroups per CU, a multiple of the channel tiles
    const size_t lds = ((size_t)nch * 9 * bn * 32 + (size_t)2 * nch * pl.maxpix * 32) * 2;
    const bool res = a.res_mode != RES_NONE;
    if (res && (((a.res.cs | a.res.coff) & 7) != 0)) return hipErrorNotSupported;  // halo_rw_applicable() keeps such layers on conv_halo
    ADAS_DISPATCH_E16(a.prec == PREC_FP16, E, {
        if (bn == 32) {
            if (nch == 1) return res ? rw_launch<E, 1, true, 32>(d, a.act, grid, lds, st) : rw_launch<E, 1, false, 32>(d, a.act, grid, lds, st);
            return res ? rw_launch<E, 2, true, 32>(d, a.act, grid, lds, st) : rw_launch<E, 2, false, 32>(d, a.act, grid, lds, st);
        }
        if (nch == 1) return res ? rw_launch<E, 1, true, 64>(d, a.act, grid, lds, st) : rw_launch<E, 1, false, 64>(d, a.act, grid, lds, st);
        return res ? rw_launch<E, 2, true, 64>(d, a.act, grid, lds, st) : rw_launch<E, 2, false, 64>(d, a.act, grid, lds, st);
    });
    return hipErrorInvalidValue;
}

}  // namespace adas
