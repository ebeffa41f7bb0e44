// conv_halo2.hip -- second-generation stride-1 3x3 convolution: the conv_halo tiling (256 output pixels of a
// vertical strip x 64 output channels per 4-wave workgroup, input window + 9 weight slabs in LDS, all taps read
// shifted windows) rebuilt around CDNA4 features so that the matrix pipe is not left waiting:
//
//   * LDS-DMA: window and weight chunks go HBM/L2 -> LDS with `buffer_load_dwordx4 ... lds` (lane i lands at
//     M0 + 16*i; out-of-range offsets write zeros = free zero padding).  No staging VGPRs (v1 holds 76), no
//     ds_write instructions, no second barrier.
//   * Double-buffered LDS: 2 x (40 KB window + 36 KB weights) = 152 KB of the CU's 160 KB, one workgroup per CU, one
//     wave per SIMD (the regime of the guide's attention kernels).  The DMA of the next stage is in flight under the
//     MFMAs of the current one; ONE barrier per 32-channel chunk.
//   * Persistent workgroups: work items (spatial tile x feature tile) are walked with a grid-stride loop and the DMA of
//     the NEXT item's first chunk is issued under the current item's last chunk, so prologue latency is paid once per
//     workgroup, not once per tile (at Cin = 64 a tile is only two chunks).
//   * v_mfma_f32_32x32x16_bf16: a wave's 64 px x 64 cout tile is 8 MFMAs (32 cycles each) and 8 ds_read_b128 per tap.
//
// LDS layout: a window pixel / weight row is 64 B = four 16-byte pieces (piece = 2*kk + kh: K step kk, lane half kh);
// piece q of row p is stored at position q ^ ((p >> 2) & 3).  For ds_read_b128's four 16-lane service groups this is
// conflict-free at every alignment of 32 consecutive rows (exhaustive check; un-swizzled is conflicted).  The swizzle is
// applied on the SOURCE side of the DMA (lane -> which global 16 B it fetches), since the LDS side of a DMA is fixed.
#include "kernels.h"
#include <map>
#include <mutex>
#include <tuple>

namespace adas {

typedef __attribute__((ext_vector_type(8))) __bf16 gbf16x8;
typedef __attribute__((ext_vector_type(16))) float gf32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t gu32x4;
typedef __attribute__((ext_vector_type(2))) float gf32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 gbf16x2;

__device__ __forceinline__ uint32_t g_pack2(float a, float b) {
    gbf16x2 r = __builtin_convertvector(gf32x2{a, b}, gbf16x2);
    return __builtin_bit_cast(uint32_t, r);
}
template <int ACT>
__device__ __forceinline__ float g_act(float v) {
    if (ACT == ACT_SILU) return v * __frcp_rn(1.0f + __expf(-v));
    if (ACT == ACT_RELU) return fmaxf(v, 0.0f);
    return v;
}

struct Halo2Dev {
    const uint16_t* in;
    const uint16_t* wgt;
    const float* bias;
    void* out;
    const uint16_t* res;
    int in_cs, in_coff, cin, H, W;
    int out_cs, out_coff, cout;
    int res_cs, res_coff, res_mode;
    int kpad, cin_pad, wrows;     // packed weight geometry (rows = cout_pad)
    int SW, NS, TPS, WW, maxpix;  // strip width, strips per row, tiles per strip, window width, LDS pixels
    int out_f32;
    int n_spatial, n_items;       // work items = spatial tiles x feature tiles (feature-major)
    uint32_t mg_ww, mg_sw;
};

constexpr int H2_BM = 256;
constexpr int H2_CK = 32;       // channels per chunk = two MFMA K steps
constexpr int H2_MAXPIX = 640;  // window pixels (40 KB per buffer)
constexpr int H2_WI = H2_MAXPIX / 16 / 4;  // window DMA instructions per wave per chunk (10): one moves 16 pixels x 64 B

// The LDS-DMA builtin must live in a plain __device__ function: used directly inside a __global__ template, clang
// (ROCm 7.2) silently drops the host-side launch stub of every instantiation.
__device__ __forceinline__ void h2_dma16(__amdgpu_buffer_rsrc_t rsrc, uint16_t* lds_dst, uint32_t byte_off) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_dst, 16, byte_off, 0, 0, 0);
}

template <int BN, int ACT>
__global__ __launch_bounds__(256, 1) void conv_halo2_kernel(Halo2Dev a) {
    constexpr int TAPS = 9;
    constexpr int TM = 2, TN = BN / 32;
    constexpr int WROWS = TAPS * BN;            // weight rows per chunk (64 B each)
    constexpr int WQ = WROWS / 16;              // weight DMA instructions per chunk (36 | 18)
    constexpr int WWI = (WQ + 3) / 4;           // per wave
    constexpr int WIN_ELEMS = H2_MAXPIX * 32;   // uint16 elements per window buffer
    constexpr int BUF_ELEMS = WIN_ELEMS + WROWS * 32;
    __shared__ __attribute__((aligned(1024))) uint16_t buf0[BUF_ELEMS];
    __shared__ __attribute__((aligned(1024))) uint16_t buf1[BUF_ELEMS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lcol = lane & 31, kh = lane >> 5;
    const int per_img = a.NS * a.TPS;
    const int nchunk = (a.cin + H2_CK - 1) / H2_CK;
    __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.wgt, 0, a.wrows * a.kpad * 2, 0x00020000);

    // ---- DMA descriptors of one work item.  Window instruction q = i*4 + wave fills window pixels [16q, 16q+16):
    // lane -> pixel 16q + (lane>>2), LDS piece lane&3 <- source piece (lane&3) ^ ((pixel>>2)&3).
    struct Item {
        int img, sx0, p0, y_first, n0, nwin_q;
        __amdgpu_buffer_rsrc_t rs_in;
        uint32_t goff[H2_WI];
        uint32_t woff[WWI];
    };
    auto setup = [&](int item, Item& it) {
        const int nt = item / a.n_spatial;
        int tile = item - nt * a.n_spatial;
        it.n0 = nt * BN;
        it.img = tile / per_img;
        tile -= it.img * per_img;
        const int strip = tile / a.TPS, t = tile - strip * a.TPS;
        it.sx0 = strip * a.SW;
        it.p0 = t * H2_BM;
        it.y_first = (int)(((uint32_t)it.p0 * a.mg_sw) >> 20);
        const int y_lastp = (int)(((uint32_t)(it.p0 + H2_BM - 1) * a.mg_sw) >> 20);
        const int WH = y_lastp - it.y_first + 3;
        const int wy0 = it.y_first - 1, wx0 = it.sx0 - 1;
        const int npix = WH * a.WW;
        it.nwin_q = (npix + 15) >> 4;
        const uint16_t* in_img = a.in + (size_t)it.img * a.H * a.W * a.in_cs + a.in_coff;
        it.rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)in_img, 0, (a.H * a.W * a.in_cs - a.in_coff) * 2, 0x00020000);
#pragma unroll
        for (int i = 0; i < H2_WI; ++i) {
            const int pix = (i * 4 + wave) * 16 + (lane >> 2);
            const int wy = (int)(((uint32_t)pix * a.mg_ww) >> 20), wx = pix - wy * a.WW;
            const int iy = wy0 + wy, ix = wx0 + wx;
            const bool ok = pix < npix && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            const int piece = (lane & 3) ^ ((pix >> 2) & 3);
            it.goff[i] = ok ? (uint32_t)(((iy * a.W + ix) * a.in_cs) * 2 + piece * 16) : 0x80000000u;
        }
#pragma unroll
        for (int i = 0; i < WWI; ++i) {
            const int row = (i * 4 + wave) * 16 + (lane >> 2);  // tap*BN + n
            const int tap = row / BN, n = row - tap * BN;
            const int piece = (lane & 3) ^ ((row >> 2) & 3);
            it.woff[i] = (uint32_t)(((it.n0 + n) * a.kpad + tap * a.cin_pad) * 2 + piece * 16);
        }
    };
    auto dma = [&](uint16_t* buf, const Item& it, int c0) {
#pragma unroll
        for (int i = 0; i < H2_WI; ++i) {
            const int q = i * 4 + wave;
            if (q < it.nwin_q) h2_dma16(it.rs_in, buf + q * 512, it.goff[i] + (uint32_t)c0 * 2u);
        }
#pragma unroll
        for (int i = 0; i < WWI; ++i) {
            const int q = i * 4 + wave;
            if (q < WQ) h2_dma16(rs_w, buf + WIN_ELEMS + q * 512, it.woff[i] + (uint32_t)c0 * 2u);
        }
    };

    gf32x16 acc[TN][TM];
    int apix[TM];
    // A-fragment offset inside a 32-row weight block (blocks start at multiples of 32 rows): piece kk*2+kh of row lcol
    int wrd[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) wrd[kk] = lcol * 32 + (((kk * 2 + kh) ^ ((lcol >> 2) & 3)) << 3);

    auto compute = [&](const uint16_t* buf) {
        const uint16_t* Aw = buf;
        const uint16_t* Ww = buf + WIN_ELEMS;
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int r = tap / 3, s = tap - r * 3;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                gbf16x8 wf[TN], xf[TM];
#pragma unroll
                for (int i = 0; i < TN; ++i) wf[i] = *reinterpret_cast<const gbf16x8*>(Ww + (tap * BN + i * 32) * 32 + wrd[kk]);
#pragma unroll
                for (int j = 0; j < TM; ++j) {
                    const int pw = apix[j] + r * a.WW + s;
                    xf[j] = *reinterpret_cast<const gbf16x8*>(Aw + pw * 32 + (((kk * 2 + kh) ^ ((pw >> 2) & 3)) << 3));
                }
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int j = 0; j < TM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
            }
        }
    };

    // ---- persistent loop over work items; the DMA of the next stage (next chunk, or chunk 0 of the next item) is always
    // issued before the MFMAs of the current one.  Stage parity selects the LDS buffer (two distinct static arrays so the
    // compiler can tell the DMA target from the buffer being read).
    Item cur, nxt;
    int item = blockIdx.x;
    if (item >= a.n_items) return;
    setup(item, cur);
    dma(buf0, cur, 0);
    int par = 0;
    for (; item < a.n_items; item += gridDim.x) {
        const int next_item = item + gridDim.x;
        const bool has_next = next_item < a.n_items;
        int oy[TM], ox[TM];
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int p = cur.p0 + (wave * TM + j) * 32 + lcol;
            const int y = (int)(((uint32_t)p * a.mg_sw) >> 20), xs = p - y * a.SW;
            oy[j] = y;
            ox[j] = cur.sx0 + xs;
            apix[j] = (y - cur.y_first) * a.WW + xs;
        }
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        const int n0 = cur.n0, img = cur.img;

        for (int cc = 0; cc < nchunk; ++cc) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();  // this stage's data has landed for every wave; everyone is done reading the other buffer
            const bool last = cc + 1 == nchunk;
            if (last && has_next) setup(next_item, nxt);
            if (par == 0) {
                if (!last) dma(buf1, cur, (cc + 1) * H2_CK);
                else if (has_next) dma(buf1, nxt, 0);
                compute(buf0);
            } else {
                if (!last) dma(buf0, cur, (cc + 1) * H2_CK);
                else if (has_next) dma(buf0, nxt, 0);
                compute(buf1);
            }
            par ^= 1;
        }

        // ---- epilogue: for its pixel (lcol) the lane holds channels n0 + i*32 + 8*g + 4*kh + {0..3}, g = 0..3 (reg = 4*g + k)
        const bool full_n = n0 + BN <= a.cout;
        bool pok[TM];
        size_t mpix[TM];
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            pok[j] = oy[j] < a.H && ox[j] < a.W;
            mpix[j] = pok[j] ? ((size_t)img * a.H + oy[j]) * a.W + ox[j] : 0;
        }
        float4 bias4[TN][4];
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) bias4[i][g] = *reinterpret_cast<const float4*>(a.bias + n0 + i * 32 + g * 8 + kh * 4);  // bias is padded to 128
        uint2 rq[TM][TN][4];
        if (a.res_mode != RES_NONE) {
#pragma unroll
            for (int j = 0; j < TM; ++j)
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int g = 0; g < 4; ++g)
                        rq[j][i][g] = *reinterpret_cast<const uint2*>(a.res + mpix[j] * a.res_cs + a.res_coff + n0 + i * 32 + g * 8 + kh * 4);
        }
#pragma unroll
        for (int i = 0; i < TN; ++i) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = n0 + i * 32 + g * 8 + kh * 4;
                const float4 b4 = bias4[i][g];
                const bool cok = full_n || c < a.cout;
#pragma unroll
                for (int j = 0; j < TM; ++j) {
                    float v[4] = {acc[i][j][g * 4 + 0] + b4.x, acc[i][j][g * 4 + 1] + b4.y, acc[i][j][g * 4 + 2] + b4.z, acc[i][j][g * 4 + 3] + b4.w};
                    if (a.res_mode != RES_NONE) {
                        const uint2 q = rq[j][i][g];
                        const float rv[4] = {__uint_as_float(q.x << 16), __uint_as_float(q.x & 0xffff0000u),
                                             __uint_as_float(q.y << 16), __uint_as_float(q.y & 0xffff0000u)};
                        if (a.res_mode == RES_BEFORE_ACT) {
#pragma unroll
                            for (int k = 0; k < 4; ++k) v[k] = g_act<ACT>(v[k] + rv[k]);
                        } else {
#pragma unroll
                            for (int k = 0; k < 4; ++k) v[k] = g_act<ACT>(v[k]) + rv[k];
                        }
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k) v[k] = g_act<ACT>(v[k]);
                    }
                    const size_t ob = mpix[j] * a.out_cs + a.out_coff + c;
                    if (pok[j] && cok) {
                        if (a.out_f32) {
                            *reinterpret_cast<float4*>((float*)a.out + ob) = make_float4(v[0], v[1], v[2], v[3]);
                        } else {
                            uint2 q;
                            q.x = g_pack2(v[0], v[1]);
                            q.y = g_pack2(v[2], v[3]);
                            *reinterpret_cast<uint2*>((uint16_t*)a.out + ob) = q;
                        }
                    }
                }
            }
        }
        if (has_next) cur = nxt;
    }
}

// -------------------------------------------------------------------------------------
struct Halo2Plan {
    int SW, NS, TPS, WW, maxpix;
    double eff;
    uint32_t mg_ww, mg_sw;
};

static bool h2_magic_ok(int d, int nmax, uint32_t* magic) {
    uint32_t m = ((1u << 20) + d - 1) / d;
    if ((uint64_t)nmax * m >= (1ull << 32)) return false;
    for (int n = 0; n < nmax; ++n)
        if ((int)(((uint32_t)n * m) >> 20) != n / d) return false;
    *magic = m;
    return true;
}

static bool plan_halo2_uncached(int H, int W, Halo2Plan* best) {
    int cand[6] = {32, 64, 128, 256, 16, W};
    bool found = false;
    double best_score = 0.0;
    for (int k = 0; k < 6; ++k) {
        int SW = cand[k];
        if (SW > W && k != 5) continue;
        if (k == 5 && (W == 16 || W == 32 || W == 64 || W == 128 || W == 256)) continue;
        int rows = (H2_BM + SW - 1) / SW + ((H2_BM % SW) ? 1 : 0);
        int WW = SW + 2;
        int maxpix = (rows + 2) * WW;
        if (maxpix > H2_MAXPIX) continue;
        int NS = (W + SW - 1) / SW;
        int TPS = (H * SW + H2_BM - 1) / H2_BM;
        double eff = (double)H * W / ((double)NS * TPS * H2_BM);
        uint32_t mw, ms;
        if (!h2_magic_ok(WW, H2_MAXPIX + 64, &mw) || !h2_magic_ok(SW, TPS * H2_BM + H2_BM, &ms)) continue;
        // 32-pixel MFMA columns: strips that are multiples of 32 keep a fragment's pixels on one window row (the layout
        // the LDS swizzle is conflict-free for); another width must buy > 6 % more useful pixels to be preferred.
        const bool m32 = (SW % 32) == 0;
        const double score = eff + (m32 ? 0.06 : 0.0);
        if (!found || score > best_score + 1e-9) {
            *best = Halo2Plan{SW, NS, TPS, WW, maxpix, eff, mw, ms};
            best_score = score;
            found = true;
        }
    }
    return found;
}

static bool plan_halo2(int H, int W, Halo2Plan* out) {
    static std::mutex mu;
    static std::map<std::pair<int, int>, std::pair<bool, Halo2Plan>> cache;
    std::lock_guard<std::mutex> lk(mu);
    auto key = std::make_pair(H, W);
    auto it = cache.find(key);
    if (it == cache.end()) {
        Halo2Plan p{};
        bool ok = plan_halo2_uncached(H, W, &p);
        it = cache.emplace(key, std::make_pair(ok, p)).first;
    }
    *out = it->second.second;
    return it->second.first;
}

static bool halo2_enabled() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("ADAS_NO_HALO2");
        v = (e && e[0] == '1') ? 0 : 1;
    }
    return v == 1;
}

// Same weight packing as conv_halo (per-tap channel runs padded to 32), so the choice between the two is a pure
// launch-time decision on static shapes.
bool halo2_applicable(int kh, int kw, int stride, int pad, const TView& in, const TView& out) {
    if (!halo2_enabled()) return false;
    if (stride != 1 || kh != 3 || kw != 3 || pad != 1) return false;
    if (in.f32 || out.h != in.h || out.w != in.w) return false;
    if ((in.c & 7) || (in.cs & 7) || (in.coff & 7) || (out.c & 3) || (out.cs & 3) || (out.coff & 3)) return false;
    if (in.c < 32 || out.c <= 16) return false;
    if ((long)in.h * in.w * in.cs >= (1L << 30)) return false;
    Halo2Plan pl;
    return plan_halo2(in.h, in.w, &pl) && pl.eff >= 0.6;
}

template <int BN>
static hipError_t launch2_bn(const Halo2Dev& d, int act, dim3 grid, hipStream_t st) {
    if (act == ACT_SILU) hipLaunchKernelGGL((conv_halo2_kernel<BN, ACT_SILU>), grid, dim3(256), 0, st, d);
    else if (act == ACT_RELU) hipLaunchKernelGGL((conv_halo2_kernel<BN, ACT_RELU>), grid, dim3(256), 0, st, d);
    else hipLaunchKernelGGL((conv_halo2_kernel<BN, ACT_NONE>), grid, dim3(256), 0, st, d);
    return hipGetLastError();
}

hipError_t launch_conv_halo2(const ConvArgs& a, hipStream_t st) {
    Halo2Plan pl;
    if (!halo2_applicable(a.kh, a.kw, a.stride, a.pad, a.in, a.out) || !plan_halo2(a.in.h, a.in.w, &pl)) return hipErrorNotSupported;
    Halo2Dev d;
    d.in = (const uint16_t*)a.in.p; d.wgt = (const uint16_t*)a.wgt; d.bias = a.bias; d.out = a.out.p;
    d.res = (const uint16_t*)a.res.p;
    d.in_cs = a.in.cs; d.in_coff = a.in.coff; d.cin = a.in.c; d.H = a.in.h; d.W = a.in.w;
    d.out_cs = a.out.cs; d.out_coff = a.out.coff; d.cout = a.out.c;
    d.res_cs = a.res.cs; d.res_coff = a.res.coff; d.res_mode = a.res_mode;
    d.kpad = a.kpad; d.cin_pad = (a.in.c + 31) / 32 * 32; d.wrows = (a.out.c + 127) / 128 * 128;
    d.SW = pl.SW; d.NS = pl.NS; d.TPS = pl.TPS; d.WW = pl.WW; d.maxpix = pl.maxpix;
    d.out_f32 = a.out.f32;
    d.mg_ww = pl.mg_ww;
    d.mg_sw = pl.mg_sw;
    const int bn = a.out.c <= 32 ? 32 : 64;
    d.n_spatial = a.n * pl.NS * pl.TPS;
    d.n_items = d.n_spatial * ((a.out.c + bn - 1) / bn);
    // one persistent workgroup per CU (152 KB of LDS each); fewer when there is less work than CUs
    dim3 grid(d.n_items < 256 ? d.n_items : 256);
    if (bn == 64) return launch2_bn<64>(d, a.act, grid, st);
    return launch2_bn<32>(d, a.act, grid, st);
}

}  // namespace adas
