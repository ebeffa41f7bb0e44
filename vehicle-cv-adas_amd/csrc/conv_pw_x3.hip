// conv_pw_x3.hip -- the split precision (ADAS_PREC_FP16X3) on the two streaming kernels: the pointwise (1x1, stride 1 / 2)
// convolution of conv_pw.hip and the Linear layers of conv_fc.hip.
//
// conv_x3.hip (the generic kernel) stages both operands through LDS and walks K in the (tap, channel) table; on the 1x1 layers
// of the YOLO graphs that is 3.0-3.5 TB/s of the 4-byte-per-channel G8 tensors and 7-115 TFLOP/s on the two UFLD Linear layers.
// Both are streaming problems (activations in, activations out, weights resident / weights streamed once), so the
// split operands take conv_pw's and conv_fc's routes unchanged:
//   * a lane's MFMA fragment (8 consecutive channels of one pixel) is ONE aligned 32-byte piece of the G8 tensor:
//     [16 B hi | 16 B lo] -- two 16-byte loads at consecutive addresses, no LDS staging of activations;
//   * weights in fragment order, a hi block and a lo block of 1 KB per (16-feature tile, 32-channel K step)
//     ([tile][step][hi | lo][64 lanes][8 halves]); pointwise: resident in LDS per workgroup; Linear: streamed from HBM
//     through a U-deep register ring;
//   * three MFMAs per (tile, step): main += w_hi a_hi, cross += w_lo a_hi, cross += w_hi a_lo;
//     epilogue (main + 2^-11 cross) + bias -> activation (fp32-class forms: elem16.h x3_silu) -> split -> G8 store (8 B hi + 8 B lo per lane) or fp32.
// Same sums in the same K order as conv_x3.hip.
#include "kernels.h"
#include "elem16.h"
#include <stdlib.h>

namespace adas {

typedef __attribute__((ext_vector_type(4))) float wf32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t wu32x4;

__device__ __forceinline__ float pwx_act(float v, int act) {
    if (act == ACT_SILU) return x3_silu(v);   // (elem16.h: fp32-class, 12 instructions)
    if (act == ACT_RELU) return fmaxf(v, 0.0f);
    if (act == ACT_LEAKY) return fmaxf(v, 0.1f * v);
    return v;
}

// lane's 4 results (channels c .. c+3, c % 4 == 0) of one pixel: fp32 or G8
__device__ __forceinline__ void pwx_store(void* out, size_t o, const float v[4], int out_f32) {
    if (out_f32) *reinterpret_cast<float4*>((float*)out + o) = make_float4(v[0], v[1], v[2], v[3]);
    else x3_store4((x3s*)out + o, v);
}

struct PwxDev {
    const x3s* in;
    const uint16_t* wfrag;  // [NT][KS][2][64][8] halves
    const float* bias;
    void* out;
    int in_cs, in_coff, cin;
    int out_cs, out_coff, cout, out_f32;
    int M;
    int stride, Wo, HoWo, W, HW;
    int act;
    int NT, NTL, mtiles;
    const x3s* up;          // half-resolution source of the first up_ks K steps (nearest 2x upsample folded in), or null
    int up_cs, up_coff, up_ks, up_W, up_HW;
};

template <int KS, bool TAIL>
__global__ __launch_bounds__(512) void conv_pwx3_kernel(PwxDev a) {
    Fp16::enter();
    extern __shared__ __attribute__((aligned(16))) uint16_t wl[];  // [NTL][KS][2][64][8]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 15, kg = lane >> 4;
    const int nt0 = blockIdx.y * a.NTL;
    const int ntl = a.NT - nt0 < a.NTL ? a.NT - nt0 : a.NTL;
    float* bl = reinterpret_cast<float*>(wl + (size_t)a.NTL * KS * 1024);  // [NTL*16] bias, behind the weights
    {
        const int n16 = ntl * KS * 128;
        const uint16_t* wsrc = a.wfrag + (size_t)nt0 * KS * 1024;
        stage_lds16<512, 8>(wl, wsrc, n16, tid);
        for (int i = tid; i < ntl * 16; i += 512) bl[i] = a.bias[nt0 * 16 + i];   // bias is padded to a multiple of 128 entries
    }
    __syncthreads();
    const int tail_valid = a.cin - (KS - 1) * 32;
    const bool tail_zero = TAIL && kg * 8 >= tail_valid;

    for (int mt = blockIdx.x * 8 + wave; mt < a.mtiles; mt += gridDim.x * 8) {
        const int m = mt * 16 + lrow;
        const bool ok = m < a.M;
        const int mm = ok ? m : 0;
        size_t ipix;
        if (a.stride == 1) {
            ipix = (size_t)mm;
        } else {
            const int n = mm / a.HoWo, rem = mm - n * a.HoWo;
            const int oy = rem / a.Wo, ox = rem - oy * a.Wo;
            ipix = (size_t)n * a.HW + (size_t)(oy * a.stride) * a.W + ox * a.stride;
        }
        // 16-byte units: a group of 8 channels is two of them (hi, lo)
        const wu32x4* ip = reinterpret_cast<const wu32x4*>(a.in + ipix * a.in_cs + a.in_coff + kg * 8);
        wu32x4 xh[KS], xl[KS];
        if (a.up) {
            const int n = mm / a.HW, rem = mm - n * a.HW;
            const int oy = rem / a.W, ox = rem - oy * a.W;
            const wu32x4* up = reinterpret_cast<const wu32x4*>(a.up + ((size_t)n * a.up_HW + (size_t)(oy >> 1) * a.up_W + (ox >> 1)) * a.up_cs + a.up_coff + kg * 8);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (ks < a.up_ks) {
                    xh[ks] = up[ks * 8];
                    xl[ks] = up[ks * 8 + 1];
                } else {
                    xh[ks] = __builtin_nontemporal_load(ip + ks * 8);
                    xl[ks] = __builtin_nontemporal_load(ip + ks * 8 + 1);
                }
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                xh[ks] = __builtin_nontemporal_load(ip + ks * 8);
                xl[ks] = __builtin_nontemporal_load(ip + ks * 8 + 1);
            }
        }
        if (tail_zero) {
            xh[KS - 1] = wu32x4{0u, 0u, 0u, 0u};
            xl[KS - 1] = wu32x4{0u, 0u, 0u, 0u};
        }

        const size_t obase = (size_t)mm * a.out_cs + a.out_coff + kg * 4 + nt0 * 16;
        for (int nt = 0; nt < ntl; ++nt) {
            wf32x4 accm{0.f, 0.f, 0.f, 0.f}, accx{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const uint16_t* wp = wl + ((size_t)(nt * KS + ks) * 128 + lane) * 8;
                const wu32x4 wh = *reinterpret_cast<const wu32x4*>(wp);
                const wu32x4 wlo = *reinterpret_cast<const wu32x4*>(wp + 512);
                accm = Fp16::mfma(wh, xh[ks], accm);
                accx = Fp16::mfma(wh, xl[ks], accx);
                accx = Fp16::mfma(wlo, xh[ks], accx);
            }
            const int c = nt * 16 + kg * 4;
            if (!ok || nt0 * 16 + c >= a.cout) continue;
            const float4 b4 = *reinterpret_cast<const float4*>(bl + c);
            const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = pwx_act((accm[r] + accx[r] * kX3Down) + bb[r], a.act);
            pwx_store(a.out, obase + nt * 16, v, a.out_f32);
        }
    }
}

static const int PWX_MAX_LDS = 150 * 1024;

static int pwx_max_split() {   // ADAS_PWX3_MAXSPLIT: feature-tile ranges a layer may be cut into (each re-reads the activations); 0 = kernel off
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("ADAS_PWX3_MAXSPLIT");
        v = e ? atoi(e) : 4;
        if (v < 0) v = 0;
        if (v > 8) v = 8;
    }
    return v;
}
static int pwx_tiles_per_wg(int nt, int ks) {
    for (int split = 1; split <= pwx_max_split(); ++split) {
        const int ntl = (nt + split - 1) / split;
        if ((size_t)ntl * ks * 2048 + (size_t)ntl * 64 <= (size_t)PWX_MAX_LDS) return ntl;
    }
    return 0;
}

bool pw_x3_applicable(int kh, int kw, int stride, int pad, int res_mode, const TView& in, const TView& out) {
    if (in.f32) return false;
    if (kh != 1 || kw != 1 || pad != 0 || (stride != 1 && stride != 2) || res_mode != RES_NONE) return false;
    if (in.h == 1 && in.w == 1) return false;  // Linear layers: fc_x3
    if ((in.c & 7) || (in.cs & 7) || (in.coff & 7) || (out.c & 3) || (out.cs & 3) || (out.coff & 3)) return false;
    if (!out.f32 && ((out.c & 7) || (out.cs & 7) || (out.coff & 7))) return false;
    const int ks = (in.c + 31) / 32, nt = (out.c + 15) / 16;
    if (!(ks <= 6 || ks == 8 || ks == 10 || ks == 12 || ks == 16)) return false;
    return pwx_tiles_per_wg(nt, ks) > 0;
}

template <int KS>
static hipError_t pwx_launch(const PwxDev& d, bool tail, dim3 grid, size_t lds, hipStream_t st) {
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)conv_pwx3_kernel<KS, false>, hipFuncAttributeMaxDynamicSharedMemorySize, PWX_MAX_LDS);
        (void)hipFuncSetAttribute((const void*)conv_pwx3_kernel<KS, true>, hipFuncAttributeMaxDynamicSharedMemorySize, PWX_MAX_LDS);
        attr_done = true;
    }
    if (tail) hipLaunchKernelGGL((conv_pwx3_kernel<KS, true>), grid, dim3(512), lds, st, d);
    else hipLaunchKernelGGL((conv_pwx3_kernel<KS, false>), grid, dim3(512), lds, st, d);
    return hipGetLastError();
}

hipError_t launch_conv_pw_x3(const ConvArgs& a, hipStream_t st) {
    PwxDev d;
    d.in = (const x3s*)a.in.p; d.wfrag = (const uint16_t*)a.wgt; d.bias = a.bias; d.out = a.out.p;
    d.in_cs = a.in.cs; d.in_coff = a.in.coff; d.cin = a.in.c;
    d.out_cs = a.out.cs; d.out_coff = a.out.coff; d.cout = a.out.c; d.out_f32 = a.out.f32;
    d.M = a.m; d.stride = a.stride; d.Wo = a.out.w; d.HoWo = a.out.h * a.out.w; d.W = a.in.w; d.HW = a.in.h * a.in.w;
    d.act = a.act;
    d.up = nullptr; d.up_cs = d.up_coff = d.up_ks = d.up_W = d.up_HW = 0;
    if (a.up_c > 0) {
        if (a.stride != 1 || (a.up_c & 31) || a.up.c != a.up_c || 2 * a.up.h != a.in.h || 2 * a.up.w != a.in.w || a.up.f32 || ((a.up.cs | a.up.coff) & 7)) return hipErrorInvalidValue;
        d.up = (const x3s*)a.up.p; d.up_cs = a.up.cs; d.up_coff = a.up.coff; d.up_ks = a.up_c / 32; d.up_W = a.up.w; d.up_HW = a.up.h * a.up.w;
    }
    const int ks = (a.in.c + 31) / 32;
    d.NT = (a.out.c + 15) / 16;
    d.mtiles = (a.m + 15) / 16;
    d.NTL = pwx_tiles_per_wg(d.NT, ks);
    if (d.NTL <= 0) return hipErrorNotSupported;
    const int nsplit = (d.NT + d.NTL - 1) / d.NTL;
    const size_t lds = (size_t)d.NTL * ks * 2048 + (size_t)d.NTL * 64;
    int per_cu = (int)((160 * 1024) / (lds > 4096 ? lds : 4096));
    if (per_cu > 4) per_cu = 4;
    if (per_cu < 1) per_cu = 1;
    int gx = 256 * per_cu / nsplit;
    if (gx < 1) gx = 1;
    const int need = (d.mtiles + 7) / 8;
    if (gx > need) gx = need;
    const dim3 grid(gx, nsplit);
    const bool tail = (a.in.c & 31) != 0;
    switch (ks) {
        case 1: return pwx_launch<1>(d, tail, grid, lds, st);
        case 2: return pwx_launch<2>(d, tail, grid, lds, st);
        case 3: return pwx_launch<3>(d, tail, grid, lds, st);
        case 4: return pwx_launch<4>(d, tail, grid, lds, st);
        case 5: return pwx_launch<5>(d, tail, grid, lds, st);
        case 6: return pwx_launch<6>(d, tail, grid, lds, st);
        case 8: return pwx_launch<8>(d, tail, grid, lds, st);
        case 10: return pwx_launch<10>(d, tail, grid, lds, st);
        case 12: return pwx_launch<12>(d, tail, grid, lds, st);
        case 16: return pwx_launch<16>(d, tail, grid, lds, st);
    }
    return hipErrorInvalidValue;
}

// ------------------------------------------------------------------------------------- Linear
struct FcxDev {
    const x3s* x;         // [batch][x_cs] G8 (+ x_coff)
    const uint16_t* w;    // [cout_pad/16][kpad/32][2][64][8] halves
    const float* bias;
    void* out;            // [batch][out_cs] (+ out_coff), G8 or fp32
    int x_cs, x_coff, out_cs, out_coff;
    int batch, cout, kpad, act, out_f32;
};

template <int TN, int TM, int KS, int U>
__global__ __launch_bounds__(64 * KS) void fc_x3_kernel(FcxDev a) {
    Fp16::enter();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 15, kg = lane >> 4;
    const int n0 = blockIdx.x * (TN * 16);
    const int KT = a.kpad >> 5;
    const int ks0 = (int)((long)KT * wave / KS), ks1 = (int)((long)KT * (wave + 1) / KS);
    const int n = ks1 - ks0;

    // 16-byte units: block (tile, step) is 64 lanes of hi then 64 lanes of lo
    const wu32x4* wp[TN];
#pragma unroll
    for (int i = 0; i < TN; ++i) wp[i] = reinterpret_cast<const wu32x4*>(a.w) + ((size_t)(blockIdx.x * TN + i) * KT + ks0) * 128 + lane;
    const wu32x4* xp[TM];
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int m = j * 16 + lrow;
        xp[j] = reinterpret_cast<const wu32x4*>(a.x + (size_t)(m < a.batch ? m : 0) * a.x_cs + a.x_coff + kg * 8 + ks0 * 32);
    }

    wf32x4 accm[TN][TM], accx[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            accm[i][j] = wf32x4{0.f, 0.f, 0.f, 0.f};
            accx[i][j] = wf32x4{0.f, 0.f, 0.f, 0.f};
        }

    wu32x4 wh[U][TN], wlo[U][TN], xh[U][TM], xl[U][TM];
    auto load = [&](int buf, int s) {  // s: K step relative to ks0
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            wh[buf][i] = __builtin_nontemporal_load(wp[i] + (size_t)s * 128);
            wlo[buf][i] = __builtin_nontemporal_load(wp[i] + (size_t)s * 128 + 64);
        }
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            xh[buf][j] = xp[j][s * 8];
            xl[buf][j] = xp[j][s * 8 + 1];
        }
    };
    auto mma = [&](int buf) {
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                accm[i][j] = Fp16::mfma(wh[buf][i], xh[buf][j], accm[i][j]);
                accx[i][j] = Fp16::mfma(wh[buf][i], xl[buf][j], accx[i][j]);
                accx[i][j] = Fp16::mfma(wlo[buf][i], xh[buf][j], accx[i][j]);
            }
    };
    if (n >= U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            load(u, u);
            __builtin_amdgcn_sched_barrier(0);   // ring stages stay in issue order (conv_fc.hip)
        }
        const int main_steps = ((n - U) / U) * U;
        int s = 0;
        for (; s < main_steps; s += U) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                mma(u);
                __builtin_amdgcn_sched_barrier(0);
                load(u, s + u + U);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            mma(u);
            if (s + u + U < n) load(u, s + u + U);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (s + U + u < n) mma(u);
    } else {
        for (int s = 0; s < n; ++s) {
            load(0, s);
            mma(0);
        }
    }

    if constexpr (KS > 1) {
        __shared__ float red[KS > 1 ? KS - 1 : 1][TN * TM * 8][64];
        if (wave > 0) {
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        red[wave - 1][(i * TM + j) * 8 + r][lane] = accm[i][j][r];
                        red[wave - 1][(i * TM + j) * 8 + 4 + r][lane] = accx[i][j][r];
                    }
        }
        __syncthreads();
        if (wave > 0) return;
#pragma unroll
        for (int s = 0; s < KS - 1; ++s)
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        accm[i][j][r] += red[s][(i * TM + j) * 8 + r][lane];
                        accx[i][j][r] += red[s][(i * TM + j) * 8 + 4 + r][lane];
                    }
    }

#pragma unroll
    for (int i = 0; i < TN; ++i) {
        const int c = n0 + i * 16 + kg * 4;
        if (c >= a.cout) continue;
        const float4 b4 = *reinterpret_cast<const float4*>(a.bias + c);
        const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int m = j * 16 + lrow;
            if (m >= a.batch) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = pwx_act((accm[i][j][r] + accx[i][j][r] * kX3Down) + bb[r], a.act);
            pwx_store(a.out, (size_t)m * a.out_cs + a.out_coff + c, v, a.out_f32);
        }
    }
}

bool fc_x3_applicable(int kh, int kw, int stride, const TView& in, const TView& out) {
    static int off = -1;
    if (off < 0) {
        const char* e = getenv("ADAS_NO_FC_X3");
        off = (e && e[0] == '1') ? 1 : 0;
    }
    if (off || in.f32) return false;
    if (kh != 1 || kw != 1 || stride != 1 || in.h != 1 || in.w != 1 || out.h != 1 || out.w != 1) return false;
    if ((in.cs & 7) || (in.coff & 7) || (in.c & 7) || (out.c & 3) || (out.cs & 3) || (out.coff & 3)) return false;
    if (!out.f32 && ((out.c & 7) || (out.cs & 7) || (out.coff & 7))) return false;
    return true;
}

template <int TN, int TM, int KS, int U>
static hipError_t fcx_launch(const FcxDev& d, hipStream_t st) {
    const int tiles = (d.cout + TN * 16 - 1) / (TN * 16);
    hipLaunchKernelGGL((fc_x3_kernel<TN, TM, KS, U>), dim3(tiles), dim3(64 * KS), 0, st, d);
    return hipGetLastError();
}

static int fcx_wide_tn() {   // 16-feature tiles per wave on the wide (weight-streaming) layers; ADAS_FCX3_TN overrides (2 | 4)
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("ADAS_FCX3_TN");
        v = e ? atoi(e) : 4;
        if (v != 2 && v != 4) v = 4;
    }
    return v;
}

hipError_t launch_fc_x3(const ConvArgs& a, hipStream_t st) {
    for (int r0 = 0; r0 < a.n; r0 += 64) {  // groups of <= 64 batch rows: the weights stream once per group
        FcxDev d;
        const int rows = a.n - r0 < 64 ? a.n - r0 : 64;
        d.x = (const x3s*)a.in.p + (size_t)r0 * a.in.cs;
        d.w = (const uint16_t*)a.wgt; d.bias = a.bias;
        d.out = (char*)a.out.p + (size_t)r0 * a.out.cs * 4;   // fp32 and G8 are both 4 bytes per channel
        d.x_cs = a.in.cs; d.x_coff = a.in.coff; d.out_cs = a.out.cs; d.out_coff = a.out.coff;
        d.batch = rows; d.cout = a.out.c; d.kpad = a.kpad; d.act = a.act; d.out_f32 = a.out.f32;
        const int tm = rows <= 16 ? 1 : (rows <= 32 ? 2 : 4);
        hipError_t e;
        if (a.out.c <= 8192) {   // few output tiles: 16 features per workgroup, K split over 4 waves
            e = tm == 1 ? fcx_launch<1, 1, 4, 3>(d, st) : (tm == 2 ? fcx_launch<1, 2, 4, 3>(d, st) : fcx_launch<1, 4, 4, 3>(d, st));
        } else if (fcx_wide_tn() == 4) {
            e = tm == 1 ? fcx_launch<4, 1, 1, 3>(d, st) : (tm == 2 ? fcx_launch<4, 2, 1, 3>(d, st) : fcx_launch<4, 4, 1, 2>(d, st));
        } else {
            e = tm == 1 ? fcx_launch<2, 1, 1, 3>(d, st) : (tm == 2 ? fcx_launch<2, 2, 1, 3>(d, st) : fcx_launch<2, 4, 1, 3>(d, st));
        }
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// fp32 [cout][cin] -> fragment order [cout_pad/16][kpad/32][hi | lo][64 lanes][8]: lane = (k%32/8)*16 + row%16
__global__ void pack_weights_fcx3_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst, int cout, int cin, int kpad, size_t total) {
    const int KT = kpad >> 5;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int e = (int)(i & 7), lane = (int)((i >> 3) & 63);
        const size_t blk = i >> 9;
        const int ks = (int)(blk % KT);
        const size_t tile = blk / KT;
        const size_t row = tile * 16 + (lane & 15);
        const int k = ks * 32 + (lane >> 4) * 8 + e;
        const float v = (row < (size_t)cout && k < cin) ? src[row * cin + k] : 0.0f;
        _Float16 h, l;
        x3_split(v, h, l);
        const size_t o = blk * 1024 + (size_t)lane * 8 + e;
        dst[o] = __builtin_bit_cast(uint16_t, h);
        dst[o + 512] = __builtin_bit_cast(uint16_t, l);
    }
}

hipError_t launch_pack_weights_fcx3(const float* src, void* dst, int cout, int cout_pad, int cin, int kpad, hipStream_t st) {
    (void)cout_pad;
    const size_t total = (size_t)cout_pad * kpad;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(pack_weights_fcx3_kernel, dim3(blocks), dim3(256), 0, st, src, (uint16_t*)dst, cout, cin, kpad, total);
    return hipGetLastError();
}

}  // namespace adas
