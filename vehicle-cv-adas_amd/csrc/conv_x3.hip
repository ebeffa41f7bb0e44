// conv_x3.hip -- the convolution of the split precision (ADAS_PREC_FP16X3): f32-class results on the 16-bit matrix cores.
//
//   out[m][co] = act( bias[co] + sum_k a[m][k] * w[co][k] ) (+ residual),   a = a_hi + 2^-11 a_lo,  w = w_hi + 2^-11 w_lo  (elem16.h)
//   sum a*w = sum a_hi*w_hi  +  2^-11 ( sum a_hi*w_lo + sum a_lo*w_hi )      [the a_lo*w_lo term is 2^-22 of a product: dropped]
//
// Three v_mfma_f32_16x16x32_f16 per (fragment pair, 32-deep K step) into TWO fp32 accumulator sets -- `main` (hi*hi) and `cross`
// (hi*lo + lo*hi, both carrying the same 2^11 scale) -- combined once in the epilogue.  Products of two halves are exact in fp32, so
// what is lost against an f32 convolution is the 2^-22 representation of the operands and the dropped lo*lo term: the parity mode's
// error class at 3/16 of its MFMA cost (the f32-input MFMA runs at 1/16 of the 16-bit rate: ADAS_PREC_FP32).
//
// This file: the GENERIC kernel (any kernel size / stride / channel count multiple of 8), the implicit-GEMM structure of
// conv_kernels.hip with both halves of every operand staged -- per 32-deep K step a workgroup (4 waves) gathers the BM x 32
// activation slab and the BN x 32 weight slab, each as a hi and a lo plane, into padded LDS rows; a wave's K step is 12 fragment
// reads for 24 MFMAs.  Storage is the G8 layout of elem16.h (a K chunk of 8 channels = 16 bytes of hi + 16 bytes of lo, adjacent),
// for activations and for the packed weights alike.
#include "kernels.h"
#include "elem16.h"
#include <stdlib.h>
#include <type_traits>

namespace adas {

typedef __attribute__((ext_vector_type(4))) float xf32x4;

struct X3Dev {
    const void* in;
    const void* wgt;
    const float* bias;
    void* out;
    const void* res;
    int in_cs, in_coff, cin, H, W;
    int out_cs, out_coff, cout, Ho, Wo;
    int res_cs, res_coff, res_mode;
    int kh, kw, stride, pad, act;
    int nq, kpad, M;
};

#define ADAS_X3_MAX_Q 1152   // K / 8 chunks (conv_kernels.hip ADAS_MAX_Q)

__device__ __forceinline__ float x3_act(float v, int act) {
    if (act == ACT_SILU) return x3_silu(v);
    if (act == ACT_RELU) return fmaxf(v, 0.0f);
    if (act == ACT_LEAKY) return fmaxf(v, 0.1f * v);
    return v;
}

template <int BM, int BN, int WM, int WN, bool OUT_F32>
__global__ __launch_bounds__(256) void conv_x3_igemm_kernel(X3Dev a) {
    Fp16::enter();
    constexpr int LDK = 40;   // 32 halves + 8 of padding: 80-byte rows, conflict-free ds_read_b128
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    constexpr int A_IT = (BM * 4 + 255) / 256, B_IT = (BN * 4 + 255) / 256;
    static_assert(WM * WN == 4 && TM >= 1 && TN >= 1, "tile");
    __shared__ __attribute__((aligned(16))) uint16_t Ah[2][BM][LDK], Al[2][BM][LDK];
    __shared__ __attribute__((aligned(16))) uint16_t Bh[2][BN][LDK], Bl[2][BN][LDK];
    __shared__ uint16_t ktab[ADAS_X3_MAX_Q];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const x3s* __restrict__ in = (const x3s*)a.in;
    const x3s* __restrict__ wgt = (const x3s*)a.wgt;

    // K chunk table: q -> (r, s, c8), 3 + 3 + 10 bits
    const int cin8 = a.cin >> 3;
    for (int q = tid; q < a.nq; q += 256) {
        int tap = q / cin8, c8 = q - tap * cin8;
        int r = tap / a.kw, s = tap - r * a.kw;
        ktab[q] = (uint16_t)((r << 13) | (s << 10) | c8);
    }
    const int kc = tid & 3;
    int iy0[A_IT], ix0[A_IT], pb[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        int row = (tid >> 2) + 64 * i;
        int m = m0 + row;
        bool ok = (row < BM) && (m < a.M);
        int mm = ok ? m : 0;
        int hw = a.Ho * a.Wo;
        int n = mm / hw, rem = mm - n * hw;
        int oy = rem / a.Wo, ox = rem - oy * a.Wo;
        iy0[i] = oy * a.stride - a.pad;
        ix0[i] = ox * a.stride - a.pad;
        pb[i] = ok ? n * a.H * a.W : -1;
    }
    __syncthreads();

    const int KT = a.kpad >> 5;
    // two register stages: K step ks + 2 is requested while ks + 1 waits in the other stage and ks is multiplied out of LDS -- with one
    // stage a step's global latency had one step's MFMAs (12-24 per wave) to hide behind, and a small map's launch (one workgroup per CU
    // or fewer, 144 steps for a 512-channel 3x3 layer) ran at ~0.8 us per step
    e_u32x4 rah2[2][A_IT], ral2[2][A_IT], rbh2[2][B_IT], rbl2[2][B_IT];
    const e_u32x4 zero4 = {0u, 0u, 0u, 0u};

    auto gload = [&](auto set_c, int ks) {
        constexpr int SET = decltype(set_c)::value;
        e_u32x4 (&rah)[A_IT] = rah2[SET], (&ral)[A_IT] = ral2[SET], (&rbh)[B_IT] = rbh2[SET], (&rbl)[B_IT] = rbl2[SET];
        const int q = ks * 4 + kc;
        int r = 0, s = 0, c8 = 0;
        const bool qok = q < a.nq;
        if (qok) {
            int e = ktab[q];
            r = e >> 13;
            s = (e >> 10) & 7;
            c8 = e & 1023;
        }
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            int iy = iy0[i] + r, ix = ix0[i] + s;
            bool ok = qok && pb[i] >= 0 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            // always-valid address + select (a branch around the load would serialise the A_IT loads on vmcnt(0))
            const size_t pix = ok ? (size_t)(pb[i] + iy * a.W + ix) : 0;
            const e_u32x4* g = reinterpret_cast<const e_u32x4*>(in + (pix * a.in_cs + a.in_coff + c8 * 8));
            const e_u32x4 h = g[0], l = g[1];
            rah[i] = ok ? h : zero4;
            ral[i] = ok ? l : zero4;
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            int row = (tid >> 2) + 64 * i;
            if (row < BN) {
                const e_u32x4* g = reinterpret_cast<const e_u32x4*>(wgt + ((size_t)(n0 + row) * a.kpad + ks * 32 + kc * 8));
                rbh[i] = g[0];
                rbl[i] = g[1];
            }
        }
    };
    auto lstore = [&](auto set_c, int buf) {
        constexpr int SET = decltype(set_c)::value;
        e_u32x4 (&rah)[A_IT] = rah2[SET], (&ral)[A_IT] = ral2[SET], (&rbh)[B_IT] = rbh2[SET], (&rbl)[B_IT] = rbl2[SET];
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            int row = (tid >> 2) + 64 * i;
            if (row < BM) {
                *reinterpret_cast<e_u32x4*>(&Ah[buf][row][kc * 8]) = rah[i];
                *reinterpret_cast<e_u32x4*>(&Al[buf][row][kc * 8]) = ral[i];
            }
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            int row = (tid >> 2) + 64 * i;
            if (row < BN) {
                *reinterpret_cast<e_u32x4*>(&Bh[buf][row][kc * 8]) = rbh[i];
                *reinterpret_cast<e_u32x4*>(&Bl[buf][row][kc * 8]) = rbl[i];
            }
        }
    };

    xf32x4 accm[TN][TM], accx[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) accm[i][j] = accx[i][j] = xf32x4{0.f, 0.f, 0.f, 0.f};

    const int lrow = lane & 15, kg = lane >> 4;
    typedef std::integral_constant<int, 0> S0;
    typedef std::integral_constant<int, 1> S1;
    gload(S0{}, 0);
    lstore(S0{}, 0);
    if (KT > 1) gload(S1{}, 1);
    __syncthreads();
    // (step ks: LDS buffer ks & 1 holds it, stage (ks + 1) & 1 holds step ks + 1; the stage index is a compile-time constant per parity)
    auto step = [&](auto par_c, const int ks) {
        constexpr int PAR = decltype(par_c)::value;
        const int buf = PAR;
        // (unconditional -- the last two steps re-request step KT - 1: with the request under a condition the compiler has to wait for EVERY
        // outstanding load before the stage that is stored below, the fresh ones included)
        gload(std::integral_constant<int, PAR>{}, ks + 2 < KT ? ks + 2 : KT - 1);
        e_u32x4 wh[TN], wl[TN], xh[TM], xl[TM];
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            wh[i] = *reinterpret_cast<const e_u32x4*>(&Bh[buf][(wn * TN + i) * 16 + lrow][kg * 8]);
            wl[i] = *reinterpret_cast<const e_u32x4*>(&Bl[buf][(wn * TN + i) * 16 + lrow][kg * 8]);
        }
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            xh[j] = *reinterpret_cast<const e_u32x4*>(&Ah[buf][(wm * TM + j) * 16 + lrow][kg * 8]);
            xl[j] = *reinterpret_cast<const e_u32x4*>(&Al[buf][(wm * TM + j) * 16 + lrow][kg * 8]);
        }
        // three passes over the tile grid: the two MFMAs into one cross accumulator are TM * TN instructions apart (no dependent stall)
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j) accm[i][j] = Fp16::mfma(wh[i], xh[j], accm[i][j]);
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j) accx[i][j] = Fp16::mfma(wl[i], xh[j], accx[i][j]);
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j) accx[i][j] = Fp16::mfma(wh[i], xl[j], accx[i][j]);
        if (ks + 1 < KT) lstore(std::integral_constant<int, PAR ^ 1>{}, buf ^ 1);
        __syncthreads();
    };
    for (int ks = 0; ks < KT; ks += 2) {
        step(S0{}, ks);
        if (ks + 1 < KT) step(S1{}, ks + 1);
    }

    // ---- fused epilogue: lane holds channels c..c+3 of pixel m
    const bool vec_ok = ((a.cout & 3) == 0) && ((a.out_cs & 3) == 0) && ((a.out_coff & 3) == 0);
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int m = m0 + (wm * TM + j) * 16 + lrow;
        if (m >= a.M) continue;
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            const int c = n0 + (wn * TN + i) * 16 + kg * 4;
            if (c >= a.cout) continue;
            float v[4];
            const float4 b = *reinterpret_cast<const float4*>(a.bias + c);   // bias is padded to 128
            const float bb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
            for (int t = 0; t < 4; ++t) v[t] = (accm[i][j][t] + accx[i][j][t] * kX3Down) + bb[t];
            if (a.res_mode != RES_NONE) {
                float rv[4];
                x3_load4((const x3s*)a.res + ((size_t)m * a.res_cs + a.res_coff + c), rv);
                if (a.res_mode == RES_BEFORE_ACT) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) v[t] = x3_act(v[t] + rv[t], a.act);
                } else {
#pragma unroll
                    for (int t = 0; t < 4; ++t) v[t] = x3_act(v[t], a.act) + rv[t];
                }
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) v[t] = x3_act(v[t], a.act);
            }
            const size_t o = (size_t)m * a.out_cs + a.out_coff + c;
            if (OUT_F32) {
                float* op = (float*)a.out + o;
                if (vec_ok && c + 3 < a.cout) {
                    *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        if (c + t < a.cout) op[t] = v[t];
                }
            } else {
                x3s* op = (x3s*)a.out + o;
                if (vec_ok && c + 3 < a.cout) {
                    x3_store4(op, v);
                } else {
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        if (c + t < a.cout) x3_st(op + t, v[t]);
                }
            }
        }
    }
}

// ---- small maps (one frame at a time): the K loop split over the waves of the workgroup.
// conv_x3_igemm's workgroup walks all of K behind one barrier per 32-deep step; a 10x50 map of a 512-channel layer is 8-16 tiles, so a
// handful of workgroups each run 144 dependent steps (0.57 us each even with two register stages: 82 us for 2.4 GFLOP) while the chip
// idles.  Here a workgroup of eight waves owns a 32 x 32 tile; wave w takes the K steps w, w + 8, ... (its MFMA fragments straight from
// global memory, as conv_pwx3 reads them: a fragment is one pixel's / one weight row's 16 bytes of hi and 16 of lo -- no LDS staging, no
// barrier in the loop, two steps in flight per wave), and the eight partial tiles are summed in a fixed tree through LDS (deterministic:
// the same bits every run), wave 0 applying the epilogue.
constexpr int KSW = 8;   // waves per workgroup = K slices
// TN: 16-channel tiles per workgroup (2: a 32 x 32 tile; 4: 32 pixels x 64 channels, taken when the 32 x 32 grid would need a second round
// of workgroups -- the launch is one workgroup's latency chain per round)
template <bool OUT_F32, int TN>
__global__ __launch_bounds__(64 * KSW) void conv_x3_ksplit_kernel(X3Dev a) {
    Fp16::enter();
    constexpr int TM = 2;
    __shared__ uint16_t ktab[ADAS_X3_MAX_Q];
    __shared__ __attribute__((aligned(16))) float red[KSW / 2][TM * TN * 8][64];   // [writer][value][lane]: 32 / 64 KB

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 15, kg = lane >> 4;
    const int m0 = blockIdx.x * 32, n0 = blockIdx.y * (TN * 16);
    const x3s* __restrict__ in = (const x3s*)a.in;
    const x3s* __restrict__ wgt = (const x3s*)a.wgt;
    const int cin8 = a.cin >> 3;
    for (int q = tid; q < a.nq; q += 64 * KSW) {
        int tap = q / cin8, c8 = q - tap * cin8;
        int r = tap / a.kw, s2 = tap - r * a.kw;
        ktab[q] = (uint16_t)((r << 13) | (s2 << 10) | c8);
    }
    int iy0[TM], ix0[TM], pb[TM];
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int m = m0 + j * 16 + lrow;
        const bool ok = m < a.M;
        const int mm = ok ? m : 0;
        const int hw = a.Ho * a.Wo;
        const int n = mm / hw, rem = mm - n * hw;
        const int oy = rem / a.Wo, ox = rem - oy * a.Wo;
        iy0[j] = oy * a.stride - a.pad;
        ix0[j] = ox * a.stride - a.pad;
        pb[j] = ok ? n * a.H * a.W : -1;
    }
    __syncthreads();

    const int KT = a.kpad >> 5;
    const e_u32x4 zero4 = {0u, 0u, 0u, 0u};
    e_u32x4 fwh[2][TN], fwl[2][TN], fxh[2][TM], fxl[2][TM];   // two stages of fragments
    auto fetch = [&](auto set_c, int ks) {
        constexpr int SET = decltype(set_c)::value;
        const int q = ks * 4 + kg;
        const bool qok = ks < KT && q < a.nq;
        const int e = qok ? (int)ktab[q] : 0;
        const int r = e >> 13, s2 = (e >> 10) & 7, c8 = e & 1023;
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int iy = iy0[j] + r, ix = ix0[j] + s2;
            const bool ok = qok && pb[j] >= 0 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            const size_t pix = ok ? (size_t)(pb[j] + iy * a.W + ix) : 0;      // always-valid address + select: no branch around the loads
            const e_u32x4* g = reinterpret_cast<const e_u32x4*>(in + (pix * a.in_cs + a.in_coff + c8 * 8));
            const e_u32x4 h = g[0], l = g[1];
            fxh[SET][j] = ok ? h : zero4;
            fxl[SET][j] = ok ? l : zero4;
        }
        const int ksc = ks < KT ? ks : KT - 1;                                 // (past the end: a valid address, multiplied by zero pixels)
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            const e_u32x4* g = reinterpret_cast<const e_u32x4*>(wgt + ((size_t)(n0 + i * 16 + lrow) * a.kpad + ksc * 32 + kg * 8));
            fwh[SET][i] = g[0];
            fwl[SET][i] = g[1];
        }
    };
    xf32x4 accm[TN][TM], accx[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) accm[i][j] = accx[i][j] = xf32x4{0.f, 0.f, 0.f, 0.f};
    auto mma = [&](auto set_c) {
        constexpr int SET = decltype(set_c)::value;
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j) accm[i][j] = Fp16::mfma(fwh[SET][i], fxh[SET][j], accm[i][j]);
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j) accx[i][j] = Fp16::mfma(fwl[SET][i], fxh[SET][j], accx[i][j]);
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j) accx[i][j] = Fp16::mfma(fwh[SET][i], fxl[SET][j], accx[i][j]);
    };
    typedef std::integral_constant<int, 0> S0;
    typedef std::integral_constant<int, 1> S1;
    // wave w: steps w, w + 8, ...; requests run one step ahead of the multiplies (unconditional: past the end they fetch zeros)
    fetch(S0{}, wave);
    for (int ks = wave; ks < KT; ks += 2 * KSW) {
        fetch(S1{}, ks + KSW);
        mma(S0{});
        fetch(S0{}, ks + 2 * KSW);
        mma(S1{});          // (step ks + 8; zeros when it lies past the end)
    }

    // ---- the eight partial tiles: 4-7 -> 0-3, 2-3 -> 0-1, 1 -> 0 (a fixed order: bit-identical from run to run)
    auto put = [&](int slot) {
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    red[slot][((i * TM + j) * 4 + t) * 2][lane] = accm[i][j][t];
                    red[slot][((i * TM + j) * 4 + t) * 2 + 1][lane] = accx[i][j][t];
                }
    };
    auto add = [&](int slot) {
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    accm[i][j][t] += red[slot][((i * TM + j) * 4 + t) * 2][lane];
                    accx[i][j][t] += red[slot][((i * TM + j) * 4 + t) * 2 + 1][lane];
                }
    };
#pragma unroll
    for (int half = KSW / 2; half >= 1; half >>= 1) {
        if (wave >= half && wave < 2 * half) put(wave - half);
        __syncthreads();
        if (wave < half) add(wave);
        __syncthreads();
    }
    if (wave != 0) return;

    // ---- epilogue (conv_x3_igemm_kernel's): lane holds channels c..c+3 of pixel m
    const bool vec_ok = ((a.cout & 3) == 0) && ((a.out_cs & 3) == 0) && ((a.out_coff & 3) == 0);
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int m = m0 + j * 16 + lrow;
        if (m >= a.M) continue;
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            const int c = n0 + i * 16 + kg * 4;
            if (c >= a.cout) continue;
            float v[4];
            const float4 b = *reinterpret_cast<const float4*>(a.bias + c);   // bias is padded to 128
            const float bb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
            for (int t = 0; t < 4; ++t) v[t] = (accm[i][j][t] + accx[i][j][t] * kX3Down) + bb[t];
            if (a.res_mode != RES_NONE) {
                float rv[4];
                x3_load4((const x3s*)a.res + ((size_t)m * a.res_cs + a.res_coff + c), rv);
                if (a.res_mode == RES_BEFORE_ACT) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) v[t] = x3_act(v[t] + rv[t], a.act);
                } else {
#pragma unroll
                    for (int t = 0; t < 4; ++t) v[t] = x3_act(v[t], a.act) + rv[t];
                }
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) v[t] = x3_act(v[t], a.act);
            }
            const size_t o = (size_t)m * a.out_cs + a.out_coff + c;
            if (OUT_F32) {
                float* op = (float*)a.out + o;
                if (vec_ok && c + 3 < a.cout) {
                    *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        if (c + t < a.cout) op[t] = v[t];
                }
            } else {
                x3s* op = (x3s*)a.out + o;
                if (vec_ok && c + 3 < a.cout) {
                    x3_store4(op, v);
                } else {
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        if (c + t < a.cout) x3_st(op + t, v[t]);
                }
            }
        }
    }
}

// the K-split kernel takes a launch whose 64 x 64 tiling would be at most 128 workgroups (ADAS_X3_KSPLIT_TILES = 129; measured at one frame: 128 tiles 0.047 -> 0.039 ms, 250 tiles 0.027 -> 0.041) and whose K loop is at least 16 steps
// (ADAS_X3_KSPLIT=0: off)
static bool x3_ksplit_applies(const ConvArgs& a) {
    static int on = -1;
    if (on < 0) { const char* e = getenv("ADAS_X3_KSPLIT"); on = (e && e[0] == '0') ? 0 : 1; }
    if (!on || a.out.c < 32 || (a.kpad >> 5) < 16) return false;
    static int lim = -1;
    if (lim < 0) { const char* e = getenv("ADAS_X3_KSPLIT_TILES"); lim = e ? atoi(e) : 129; }
    return (long)((a.m + 63) / 64) * ((a.out.c + 63) / 64) < lim;
}

struct X3Tile {
    int bm, bn;
};
static X3Tile x3_pick_tile(const ConvArgs& a) {
    int bn = a.out.c <= 16 ? 16 : (a.out.c <= 32 ? 32 : 64);
    const long tiles128 = (long)((a.m + 127) / 128) * ((a.out.c + bn - 1) / bn);
    if (tiles128 >= 512 && a.m > 64) return X3Tile{128, bn};
    // small maps (one frame at a time: 10x50 .. 40x40 pixels): a 64 x 64 tiling leaves most CUs without a workgroup while each of the few
    // walks the whole K loop; 32-row tiles, then 32-column tiles, until the launch has a workgroup for every other CU (ADAS_X3_SMALL_TILES=0: off)
    static int small = -1;
    if (small < 0) { const char* e = getenv("ADAS_X3_SMALL_TILES"); small = (e && e[0] == '0') ? 0 : 1; }
    int bm = 64;
    if (small && bn >= 32) {
        if ((long)((a.m + 63) / 64) * ((a.out.c + bn - 1) / bn) < 128) bm = 32;
        if (bm == 32 && bn == 64 && (long)((a.m + 31) / 32) * ((a.out.c + 63) / 64) < 128) bn = 32;
    }
    return X3Tile{bm, bn};
}

static bool x3_ksplit_applies(const ConvArgs& a);
const char* conv_x3_kernel_name(const ConvArgs& a) {
    static thread_local char buf[64];
    if (x3_ksplit_applies(a)) {
        snprintf(buf, sizeof(buf), "conv_x3_ksplit_kernel%s", a.out.f32 ? "<f32>" : "");
        return buf;
    }
    const X3Tile t = x3_pick_tile(a);
    snprintf(buf, sizeof(buf), "conv_x3_igemm_kernel<%d,%d%s>", t.bm, t.bn, a.out.f32 ? ",f32" : "");
    return buf;
}

template <bool OUT_F32>
static hipError_t launch_x3_typed(const X3Dev& d, X3Tile t, hipStream_t st) {
    dim3 grid((d.M + t.bm - 1) / t.bm, (d.cout + t.bn - 1) / t.bn);
#define LAUNCH(BM_, BN_, WM_, WN_)                                                                           \
    if (t.bm == BM_ && t.bn == BN_) {                                                                         \
        hipLaunchKernelGGL((conv_x3_igemm_kernel<BM_, BN_, WM_, WN_, OUT_F32>), grid, dim3(256), 0, st, d);   \
        return hipGetLastError();                                                                             \
    }
    LAUNCH(128, 64, 2, 2)
    LAUNCH(128, 32, 4, 1)
    LAUNCH(128, 16, 4, 1)
    LAUNCH(64, 64, 2, 2)
    LAUNCH(64, 32, 2, 2)
    LAUNCH(64, 16, 4, 1)
    LAUNCH(32, 64, 2, 2)
    LAUNCH(32, 32, 2, 2)
#undef LAUNCH
    return hipErrorInvalidValue;
}

hipError_t launch_conv_x3(const ConvArgs& a, hipStream_t st) {
    X3Dev d;
    d.in = a.in.p; d.wgt = a.wgt; d.bias = a.bias; d.out = a.out.p; d.res = a.res.p;
    d.in_cs = a.in.cs; d.in_coff = a.in.coff; d.cin = a.in.c; d.H = a.in.h; d.W = a.in.w;
    d.out_cs = a.out.cs; d.out_coff = a.out.coff; d.cout = a.out.c; d.Ho = a.out.h; d.Wo = a.out.w;
    d.res_cs = a.res.cs; d.res_coff = a.res.coff; d.res_mode = a.res_mode;
    d.kh = a.kh; d.kw = a.kw; d.stride = a.stride; d.pad = a.pad; d.act = a.act;
    d.nq = a.k / 8; d.kpad = a.kpad; d.M = a.m;
    if (d.nq > ADAS_X3_MAX_Q || (a.in.c & 7) || (a.in.cs & 7) || (a.in.coff & 7) || a.in.c > 8184 || a.kh > 7 || a.kw > 7) return hipErrorInvalidValue;
    if (a.in.f32) return hipErrorInvalidValue;                       // conv inputs are always in the compute type
    if (!a.out.f32 && ((a.out.cs | a.out.coff) & 7)) return hipErrorInvalidValue;   // G8 groups: 8-channel aligned views
    if (a.res_mode != RES_NONE && (a.res.f32 || ((a.res.cs | a.res.coff) & 7))) return hipErrorInvalidValue;
    if (x3_ksplit_applies(a)) {
        const long g32 = (long)((d.M + 31) / 32) * ((d.cout + 31) / 32);
        static int wide = -1;
        if (wide < 0) { const char* e = getenv("ADAS_X3_KSPLIT_WIDE"); wide = e ? atoi(e) : 320; }
        if (d.cout >= 64 && g32 > wide) {     // 32 x 64 tiles: one round of workgroups instead of two
            const dim3 grid((d.M + 31) / 32, (d.cout + 63) / 64);
            if (a.out.f32) hipLaunchKernelGGL((conv_x3_ksplit_kernel<true, 4>), grid, dim3(64 * KSW), 0, st, d);
            else hipLaunchKernelGGL((conv_x3_ksplit_kernel<false, 4>), grid, dim3(64 * KSW), 0, st, d);
        } else {
            const dim3 grid((d.M + 31) / 32, (d.cout + 31) / 32);
            if (a.out.f32) hipLaunchKernelGGL((conv_x3_ksplit_kernel<true, 2>), grid, dim3(64 * KSW), 0, st, d);
            else hipLaunchKernelGGL((conv_x3_ksplit_kernel<false, 2>), grid, dim3(64 * KSW), 0, st, d);
        }
        return hipGetLastError();
    }
    const X3Tile t = x3_pick_tile(a);
    return a.out.f32 ? launch_x3_typed<true>(d, t, st) : launch_x3_typed<false>(d, t, st);
}

// fp32 [cout][taps][cin] -> G8 [cout_pad][kpad / 8][hi 8 | lo 8], element (row, tap * cin_pad + c), zero padded
__global__ void pack_weights_x3_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst, int cout, int taps, int cin, int cin_pad, int kpad,
                                       size_t total) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t row = i / kpad;
        const int col = (int)(i - row * kpad);
        const int tap = col / cin_pad, c = col - tap * cin_pad;
        const float v = (row < (size_t)cout && tap < taps && c < cin) ? src[(row * taps + tap) * cin + c] : 0.0f;
        _Float16 h, l;
        x3_split(v, h, l);
        const size_t g = (i >> 3) * 16 + (i & 7);
        dst[g] = __builtin_bit_cast(uint16_t, h);
        dst[g + 8] = __builtin_bit_cast(uint16_t, l);
    }
}
hipError_t launch_pack_weights_x3(const float* src, void* dst, int cout, int cout_pad, int taps, int cin, int cin_pad, int kpad, hipStream_t st) {
    if (kpad & 7) return hipErrorInvalidValue;
    const size_t total = (size_t)cout_pad * kpad;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(pack_weights_x3_kernel, dim3(blocks), dim3(256), 0, st, src, (uint16_t*)dst, cout, taps, cin, cin_pad, kpad, total);
    return hipGetLastError();
}

}  // namespace adas
