// conv_pwg.hip -- pointwise (1x1, stride 1) convolution as a K-looped MFMA GEMM: the 1x1 layers conv_pw.hip does not take
// (Cin > 512: its weights-resident-in-LDS design stops there; and the K-step counts it is not instantiated for, e.g. YOLOv8x's 160 / 400).  These are the C2f / SPPF output convs over concats of the s / m / l / x
// scales (YOLOv8s 768 / 1024, YOLOv8l 1024 .. 2048, YOLOv8x up to 2560 input channels): arithmetic intensity
// 2*Cin*Cout / (2*(Cin+Cout)) > 300 FLOP/B -- MFMA-bound GEMMs, not streaming layers.
//
//   out[m][n] = act(bias[n] + sum_k in[m][k] * w[n][k]) (+ residual),  M = N*H*W pixels, K = Cin
//
// A workgroup (4 waves) owns BM pixels x 128 output channels and walks K in steps of 64: both operands are K-contiguous
// (NHWC activations, CONV_GATHER weight rows), so a step is 16-byte loads straight into padded LDS rows (144 B pitch: the 16 lanes of
// one ds_read_b128 group hit 16 distinct 16-byte slots) with the loads of step k+1 in flight under the 32 MFMAs per wave of step k
// (register prefetch, two LDS stages, one barrier per step) -- twice the MFMAs per barrier of the generic implicit-GEMM kernel and no
// per-load tap / pixel arithmetic.  BM = 128, or 64 when the layer would not give every CU two workgroups otherwise (20x20 maps at
// small batches).  Same weight packing as the generic kernel ([cout_pad128][K]), so planning and packing are unchanged.
#include "kernels.h"
#include "elem16.h"
#include <stdlib.h>
#include <type_traits>

namespace adas {

typedef __attribute__((ext_vector_type(4))) float gf32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t gu32x4;

struct PwgDev {
    const uint16_t* in;
    const uint16_t* wgt;   // [cout_pad128][K]
    const float* bias;
    uint16_t* out;
    const uint16_t* res;
    int in_cs, in_coff, K, KP;   // K = Cin (any multiple of 8), KP = weight row pitch (K rounded up to 32)
    int out_cs, out_coff, cout;
    int res_cs, res_coff, res_mode;
    int act, M;
};

constexpr int PWG_BN = 128, PWG_KS = 64, PWG_LDK = PWG_KS + 8, PWG_D = 3;   // LDS row pitch in elements (144 B)

template <typename E, int BM>
__global__ __launch_bounds__(256, 2) void conv_pwg_kernel(PwgDev a) {
    E::enter();
    constexpr int WM = BM == 128 ? 2 : 1, WN = 4 / WM;            // wave grid over the tile
    constexpr int TM = BM / WM / 16, TN = PWG_BN / WN / 16;       // 16x16 MFMA tiles per wave: 4x4 (BM 128) or 4x2 (BM 64)
    constexpr int A_IT = BM * 8 / 256, B_IT = PWG_BN * 8 / 256;   // 16-byte chunks per thread per K step
    extern __shared__ __attribute__((aligned(16))) uint16_t pwg_lds[];   // As[2][BM][LDK] then Bs[2][BN][LDK]: 54 / 72 KB, two workgroups per CU
    uint16_t (*As)[BM][PWG_LDK] = reinterpret_cast<uint16_t (*)[BM][PWG_LDK]>(pwg_lds);
    uint16_t (*Bs)[PWG_BN][PWG_LDK] = reinterpret_cast<uint16_t (*)[PWG_BN][PWG_LDK]>(pwg_lds + 2 * BM * PWG_LDK);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * PWG_BN;
    const int kc = tid & 7;                 // 16-byte chunk of the 64-element K step
    const int r0 = tid >> 3;                // first row this thread stages (then +32 per iteration)

    const uint16_t* ap[A_IT];
    bool aok[A_IT];
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int m = m0 + r0 + 32 * i;
        aok[i] = m < a.M;
        ap[i] = a.in + (size_t)(aok[i] ? m : 0) * a.in_cs + a.in_coff + kc * 8;
    }
    const uint16_t* bp = a.wgt + (size_t)(n0 + r0) * a.KP + kc * 8;   // weight rows are padded to a multiple of 128: always in range

    // PWG_D register stages: the loads of K steps k+1 .. k+PWG_D are in flight under the MFMAs of step k.  One stage (round 3) left a
    // layer whose tiles do not fill the chip -- K = 1280 .. 2048 at 40x40 / 20x20, one frame -- bound by one memory round trip per step
    // (48 us for 32 steps); the loop is unrolled by PWG_D so that every stage is a fixed set of registers.
    gu32x4 ra[PWG_D][A_IT], rb[PWG_D][B_IT];
    // the last K step of a Cin that is not a multiple of 64: chunks past Cin are fetched from the step's first chunk (in range) and zeroed
    // by a select at the LDS store (no branch around the loads -- hipcc would wait vmcnt(0) at the join -- and no arithmetic on the loaded
    // registers before the store either: a select next to the load is a use, and a use ends the prefetch)
    // Branch-free on purpose: a conditional load or store splits the loop body into basic blocks and hipcc's wait-count insertion then
    // falls back to vmcnt(0) at every LDS store (measured: no prefetch at all).  So every step loads (a step past K re-reads step 0, in
    // range), every step stores (a step past K stores zeros, which the MFMAs add as zeros), and K is walked in whole rounds of PWG_D.
    auto gload = [&](auto st_c, int k0) {
        constexpr int st = decltype(st_c)::value;
        const int kq = k0 < a.K ? k0 : 0;
        const int ko = (kq + kc * 8 < a.K) ? kq : kq - kc * 8;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) ra[st][i] = *reinterpret_cast<const gu32x4*>(ap[i] + ko);   // rows past M re-read pixel 0: never stored
#pragma unroll
        for (int i = 0; i < B_IT; ++i) rb[st][i] = *reinterpret_cast<const gu32x4*>(bp + (size_t)(32 * i) * a.KP + ko);
        // keep the stages' loads in ISSUE order: hipcc's scheduler moves independent loads next to their first use, which put the stage
        // needed first LAST in the queue
        __builtin_amdgcn_sched_barrier(0);
    };
    auto lstore = [&](auto st_c, int buf, int k0) {
        constexpr int st = decltype(st_c)::value;
        const bool kin = k0 + kc * 8 < a.K;    // chunks past Cin (the tail of the last step, every chunk of a step past K): zeros
        const gu32x4 zero{0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < A_IT; ++i) *reinterpret_cast<gu32x4*>(&As[buf][r0 + 32 * i][kc * 8]) = kin ? ra[st][i] : zero;
#pragma unroll
        for (int i = 0; i < B_IT; ++i) *reinterpret_cast<gu32x4*>(&Bs[buf][r0 + 32 * i][kc * 8]) = kin ? rb[st][i] : zero;
        __builtin_amdgcn_sched_barrier(0);
    };

    gf32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = gf32x4{0.f, 0.f, 0.f, 0.f};

    const int lrow = lane & 15, kg = lane >> 4;
    const int KT = (a.K + PWG_KS - 1) / PWG_KS;
    const int KR = (KT + PWG_D - 1) / PWG_D * PWG_D;   // steps walked: whole rounds (the surplus steps multiply zeros)
    // K step s lives in register stage s % PWG_D; step 0 goes to LDS now, steps 1 .. PWG_D stay in flight
    gload(std::integral_constant<int, 0>{}, 0);
    gload(std::integral_constant<int, 1>{}, PWG_KS);
    gload(std::integral_constant<int, 2>{}, 2 * PWG_KS);
    lstore(std::integral_constant<int, 0>{}, 0, 0);
    gload(std::integral_constant<int, 0>{}, 3 * PWG_KS);
    __syncthreads();
    auto step = [&](auto u_c, int ks) {   // u = ks % PWG_D
        constexpr int u = decltype(u_c)::value;
        constexpr int nxt = (u + 1) % PWG_D;            // the stage that holds step ks + 1
        const int buf = ks & 1;
#pragma unroll
        for (int h = 0; h < 2; ++h) {   // the two 32-deep MFMA steps of this 64-deep K step
            gu32x4 wf[TN], xf[TM];
#pragma unroll
            for (int i = 0; i < TN; ++i) wf[i] = *reinterpret_cast<const gu32x4*>(&Bs[buf][(wn * TN + i) * 16 + lrow][h * 32 + kg * 8]);
#pragma unroll
            for (int j = 0; j < TM; ++j) xf[j] = *reinterpret_cast<const gu32x4*>(&As[buf][(wm * TM + j) * 16 + lrow][h * 32 + kg * 8]);
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j) acc[i][j] = E::mfma(wf[i], xf[j], acc[i][j]);
        }
        lstore(std::integral_constant<int, nxt>{}, buf ^ 1, (ks + 1) * PWG_KS);              // waits for step ks + 1's loads only
        gload(std::integral_constant<int, nxt>{}, (ks + 1 + PWG_D) * PWG_KS);                // its stage is free again
        __syncthreads();
    };
    for (int ks0 = 0; ks0 < KR; ks0 += PWG_D) {
        step(std::integral_constant<int, 0>{}, ks0);
        step(std::integral_constant<int, 1>{}, ks0 + 1);
        step(std::integral_constant<int, 2>{}, ks0 + 2);
    }

    // ---- epilogue: the weights are the MFMA A operand, so a lane holds channels c .. c+3 of pixel m
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int m = m0 + (wm * TM + j) * 16 + lrow;
        if (m >= a.M) continue;
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            const int c = n0 + (wn * TN + i) * 16 + kg * 4;
            if (c >= a.cout) continue;   // cout % 4 == 0: a 4-channel group is inside or outside as a whole
            const float4 b4 = *reinterpret_cast<const float4*>(a.bias + c);
            float v[4] = {acc[i][j][0] + b4.x, acc[i][j][1] + b4.y, acc[i][j][2] + b4.z, acc[i][j][3] + b4.w};
            float rv[4] = {0.f, 0.f, 0.f, 0.f};
            if (a.res_mode != RES_NONE) {
                const uint2 q = *reinterpret_cast<const uint2*>(a.res + (size_t)m * a.res_cs + a.res_coff + c);
                rv[0] = E::lo(q.x); rv[1] = E::hi(q.x); rv[2] = E::lo(q.y); rv[3] = E::hi(q.y);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float x = a.res_mode == RES_BEFORE_ACT ? v[t] + rv[t] : v[t];
                if (a.act == ACT_SILU) x = x * fast_rcp(1.0f + __expf(-x));
                else if (a.act == ACT_RELU) x = fmaxf(x, 0.f);
                else if (a.act == ACT_LEAKY) x = fmaxf(x, 0.1f * x);
                v[t] = a.res_mode == RES_AFTER_ACT ? x + rv[t] : x;
            }
            uint2 q;
            q.x = E::pack2(v[0], v[1]);
            q.y = E::pack2(v[2], v[3]);
            *reinterpret_cast<uint2*>(a.out + (size_t)m * a.out_cs + a.out_coff + c) = q;
        }
    }
}

static bool pwg_enabled() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("ADAS_NO_PWG");
        v = (e && e[0] == '1') ? 0 : 1;
    }
    return v == 1;
}

bool pwg_applicable(int prec, int kh, int kw, int stride, int pad, const TView& in, const TView& out, const TView& res, int res_mode) {
    if (!pwg_enabled() || !prec_is16(prec) || in.f32 || out.f32) return false;
    if (kh != 1 || kw != 1 || stride != 1 || pad != 0) return false;
    if (in.h == 1 && in.w == 1) return false;   // Linear layers: conv_fc.hip
    if ((in.c & 7) || in.c < 2 * PWG_KS || (in.cs & 7) || (in.coff & 7) || (out.c & 3) || (out.cs & 3) || (out.coff & 3)) return false;
    if (res_mode != RES_NONE && (res.f32 || (res.cs & 3) || (res.coff & 3))) return false;
    return true;
}

// BM: 128-pixel tiles unless they would leave the chip with fewer than two workgroups per CU
static int pwg_bm(int m, int cout) {
    const long wgs128 = (long)((m + 127) / 128) * ((cout + PWG_BN - 1) / PWG_BN);
    return wgs128 >= 512 ? 128 : 64;
}
const char* pwg_kernel_name(int m, int cout) { return pwg_bm(m, cout) == 128 ? "conv_pwg_kernel<128>" : "conv_pwg_kernel<64>"; }

hipError_t launch_conv_pwg(const ConvArgs& a, hipStream_t st) {
    if (!pwg_applicable(a.prec, a.kh, a.kw, a.stride, a.pad, a.in, a.out, a.res, a.res_mode) || a.kpad < a.in.c) return hipErrorNotSupported;
    PwgDev d;
    d.in = (const uint16_t*)a.in.p; d.wgt = (const uint16_t*)a.wgt; d.bias = a.bias; d.out = (uint16_t*)a.out.p; d.res = (const uint16_t*)a.res.p;
    d.in_cs = a.in.cs; d.in_coff = a.in.coff; d.K = a.in.c; d.KP = a.kpad;
    d.out_cs = a.out.cs; d.out_coff = a.out.coff; d.cout = a.out.c;
    d.res_cs = a.res.cs; d.res_coff = a.res.coff; d.res_mode = a.res_mode;
    d.act = a.act; d.M = a.m;
    const int bm = pwg_bm(a.m, a.out.c);
    const dim3 grid((a.m + bm - 1) / bm, (a.out.c + PWG_BN - 1) / PWG_BN);
    const size_t lds = (size_t)2 * (bm + PWG_BN) * PWG_LDK * 2;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)conv_pwg_kernel<Fp16, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        (void)hipFuncSetAttribute((const void*)conv_pwg_kernel<Fp16, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        (void)hipFuncSetAttribute((const void*)conv_pwg_kernel<Bf16, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        (void)hipFuncSetAttribute((const void*)conv_pwg_kernel<Bf16, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        attr_done = true;
    }
    if (a.prec == PREC_FP16) {
        if (bm == 128) hipLaunchKernelGGL((conv_pwg_kernel<Fp16, 128>), grid, dim3(256), lds, st, d);
        else hipLaunchKernelGGL((conv_pwg_kernel<Fp16, 64>), grid, dim3(256), lds, st, d);
    } else {
        if (bm == 128) hipLaunchKernelGGL((conv_pwg_kernel<Bf16, 128>), grid, dim3(256), lds, st, d);
        else hipLaunchKernelGGL((conv_pwg_kernel<Bf16, 64>), grid, dim3(256), lds, st, d);
    }
    return hipGetLastError();
}

}  // namespace adas
