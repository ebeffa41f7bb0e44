// conv_fc.hip -- the two Linear layers of the UFLDv2 head (model_culane.py:33-37: 4000 -> 2048 -> 91,224)
// at M = batch <= 64 rows: a weight-streaming kernel, HBM-bound on the bf16 weight matrix
// (cls.3: 2048 x 91,224 x 2 B = 374 MB per step).
//
// The implicit-GEMM conv kernel tiles M = batch into 64/128-row blocks, which leaves cls.1 with 32
// workgroups walking K = 4000 serially (123 us for 16 MB of weights) and cls.3 at 2.5 TB/s.  Here every
// wave owns TN x 16 output features and streams its weight rows straight from HBM into MFMA A-operand
// registers (no LDS: a 16x16x32 bf16 A fragment is 16 rows x 64 contiguous bytes, lane (row, kg) loads its
// own 16 B); the activations (<= 64 x K bf16, L2-resident) are the B operand, loaded the same way.
// The weights are packed at load time in fragment order ("CONV_FC" packing): the 1 KB a wave needs for
// (feature tile t, K step s) is one contiguous block [t][s][lane][8], so every wave-level load is a fully
// coalesced 1 KB read and a wave walks one contiguous 16-row slab of the matrix front to back.
// A U-deep register ring keeps U K-steps of loads in flight per wave.  Layers with few output tiles
// (cls.1) split K across the KS waves of a workgroup and reduce through LDS before the fused
// bias + ReLU epilogue.
#include "kernels.h"
#include "elem16.h"
#include <stdlib.h>

namespace adas {

typedef __attribute__((ext_vector_type(4))) float ff32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t fu32x4;

struct FcDev {
    const uint16_t* x;    // [batch][x_cs] bf16 (+ x_coff)
    const uint16_t* w;    // [cout_pad][kpad] bf16, zero padded
    const float* bias;    // [cout_pad]
    void* out;            // [batch][out_cs] (+ out_coff), bf16 or fp32
    int x_cs, x_coff, out_cs, out_coff;
    int batch, cout, kpad, act, out_f32;
};

template <typename E, int TN, int TM, int KS, int U>
__global__ __launch_bounds__(64 * KS) void fc_kernel(FcDev a) {
    E::enter();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = lane & 15, kg = lane >> 4;
    const int n0 = blockIdx.x * (TN * 16);
    const int KT = a.kpad >> 5;
    const int ks0 = (int)((long)KT * wave / KS), ks1 = (int)((long)KT * (wave + 1) / KS);
    const int n = ks1 - ks0;

    // fragment-ordered weights: block (tile, ks) is 64 lanes x 8 bf16
    const uint16_t* wp[TN];
#pragma unroll
    for (int i = 0; i < TN; ++i) wp[i] = a.w + ((size_t)(blockIdx.x * TN + i) * KT + ks0) * 512 + lane * 8;
    const uint16_t* xp[TM];
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int m = j * 16 + lrow;
        xp[j] = a.x + (size_t)(m < a.batch ? m : 0) * a.x_cs + a.x_coff + kg * 8 + ks0 * 32;
    }

    ff32x4 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = ff32x4{0.f, 0.f, 0.f, 0.f};

    fu32x4 wa[U][TN], xb[U][TM];
    auto load = [&](int buf, int s) {  // s: K step relative to ks0
#pragma unroll
        for (int i = 0; i < TN; ++i) wa[buf][i] = __builtin_nontemporal_load(reinterpret_cast<const fu32x4*>(wp[i] + (size_t)s * 512));
#pragma unroll
        for (int j = 0; j < TM; ++j) xb[buf][j] = *reinterpret_cast<const fu32x4*>(xp[j] + s * 32);
    };
    auto mma = [&](int buf) {
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j)
                acc[i][j] = E::mfma(wa[buf][i], xb[buf][j], acc[i][j]);
    };
    if (n >= U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            load(u, u);
            __builtin_amdgcn_sched_barrier(0);  // keep the ring stages in issue order: the counted waits below
        }                                       // can then leave U-1 stages in flight (hipcc interleaves them otherwise)
        const int main_steps = ((n - U) / U) * U;  // steps whose ring slot is refilled unconditionally
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

    // rows of batch entries that do not exist accumulated row 0's activations: never stored.
    if constexpr (KS > 1) {
        __shared__ float red[KS > 1 ? KS - 1 : 1][TN * TM * 4][64];
        if (wave > 0) {
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) red[wave - 1][(i * TM + j) * 4 + r][lane] = acc[i][j][r];
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
                    for (int r = 0; r < 4; ++r) acc[i][j][r] += red[s][(i * TM + j) * 4 + r][lane];
    }

    // ---- epilogue: lane holds features c..c+3 of batch row m
#pragma unroll
    for (int i = 0; i < TN; ++i) {
        const int c = n0 + i * 16 + kg * 4;
        if (c >= a.cout) continue;
        const float4 b4 = *reinterpret_cast<const float4*>(a.bias + c);
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int m = j * 16 + lrow;
            if (m >= a.batch) continue;
            float v[4] = {acc[i][j][0] + b4.x, acc[i][j][1] + b4.y, acc[i][j][2] + b4.z, acc[i][j][3] + b4.w};
            if (a.act == ACT_RELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            } else if (a.act == ACT_SILU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = v[r] * fast_rcp(1.0f + __expf(-v[r]));
            }
            const size_t o = (size_t)m * a.out_cs + a.out_coff + c;
            if (a.out_f32) {
                *reinterpret_cast<float4*>((float*)a.out + o) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
                uint2 q;
                q.x = E::pack2(v[0], v[1]);
                q.y = E::pack2(v[2], v[3]);
                *reinterpret_cast<uint2*>((uint16_t*)a.out + o) = q;
            }
        }
    }
}

// Static-shape test used both at load time (weight packing) and at launch time: max_n = the engine's max_batch.
bool fc_applicable(int prec, int kh, int kw, int stride, int max_n, const TView& in, const TView& out) {
    if (!prec_is16(prec) || in.f32) return false;
    if (kh != 1 || kw != 1 || stride != 1 || in.h != 1 || in.w != 1 || out.h != 1 || out.w != 1) return false;
    (void)max_n;  // any batch: launches walk the batch in groups of <= 64 rows (the weights stream once per group)
    if ((in.cs & 7) || (in.coff & 7) || (in.c & 7) || (out.c & 3) || (out.cs & 3) || (out.coff & 3)) return false;
    return true;
}

template <typename E, int TN, int TM, int KS, int U>
static hipError_t fc_launch(const FcDev& d, hipStream_t st) {
    const int tiles = (d.cout + TN * 16 - 1) / (TN * 16);
    hipLaunchKernelGGL((fc_kernel<E, TN, TM, KS, U>), dim3(tiles), dim3(64 * KS), 0, st, d);
    return hipGetLastError();
}

static int fc_wide_ks() {  // K split of the wide (weight-streaming) path at 33..64 rows; ADAS_FC_KS overrides (1 | 2 | 4)
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("ADAS_FC_KS");
        v = e ? atoi(e) : 1;   // measured at 64 rows on cls.3 (374 MB of weights): 1 -> 95.7 us, 2 -> 98.9 us, 4 -> 120.9 us
        if (v != 1 && v != 2 && v != 4) v = 1;
    }
    return v;
}

template <typename E>
static hipError_t fc_launch_rows(const FcDev& d, int cout, hipStream_t st) {
    const int tm = d.batch <= 16 ? 1 : (d.batch <= 32 ? 2 : 4);
    // few output tiles (cls.1: 2048 features): 16 features per workgroup, K split over 4 waves
    if (cout <= 8192) {
        if (tm == 1) return fc_launch<E, 1, 1, 4, 4>(d, st);
        if (tm == 2) return fc_launch<E, 1, 2, 4, 4>(d, st);
        return fc_launch<E, 1, 4, 4, 4>(d, st);
    }
    if (tm == 1) return fc_launch<E, 4, 1, 1, 4>(d, st);
    if (tm == 2) return fc_launch<E, 4, 2, 1, 4>(d, st);
    // 33..64 rows: 3.9 TB/s on cls.3.  Splitting K over the waves of a workgroup (more resident waves, partial sums through LDS)
    // was measured and does not help (see fc_wide_ks): the launch is not short of bytes in flight.  Deeper rings or narrower
    // workgroups neither: <4,4,1,U=4> 111 us, U=5 114 us (276-280 VGPRs: one wave per SIMD), <2,4,1,U=4> 125 us, <2,4,1,U=6> 116 us
    // against 94 us for <4,4,1,3>.
    const int ks = fc_wide_ks();
    if (ks == 4) return fc_launch<E, 4, 4, 4, 3>(d, st);
    if (ks == 2) return fc_launch<E, 4, 4, 2, 3>(d, st);
    return fc_launch<E, 4, 4, 1, 3>(d, st);
}

hipError_t launch_fc(const ConvArgs& a, hipStream_t st) {
    for (int r0 = 0; r0 < a.n; r0 += 64) {  // groups of <= 64 batch rows
        FcDev d;
        const int rows = a.n - r0 < 64 ? a.n - r0 : 64;
        const size_t osz = a.out.f32 ? 4 : 2;
        d.x = (const uint16_t*)a.in.p + (size_t)r0 * a.in.cs;
        d.w = (const uint16_t*)a.wgt; d.bias = a.bias;
        d.out = (char*)a.out.p + (size_t)r0 * a.out.cs * osz;
        d.x_cs = a.in.cs; d.x_coff = a.in.coff; d.out_cs = a.out.cs; d.out_coff = a.out.coff;
        d.batch = rows; d.cout = a.out.c; d.kpad = a.kpad; d.act = a.act; d.out_f32 = a.out.f32;
        hipError_t e = a.prec == PREC_FP16 ? fc_launch_rows<Fp16>(d, a.out.c, st) : fc_launch_rows<Bf16>(d, a.out.c, st);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// fp32 [cout][cin] -> bf16 fragment order [cout_pad/16][kpad/32][64 lanes][8]: lane = (k%32/8)*16 + row%16
template <typename E>
__global__ void pack_weights_fc_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst, int cout, int cin, int kpad, size_t total) {
    const int KT = kpad >> 5;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int e = (int)(i & 7), lane = (int)((i >> 3) & 63);
        const size_t blk = i >> 9;
        const int ks = (int)(blk % KT);
        const size_t tile = blk / KT;
        const size_t row = tile * 16 + (lane & 15);
        const int k = ks * 32 + (lane >> 4) * 8 + e;
        const float v = (row < (size_t)cout && k < cin) ? src[row * cin + k] : 0.0f;
        dst[i] = E::from_f32(v);
    }
}

hipError_t launch_pack_weights_fc(const float* src, void* dst, int cout, int cout_pad, int cin, int kpad, int prec, hipStream_t st) {
    if (prec == PREC_X3) return launch_pack_weights_fcx3(src, dst, cout, cout_pad, cin, kpad, st);
    size_t total = (size_t)cout_pad * kpad;
    int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    ADAS_DISPATCH_E16(prec == PREC_FP16, E,
                      hipLaunchKernelGGL(pack_weights_fc_kernel<E>, dim3(blocks), dim3(256), 0, st, src, (uint16_t*)dst, cout, cin, kpad, total));
    return hipGetLastError();
}

}  // namespace adas
